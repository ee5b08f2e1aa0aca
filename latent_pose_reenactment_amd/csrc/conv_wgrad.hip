// Weight-gradient of the fused conv for gfx950:
//   dw[co][ci][tap] = sum_{n,y,x} dy[n,y,x,co] * up2?(a)[n, y+dy_t, x+dx_t, ci]
// Both operands arrive as 16-bit operand planes (hi [, lo]; lp_act_pack / the conv epilogue): `a` is the ACTIVATED input the
// forward conv multiplied (saved by the forward pass, so no prologue is recomputed here), dy the packed output gradient.
// GEMM view: M = co, N = ci (per tap), K = pixels.  Both operands are pixel-major in HBM (NHWC), i.e. K is the slow
// dimension of both -> the MFMA fragments (8 consecutive k per lane) are fetched from [pixel][channel] LDS tiles with
// ds_read_b64_tr_b16 (gfx950 transposing LDS read; lane/element mapping verified by probes/probe_layout.hip):
//   within a 16-lane group, lane i / element j receives the data of source lane 4j+(i>>2), element i&3; so source lane
//   s = 4j+q must point at &tile[pixel_j][c0 + 4q] and lane i ends up with channel c0+i for the pixels 0..3.
// One workgroup owns a 64(co) x 64(ci) x all-taps slab and walks a strided subset of the 128-pixel tiles (split-K);
// the activated input halo and the dy tile are staged once per tile and reused by all taps.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"
#include <stdlib.h>

struct WgradParams {
    const uint16_t* a_hi; const uint16_t* a_lo; const uint16_t* d_hi; const uint16_t* d_lo; float* part;
    float* bpart;          // NULL | [splits][CoP] partial bias gradients (column sums of dy), written by the ci-block-0 workgroups
    int N, H, W, Hin, Win, Cin, Cout, C8, Co8, CoP, CiP;
    int splits, num_tiles;
    int lTH, lTW, lNB, tiles_x, tiles_y;
    int db_acc;            // add the bias gradient to dbias instead of storing it
    int diag;              // block-diagonal (grouped conv): the workgroup of output channels [co0, co0+64) only pairs with input channels
                           // [co0, co0+64); slab layout [split][tap][CoP][64] (column = ci - co0)
};

// zero-masked 16-byte load of 8 consecutive 16-bit channels
__device__ __forceinline__ s16x8_t ld16x8(const uint16_t* p) { return *(const s16x8_t*)p; }
template <bool F16> __device__ __forceinline__ void acc8(float (&s)[8], s16x8_t v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += lp_op16_to_f32<F16>((uint16_t)v[j]);
}

typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ s16x4_t tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)p);
}

// COB = output channels per workgroup: 64 (4 waves) or 128 (8 waves, two per SIMD: the staged input halo is shared by twice
// the matrix work, and the per-thread share of the fp32 -> bf16 staging work shrinks accordingly).
template <int KS, bool UPS, int PREC, int COB = 64>
__global__ __launch_bounds__(COB * 4) void conv_wgrad_kernel(WgradParams p) {
    constexpr int NT = COB * 4;                    // threads: one wave per 32(co) x 32(ci) sub-block
    constexpr int DCG = COB / 8;                   // 8-channel groups of the dy tile
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    constexpr int T = KS * KS;
    constexpr int CC = 64;
    constexpr int SA = CC * 2 + 16, SD = COB * 2 + 16;
    constexpr int BMP = 128;                       // pixels per tile (k extent per stage)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mh = wave >> 1, nh = wave & 1;       // wave -> 32(co) x 32(ci) sub-block
    const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;
    const int co0 = blockIdx.y * COB, ci0 = p.diag ? co0 : blockIdx.z * 64;

    int HH, HW;
    if (KS == 1) { HH = TH; HW = TW; } else if (UPS) { HH = (TH >> 1) + 2; HW = (TW >> 1) + 2; } else { HH = TH + 2; HW = TW + 2; }
    const int a_bytes = NBv * HH * HW * SA;
    unsigned char* A_hi = smem;
    unsigned char* A_lo = smem + a_bytes;
    unsigned char* D_hi = smem + (SPLIT ? 2 : 1) * a_bytes;
    unsigned char* D_lo = D_hi + BMP * SD;

    f32x4_t acc[T][2][2];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // bias gradient: every thread stages dy items of ONE 8-channel group (tid % DCG) -> private fp32 column sums, combined at the end
    float dsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // source-lane role for the transposing reads
    const int G = lane >> 4, sj = (lane & 15) >> 2, sq = lane & 3;
    // fast staging (one image per tile, channel counts multiple of 8): 8 channel groups x 32 pixels per pass of the block
    constexpr int APP = NT / 8;                              // halo pixels per pass of the workgroup (8 channel groups each)
    constexpr int AIT = (10 * 18 + APP - 1) / APP;           // 8x16 patch + border = 180 halo pixels -> 6 (3) items per thread
    const int halo_px = NBv * HH * HW;
    // measured: pays only where the kernel is LDS-limited to one workgroup per CU anyway (bf16x3, full-size halo); the bf16 kernel
    // keeps its small register footprint (2-3 workgroups per CU hide the staging latency instead)
    const bool fast = SPLIT && !UPS && (NBv == 1) && (halo_px <= AIT * APP);
    const int a_cg = tid & 7, a_hp0 = tid >> 3;
    const int d_cg = tid % DCG, d_p0 = tid / DCG;           // dy tile: 128 pixels x DCG groups = 4 items per thread

    // tile geometry (tile index -> image / patch origin)
    auto tile_origin = [&](int tile, int& n0, int& y0, int& x0, int& oy, int& ox) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; const int ng = t / p.tiles_y;
        n0 = ng << p.lNB; y0 = ty << p.lTH; x0 = tx << p.lTW;
        if (KS == 1) { oy = y0; ox = x0; } else if (UPS) { oy = (y0 >> 1) - 1; ox = (x0 >> 1) - 1; } else { oy = y0 - 1; ox = x0 - 1; }
    };
    // MFMAs of the tile that is staged in LDS (tile independent: the pixel -> patch decode only depends on the tile shape)
    auto compute_tile = [&]() {
#pragma unroll 1
        for (int ks = 0; ks < BMP / 32; ++ks) {
            // this lane (as a SOURCE lane) serves pixel kp(h) = ks*32 + h*16 + G*4 + sj for the two halves h
            int kp0 = ks * 32 + G * 4 + sj, kp1 = kp0 + 16;
            int nb0, py0, px0, nb1, py1, px1;
            tile_lin_decode(kp0, p.lTH, p.lTW, nb0, py0, px0);
            tile_lin_decode(kp1, p.lTH, p.lTW, nb1, py1, px1);
            // A operand (dy): fragments mf = 0,1 -> co = mh*32 + mf*16 + ...
            s16x8_t a[2], al[2];
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                int coff = (mh * 32 + mf * 16 + 4 * sq) * 2;
                s16x4_t v0 = tr_read(D_hi + kp0 * SD + coff), v1 = tr_read(D_hi + kp1 * SD + coff);
                a[mf] = (s16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (SPLIT) {
                    s16x4_t w0 = tr_read(D_lo + kp0 * SD + coff), w1 = tr_read(D_lo + kp1 * SD + coff);
                    al[mf] = (s16x8_t){w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                }
            }
            // B operand (activated input) fragments of tap t+1 are requested before the MFMAs of tap t (register double buffer):
            // the bf16x3 kernel runs one wave per SIMD, so nothing else would hide the LDS latency of 8 transposing reads per tap
            s16x8_t bf[2][2], bfl[2][2];
            auto fetch_b = [&](int tap, int set) {
                const int dy = (KS == 3) ? tap / 3 : 0, dx = (KS == 3) ? tap % 3 : 0;
                int hp0, hp1;
                if (KS == 1) { hp0 = nb0 * HH * HW + py0 * HW + px0; hp1 = nb1 * HH * HW + py1 * HW + px1; }
                else if (UPS) {
                    hp0 = nb0 * HH * HW + (((py0 + dy - 1) >> 1) + 1) * HW + ((px0 + dx - 1) >> 1) + 1;
                    hp1 = nb1 * HH * HW + (((py1 + dy - 1) >> 1) + 1) * HW + ((px1 + dx - 1) >> 1) + 1;
                } else {
                    hp0 = nb0 * HH * HW + (py0 + dy) * HW + px0 + dx;
                    hp1 = nb1 * HH * HW + (py1 + dy) * HW + px1 + dx;
                }
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    int coff = (nh * 32 + nf * 16 + 4 * sq) * 2;
                    s16x4_t v0 = tr_read(A_hi + hp0 * SA + coff), v1 = tr_read(A_hi + hp1 * SA + coff);
                    bf[set][nf] = (s16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if (SPLIT) {
                        s16x4_t w0 = tr_read(A_lo + hp0 * SA + coff), w1 = tr_read(A_lo + hp1 * SA + coff);
                        bfl[set][nf] = (s16x8_t){w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                    }
                }
            };
            fetch_b(0, 0);
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                const int cur = tap & 1;
                if (tap + 1 < T) { fetch_b(tap + 1, cur ^ 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    if (SPLIT) {
#pragma unroll
                        for (int mf = 0; mf < 2; ++mf) {
                            acc[tap][mf][nf] = mfma16(al[mf], bf[cur][nf], acc[tap][mf][nf]);
                            acc[tap][mf][nf] = mfma16(a[mf], bfl[cur][nf], acc[tap][mf][nf]);
                        }
                    }
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) acc[tap][mf][nf] = mfma16t<F16>(a[mf], bf[cur][nf], acc[tap][mf][nf]);
                }
            }
        }
    };

    if (SPLIT && !UPS && fast) {
        // bf16x3: registers and LDS admit one workgroup per CU, so nothing else hides the global latency of the staging loads:
        // the loads of the NEXT tile are issued before the MFMAs of the current one and written to LDS afterwards
        // (issue and use inside one loop iteration, unconditional with a clamped tile index -> no early waits).
        s16x8_t ald[AIT][2], dld[4][2];
        int apix[AIT], dpix[4];
        int n0, y0, x0, oy, ox;
        const int cA = ci0 + a_cg * 8, cD = co0 + d_cg * 8;
        const bool cokA = cA < p.C8, cokD = cD < p.Co8;
        auto issue_loads = [&](int tile) {
            tile_origin(tile, n0, y0, x0, oy, ox);
#pragma unroll
            for (int k = 0; k < AIT; ++k) {
                const int hp = a_hp0 + k * APP;
                const int hx = hp % HW, hy = hp / HW;
                const int iy = oy + hy, ix = ox + hx;
                const bool inb = (hp < halo_px) && (n0 < p.N) && (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
                apix[k] = inb ? ((n0 * p.Hin + iy) * p.Win + ix) : (hp < halo_px ? -1 : -2);
                const size_t off = (size_t)(apix[k] >= 0 ? apix[k] : 0) * p.C8 + (cokA ? cA : 0);
                ald[k][0] = ld16x8(p.a_hi + off); ald[k][1] = ld16x8(p.a_lo + off);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kp = d_p0 + k * 32;
                int nb, py, px;
                tile_lin_decode(kp, p.lTH, p.lTW, nb, py, px);
                const int n = n0 + nb, yy = y0 + py, xx = x0 + px;
                dpix[k] = (n < p.N && yy < p.H && xx < p.W) ? ((n * p.H + yy) * p.W + xx) : -1;
                const size_t off = (size_t)(dpix[k] >= 0 ? dpix[k] : 0) * p.Co8 + (cokD ? cD : 0);
                dld[k][0] = ld16x8(p.d_hi + off); dld[k][1] = ld16x8(p.d_lo + off);
            }
        };
        auto convert_write = [&]() {
            const s16x8_t z = (s16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < AIT; ++k) {
                const bool keep = (apix[k] >= 0) && cokA;
                const int off = (a_hp0 + k * APP) * SA + a_cg * 16;
                if (apix[k] != -2) {
                    *(s16x8_t*)(A_hi + off) = keep ? ald[k][0] : z;
                    *(s16x8_t*)(A_lo + off) = keep ? ald[k][1] : z;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool keep = (dpix[k] >= 0) && cokD;
                const s16x8_t h = keep ? dld[k][0] : z, l = keep ? dld[k][1] : z;
                acc8<false>(dsum, h); acc8<false>(dsum, l);
                const int off = (d_p0 + k * 32) * SD + d_cg * 16;
                *(s16x8_t*)(D_hi + off) = h;
                *(s16x8_t*)(D_lo + off) = l;
            }
        };
        if (COB == 64) {
            int tile = blockIdx.x;
            if (tile < p.num_tiles) { issue_loads(tile); convert_write(); }
            for (; tile < p.num_tiles; tile += p.splits) {
                const int next = tile + p.splits;
                issue_loads(next < p.num_tiles ? next : tile);       // (past the end: harmless re-load of the current tile)
                __syncthreads();                                     // staged tile visible
                compute_tile();
                __syncthreads();                                     // everyone is done reading the tile
                if (next < p.num_tiles) convert_write();
            }
        } else {
            // 8 waves share 256 registers per SIMD lane pair: no room to hold the next tile's loads across the MFMAs
            for (int tile = blockIdx.x; tile < p.num_tiles; tile += p.splits) {
                issue_loads(tile);
                __syncthreads();                                     // previous tile fully consumed
                convert_write();
                __syncthreads();
                compute_tile();
            }
        }
    } else {
        const s16x8_t z = (s16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
        for (int tile = blockIdx.x; tile < p.num_tiles; tile += p.splits) {
            int n0, y0, x0, oy, ox;
            tile_origin(tile, n0, y0, x0, oy, ox);
            __syncthreads();                 // previous tile fully consumed
            // activated input halo: [halo pixel][64 ci] 16-bit, out-of-image pixels and channels beyond C8 are zero
            for (int i = tid; i < halo_px * 8; i += NT) {
                const int cg = i & 7, hp = i >> 3;
                const int hx = hp % HW, t2 = hp / HW;
                const int hy = t2 % HH, nb = t2 / HH;
                const int n = n0 + nb, iy = oy + hy, ix = ox + hx, c = ci0 + cg * 8;
                const bool inb = (n < p.N) && (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win) && (c < p.C8);
                const size_t off = inb ? ((size_t)((n * p.Hin + iy) * p.Win + ix) * p.C8 + c) : 0;
                const s16x8_t h = ld16x8(p.a_hi + off);
                *(s16x8_t*)(A_hi + (size_t)hp * SA + cg * 16) = inb ? h : z;
                if (SPLIT) { const s16x8_t l = ld16x8(p.a_lo + off); *(s16x8_t*)(A_lo + (size_t)hp * SA + cg * 16) = inb ? l : z; }
            }
            // dy tile: [128 pixels (row-major in patch)][COB co] 16-bit
            for (int i = tid; i < BMP * DCG; i += NT) {
                const int cg = i % DCG, kp = i / DCG;
                int nb, py, px;
                tile_lin_decode(kp, p.lTH, p.lTW, nb, py, px);
                const int n = n0 + nb, yy = y0 + py, xx = x0 + px, c = co0 + cg * 8;
                const bool inb = (n < p.N && yy < p.H && xx < p.W && c < p.Co8);
                const size_t off = inb ? ((size_t)((n * p.H + yy) * p.W + xx) * p.Co8 + c) : 0;
                const s16x8_t h0 = ld16x8(p.d_hi + off);
                const s16x8_t h = inb ? h0 : z;
                acc8<F16>(dsum, h);
                *(s16x8_t*)(D_hi + kp * SD + cg * 16) = h;
                if (SPLIT) {
                    const s16x8_t l0 = ld16x8(p.d_lo + off);
                    const s16x8_t l = inb ? l0 : z;
                    acc8<false>(dsum, l);
                    *(s16x8_t*)(D_lo + kp * SD + cg * 16) = l;
                }
            }
            __syncthreads();
            compute_tile();
        }
    }

    if (p.bpart && blockIdx.z == 0) {                      // (uniform) column sums of dy over this workgroup's tiles
        __syncthreads();
        float* red = (float*)smem;                         // [NT / DCG][COB]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(tid / DCG) * COB + (tid % DCG) * 8 + j] = dsum[j];
        __syncthreads();
        if (tid < COB) {
            float s = 0.f;
            for (int r = 0; r < NT / DCG; ++r) s += red[r * COB + tid];
            if (co0 + tid < p.CoP) p.bpart[(size_t)blockIdx.x * p.CoP + co0 + tid] = s;
        }
    }
    // write the partial slab [split][tap][CoP][CiP]; C layout: row (co) = (lane>>4)*4 + r, col (ci) = lane&15
#pragma unroll
    for (int tap = 0; tap < T; ++tap)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int co = co0 + mh * 32 + mf * 16 + (lane >> 4) * 4 + r;
                    int ci = (p.diag ? 0 : ci0) + nh * 32 + nf * 16 + (lane & 15);
                    if (COB == 64 || co < p.CoP) p.part[(((size_t)blockIdx.x * T + tap) * p.CoP + co) * p.CiP + ci] = acc[tap][mf][nf][r];
                }
}

// trailing blocks of both reductions: dbias[co] = sum_s bpart[s][co], 64 channels per block, the splits dealt to 4 thread groups x 4
// independent accumulators
__device__ __forceinline__ void wred_bias_block(const float* __restrict__ bpart, float* __restrict__ dbias, int S, int Cout, int CoP, int blk,
                                                float osc, int accumulate) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int co = blk * 64 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (co < Cout) {
        int k = g;
        for (; k + 12 < S; k += 16) {
            a0 += bpart[(size_t)k * CoP + co]; a1 += bpart[(size_t)(k + 4) * CoP + co];
            a2 += bpart[(size_t)(k + 8) * CoP + co]; a3 += bpart[(size_t)(k + 12) * CoP + co];
        }
        for (; k < S; k += 4) a0 += bpart[(size_t)k * CoP + co];
    }
    red[g][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && co < Cout) {
        const float v = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) * osc;
        dbias[co] = accumulate ? dbias[co] + v : v;          // accumulate: dbias is the parameter's .grad (fused gradient accumulation)
    }
}

// dw[co][ci][tap] = sum_s part[s][tap][co][ci].  A block owns one output channel and WRED_CI input channels: the slabs are read with
// ci fastest (coalesced 4-byte lanes, 8 independent accumulators keep 8 loads in flight per thread; the slabs are streamed once
// from HBM/L2), the sums cross an LDS tile [ci][tap], and the block's WRED_CI * T outputs -- one contiguous run of the reference
// [Cout][Cin][k][k] layout -- leave as coalesced stores (a direct write is a 36-byte stride per lane: 8x write amplification).
// sn_w != NULL: the layer is spectrally normalised -- the block also leaves its share of <dw, W_orig> in sn_dot[blockIdx.x]
// (lp_sn_grad_apply needs that inner product; taking it here saves a pass over dw and a launch).
#define WRED_CI 128
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int T,
                                                           int Cout, int Cin, int CoP, int CiP, const float* __restrict__ bpart,
                                                           float* __restrict__ dbias, int wblocks, const float* __restrict__ out_scale,
                                                           const float* __restrict__ sn_w, float* __restrict__ sn_dot, int db_acc) {
    const float osc = out_scale ? out_scale[0] : 1.f;          // 1 / (input scale of the fp16 dy operand)
    if ((int)blockIdx.x >= wblocks) { wred_bias_block(bpart, dbias, S, Cout, CoP, blockIdx.x - wblocks, osc, db_acc); return; }
    __shared__ float tile[WRED_CI * 9];
    const int cib = (Cin + WRED_CI - 1) / WRED_CI;
    const int co = blockIdx.x / cib, ci0 = (blockIdx.x - co * cib) * WRED_CI;
    const int nci = min(WRED_CI, Cin - ci0), nel = nci * T;
    const size_t slab = (size_t)T * CoP * CiP;
    for (int e = threadIdx.x; e < T * WRED_CI; e += 256) {
        const int t = e / WRED_CI, ci = e - t * WRED_CI;
        if (ci < nci) {
            const float* p = part + ((size_t)t * CoP + co) * CiP + ci0 + ci;
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 8 <= S; k += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] += p[(size_t)(k + q) * slab];
            }
            for (; k < S; ++k) a[0] += p[(size_t)k * slab];
            tile[ci * T + t] = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) * osc;
        }
    }
    __syncthreads();
    const size_t obase = ((size_t)co * Cin + ci0) * T;
    float dsum = 0.f;
    for (int e = threadIdx.x; e < nel; e += 256) {
        const float val = tile[e];
        dw[obase + e] = val;
        if (sn_w) dsum = fmaf(val, sn_w[obase + e], dsum);
    }
    if (sn_w) {
        __shared__ float dred[4];
        for (int o = 32; o > 0; o >>= 1) dsum += __shfl_down(dsum, o, 64);
        if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = dsum;
        __syncthreads();
        if (threadIdx.x == 0) sn_dot[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
    }
}

// Small layers (few weights, many pixel splits: 64 x 64 x 9 weights summed over up to 512 slabs): 32 outputs (tap, co, ci; ci fastest: one
// 128-byte run per slab) x 8 groups of slabs per block, so that the grid is as wide as the layer allows AND the slab loop is short (round 4:
// one thread per output over 512 slabs left the 64 x 64 layer at 145 blocks, 25 us for 75 MB); the 8 group sums meet in LDS in a fixed
// order.  The strided 4-byte stores do not matter at these sizes.
template <int WREDF_OUT>                                  // outputs per block: 32 (x 8 slab groups) | 256 (x 1: layers that fill the chip anyway)
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int T, int Cout,
                                                                int Cin, int CoP, int CiP, const float* __restrict__ bpart,
                                                                float* __restrict__ dbias, int wblocks, const float* __restrict__ out_scale,
                                                                const float* __restrict__ sn_w, float* __restrict__ sn_dot, int db_acc) {
    const float osc = out_scale ? out_scale[0] : 1.f;
    if ((int)blockIdx.x >= wblocks) { wred_bias_block(bpart, dbias, S, Cout, CoP, blockIdx.x - wblocks, osc, db_acc); return; }
    constexpr int SG = 256 / WREDF_OUT;
    __shared__ float gsum[SG][WREDF_OUT];
    const int lo = threadIdx.x % WREDF_OUT, sg = threadIdx.x / WREDF_OUT;
    const int idx = blockIdx.x * WREDF_OUT + lo;
    const bool live = idx < T * Cout * Cin;
    size_t o = 0;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const int ci = idx % Cin, r = idx / Cin;
        const int co = r % Cout, t = r / Cout;
        const size_t slab = (size_t)T * CoP * CiP;
        const float* p = part + ((size_t)t * CoP + co) * CiP + ci;
        o = ((size_t)co * Cin + ci) * T + t;
        int k = sg;
        for (; k + 3 * SG < S; k += 4 * SG) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += p[(size_t)(k + SG * q) * slab];
        }
        for (; k < S; k += SG) a[0] += p[(size_t)k * slab];
    }
    gsum[sg][lo] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    float dsum = 0.f;
    if (sg == 0 && live) {
        float val = gsum[0][lo];
#pragma unroll
        for (int q = 1; q < SG; ++q) val += gsum[q][lo];                   // fixed order
        val *= osc;
        dw[o] = val;
        if (sn_w) dsum = val * sn_w[o];
    }
    if (sn_w) {                                          // (slab groups > 0 contribute zeros)
        __shared__ float dred[4];
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_down(dsum, off, 64);
        if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = dsum;
        __syncthreads();
        if (threadIdx.x == 0) sn_dot[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
    }
}

// which reduction a layer gets, and with it the number of <dw, W_orig> partials
static bool wred_tiled(int Cin, int Cout) { return Cout * ((Cin + WRED_CI - 1) / WRED_CI) >= 512; }
// flat reduction: 32 outputs x 8 slab groups per block where 256 outputs per block would leave the grid under ~1.5 blocks per CU
static int wred_flat_out(int Cin, int Cout, int T) { return (T * Cout * Cin + 255) / 256 < 400 ? 32 : 256; }
static int wred_blocks(int Cin, int Cout, int T) {
    const int fo = wred_flat_out(Cin, Cout, T);
    return wred_tiled(Cin, Cout) ? Cout * ((Cin + WRED_CI - 1) / WRED_CI) : (T * Cout * Cin + fo - 1) / fo;
}

static int ilog2_floor_w(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }

struct WgradParams;
static int launch_wreduce(const WgradParams& p, int T, float* dw, float* dbias, const float* out_scale, const float* sn_w, float* sn_dot,
                          hipStream_t stream);

template <int KS, bool UPS, int PREC, int COB = 64>
static int launch_wgrad(WgradParams& p, float* dw, float* dbias, const float* out_scale, const float* sn_w, float* sn_dot, hipStream_t stream) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    constexpr int SA = 64 * 2 + 16, SD = COB * 2 + 16;
    // 128-pixel tiles, row-major inside the patch; TW >= 4 so that 4 consecutive k are 4 consecutive x
    int ltw = ilog2_floor_w(p.W); if (ltw > 4) ltw = 4;
    int lth = ilog2_floor_w(p.H); if (lth > 7 - ltw) lth = 7 - ltw;
    int lnb = 7 - ltw - lth;     // NOT capped by N: every one of the 128 k-slots must decode to a staged (zero-filled) halo pixel
    if (ltw < 2) return lp_set_error(LP_ERR_UNSUPPORTED, "wgrad needs W >= 4");
    p.lTH = lth; p.lTW = ltw; p.lNB = lnb;
    const int TH = 1 << lth, TW = 1 << ltw, NBv = 1 << lnb;
    p.tiles_x = (p.W + TW - 1) / TW; p.tiles_y = (p.H + TH - 1) / TH;
    p.num_tiles = p.tiles_x * p.tiles_y * ((p.N + NBv - 1) / NBv);
    if (p.splits > p.num_tiles) p.splits = p.num_tiles;
    int HH, HW;
    if (KS == 1) { HH = TH; HW = TW; } else if (UPS) { HH = TH / 2 + 2; HW = TW / 2 + 2; } else { HH = TH + 2; HW = TW + 2; }
    size_t lds = ((size_t)NBv * HH * HW * SA + 128 * SD) * (SPLIT ? 2 : 1);
    if (lds < (size_t)COB * 4 * 8 * sizeof(float)) lds = (size_t)COB * 4 * 8 * sizeof(float);      // bias-gradient reduction scratch
    if (lds > 160 * 1024) return lp_set_error(LP_ERR_UNSUPPORTED, "wgrad tile needs too much LDS");
    auto kern = conv_wgrad_kernel<KS, UPS, PREC, COB>;
    static thread_local int attr_dev = -1;                 // per host thread and device (main and autograd threads both launch)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipGetDevice failed");
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_dev = dev;
    }
    dim3 grid(p.splits, (p.CoP + COB - 1) / COB, p.diag ? 1 : p.CiP / 64);
    hipLaunchKernelGGL(kern, grid, dim3(COB * 4), lds, stream, p);
    int rc = lp_check_launch("conv_wgrad");
    if (rc) return rc;
    if (p.diag) return LP_OK;                              // the grouped reduction is launched by lp_gconv16_wgrad
    return launch_wreduce(p, KS * KS, dw, dbias, out_scale, sn_w, sn_dot, stream);
}


// ---- 1x1 weight gradient ("pixel contraction"): dw[co][ci] = sum_p dy[p][co] * a[p][ci] ------------------------------------------------
// The embedder's pointwise layers have one tap per staged tile, i.e. 1/9 of the MFMA work per staged byte of the 3x3 case: the
// load -> register -> LDS -> barrier -> MFMA cycle of conv_wgrad_kernel leaves them at ~0.9 TB/s of operand traffic (0.11 of HBM peak,
// profiles/r03_bench_f16.json roofline_wgrad1x1).  This kernel is the same contraction as a K-major GEMM built for that regime:
//   * a workgroup (4 waves, 2 x 2) owns a 128(co) x 128(ci) block of dw and a contiguous range of 64-pixel stages (split-K over pixels);
//     a wave keeps 64 x 64 (4 x 4 MFMA 16x16x32 tiles) -> 85 MFMA flops per staged byte against 43 for the 128 x 64 workgroup tile;
//   * both operand tiles [64 pixels][128 channels] go global -> LDS by DMA (16 B per lane, no registers), double buffered: the DMA of stage
//     s+1 is in flight while stage s multiplies, one barrier per stage;
//   * rows are 256 B = one LDS bank period, so the 32-byte units of a row are XOR-swizzled with (pixel & 7): the DMA lane that fills slot k'
//     of row r fetches chunk k' ^ ((r & 7) << 1); the transposing fragment reads (ds_read_b64_tr_b16: 4 pixels x 16 channels per 16-lane
//     group) then touch 8 distinct units per 8 pixels;
//   * XCD-aware block order: the 8 hardware queues take consecutive block ids, so block b = (j, x = b & 7) works on split (j / tiles) * 8 + x
//     and tile j % tiles -- all tiles that read the same pixel range run on the SAME XCD (one L2) next to each other in dispatch order, so
//     the operand re-reads (dy once per ci tile, a once per co tile) are L2 hits, not HBM reads.
// Output: the [split][CoP][CiP] fp32 slabs of the common reduction (wgrad_reduce_*: fp16 scale, spectral-norm dot, accumulation).
// Layers that want the bias gradient (column sums of dy) keep conv_wgrad_kernel: its staging passes dy through registers.
struct W1Params {
    const uint16_t* a_hi; const uint16_t* a_lo; const uint16_t* d_hi; const uint16_t* d_lo; float* part;
    long long P;
    int C8, Co8, CoP, CiP, splits, tiles_co, tiles_ci, nstage, xcd_map;
};
static __device__ __attribute__((aligned(64))) unsigned int w1_zero_page[16];      // what out-of-range DMA lanes read

// KP = pixels per stage: 64; 32 in the bf16x3 mode, whose doubled planes would otherwise need 128 KB of LDS -- ONE workgroup per CU for a kernel
// bound by its operand stream (round 4: two 64 KB workgroups per CU, profiles/r04_wgrad1x1_x3.txt; LP_W1_KP = 64 restores the old stage)
template <int PREC, int KP>
__global__ __launch_bounds__(256, 2) void wgrad1x1_kernel(W1Params p) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    constexpr int NPW = KP / 16;                            // 1 KiB DMA pieces (4 rows) per wave, operand plane and stage
    constexpr int TILE_B = KP * 256;                        // one [KP][128] 16-bit tile
    constexpr int STAGE_B = TILE_B * (SPLIT ? 4 : 2);       // D_hi | A_hi [| D_lo | A_lo]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mh = wave >> 1, nh = wave & 1;
    const int ntile = p.tiles_co * p.tiles_ci;
    int tile, split;
    if (p.xcd_map) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; tile = j % ntile; split = (j / ntile) * 8 + x; }
    else { tile = blockIdx.x % ntile; split = blockIdx.x / ntile; }
    const int co0 = (tile / p.tiles_ci) * 128, ci0 = (tile % p.tiles_ci) * 128;
    const int per = (p.nstage + p.splits - 1) / p.splits;
    const int s_beg = split * per, s_end = min(p.nstage, s_beg + per);

    // DMA descriptors: piece q = i*4 + wave covers rows 4q .. 4q+3 of a tile; lane -> row 4q + lane/16, slot lane%16
    int d_rel[NPW], a_rel[NPW], row_[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int row = (i * 4 + wave) * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ ((row & 7) << 1);
        const int dch = co0 + chunk * 8, ach = ci0 + chunk * 8;
        row_[i] = row;
        d_rel[i] = dch < p.Co8 ? row * p.Co8 + dch : -1;
        a_rel[i] = ach < p.C8 ? row * p.C8 + ach : -1;
    }
    const uint16_t* zero16 = (const uint16_t*)w1_zero_page;
    auto issue = [&](int stage, int buf) {
        const long long pix0 = (long long)stage * KP;
        const unsigned dst = (unsigned)(uintptr_t)(smem + buf * STAGE_B);
        const size_t dbase = (size_t)pix0 * p.Co8, abase = (size_t)pix0 * p.C8;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const bool pok = pix0 + row_[i] < p.P;
            const bool dok = pok && d_rel[i] >= 0, aok = pok && a_rel[i] >= 0;
            const unsigned o = dst + (unsigned)((i * 4 + wave) * 1024);
            lp_glds16(dok ? p.d_hi + dbase + d_rel[i] : zero16, o);
            lp_glds16(aok ? p.a_hi + abase + a_rel[i] : zero16, o + TILE_B);
            if (SPLIT) {
                lp_glds16(dok ? p.d_lo + dbase + d_rel[i] : zero16, o + 2 * TILE_B);
                lp_glds16(aok ? p.a_lo + abase + a_rel[i] : zero16, o + 3 * TILE_B);
            }
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // source-lane role of the transposing reads (see the header of this file): pixel G*4 + sj of the 16-pixel half, channels 4*sq .. 4*sq+3
    const int G = lane >> 4, sj = (lane & 15) >> 2, sq = lane & 3;
    const int kpl = G * 4 + sj;                            // 0..15; the four halves of a stage are kpl + 16 h
    const int sw = (kpl & 7) << 1;                         // (pixel & 7) is the same for all halves
    int d_off[4], a_off[4];                                // byte offset inside a row of this lane's 8-byte piece, per fragment
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        d_off[f] = (((mh * 8 + f * 2 + (sq >> 1)) ^ sw) << 4) + (sq & 1) * 8;
        a_off[f] = (((nh * 8 + f * 2 + (sq >> 1)) ^ sw) << 4) + (sq & 1) * 8;
    }
    auto compute = [&](int buf) {
        const unsigned char* D_hi = smem + buf * STAGE_B;
        const unsigned char* A_hi = D_hi + TILE_B;
        s16x8_t fa[2][4], fb[2][4], fal[2][4], fbl[2][4];
        auto fetch = [&](int ks, int set) {
            const int r0 = (ks * 32 + kpl) * 256, r1 = r0 + 16 * 256;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const s16x4_t v0 = tr_read(D_hi + r0 + d_off[f]), v1 = tr_read(D_hi + r1 + d_off[f]);
                fa[set][f] = (s16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const s16x4_t w0 = tr_read(A_hi + r0 + a_off[f]), w1 = tr_read(A_hi + r1 + a_off[f]);
                fb[set][f] = (s16x8_t){w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                if (SPLIT) {
                    const s16x4_t x0 = tr_read(D_hi + 2 * TILE_B + r0 + d_off[f]), x1 = tr_read(D_hi + 2 * TILE_B + r1 + d_off[f]);
                    fal[set][f] = (s16x8_t){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    const s16x4_t y0 = tr_read(A_hi + 2 * TILE_B + r0 + a_off[f]), y1 = tr_read(A_hi + 2 * TILE_B + r1 + a_off[f]);
                    fbl[set][f] = (s16x8_t){y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
                }
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int ks = 0; ks < KP / 32; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < KP / 32) fetch(ks + 1, cur ^ 1);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    if (SPLIT) {
                        acc[mf][nf] = mfma16(fal[cur][mf], fb[cur][nf], acc[mf][nf]);
                        acc[mf][nf] = mfma16(fa[cur][mf], fbl[cur][nf], acc[mf][nf]);
                    }
                    acc[mf][nf] = mfma16t<F16>(fa[cur][mf], fb[cur][nf], acc[mf][nf]);
                }
        }
    };

    if (s_beg < s_end) issue(s_beg, 0);
    for (int s = s_beg; s < s_end; ++s) {
        const int buf = (s - s_beg) & 1;
        lp_wait_vm0();
        __syncthreads();                    // stage s has landed for every wave; everyone is done with the other buffer
        if (s + 1 < s_end) issue(s + 1, buf ^ 1);
        compute(buf);
    }
    // C layout of the 16x16 MFMA: row (co) = (lane >> 4) * 4 + r, column (ci) = lane & 15
    float* slab = p.part + (size_t)split * p.CoP * p.CiP;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + mh * 64 + mf * 16 + (lane >> 4) * 4 + r;
                const int ci = ci0 + nh * 64 + nf * 16 + (lane & 15);
                if (co < p.CoP && ci < p.CiP) slab[(size_t)co * p.CiP + ci] = acc[mf][nf][r];
            }
}

static int launch_wreduce(const WgradParams& p, int T, float* dw, float* dbias, const float* out_scale, const float* sn_w, float* sn_dot,
                          hipStream_t stream) {
    const int bblocks = p.bpart ? (p.Cout + 63) / 64 : 0;
    const int wblocks = wred_blocks(p.Cin, p.Cout, T);
    if (wred_tiled(p.Cin, p.Cout))
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wblocks + bblocks), dim3(256), 0, stream, p.part, dw, p.splits, T, p.Cout, p.Cin, p.CoP, p.CiP,
                           p.bpart, dbias, wblocks, out_scale, sn_w, sn_dot, p.db_acc);
    else if (wred_flat_out(p.Cin, p.Cout, T) == 32)
        hipLaunchKernelGGL(wgrad_reduce_flat_kernel<32>, dim3(wblocks + bblocks), dim3(256), 0, stream, p.part, dw, p.splits, T, p.Cout, p.Cin, p.CoP,
                           p.CiP, p.bpart, dbias, wblocks, out_scale, sn_w, sn_dot, p.db_acc);
    else
        hipLaunchKernelGGL(wgrad_reduce_flat_kernel<256>, dim3(wblocks + bblocks), dim3(256), 0, stream, p.part, dw, p.splits, T, p.Cout, p.Cin, p.CoP,
                           p.CiP, p.bpart, dbias, wblocks, out_scale, sn_w, sn_dot, p.db_acc);
    return lp_check_launch("wgrad_reduce");
}

template <int PREC, int KP>
static int launch_wgrad1x1(WgradParams& p, float* dw, const float* out_scale, const float* sn_w, float* sn_dot, hipStream_t stream) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    W1Params q;
    q.a_hi = p.a_hi; q.a_lo = p.a_lo; q.d_hi = p.d_hi; q.d_lo = p.d_lo; q.part = p.part;
    q.P = (long long)p.N * p.H * p.W;
    q.C8 = p.C8; q.Co8 = p.Co8; q.CoP = p.CoP; q.CiP = p.CiP;
    q.tiles_co = (p.CoP + 127) / 128; q.tiles_ci = (p.CiP + 127) / 128;
    q.nstage = (int)((q.P + KP - 1) / KP);
    int splits = p.splits < q.nstage ? p.splits : q.nstage;
    const int per = (q.nstage + splits - 1) / splits;
    splits = (q.nstage + per - 1) / per;                   // no empty split: every slab the reduction sums is written
    p.splits = q.splits = splits;
    q.xcd_map = (splits % 8 == 0);
    const size_t lds = (size_t)2 * KP * 256 * (SPLIT ? 4 : 2);
    auto kern = wgrad1x1_kernel<PREC, KP>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipGetDevice failed");
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_dev = dev;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(q.tiles_co * q.tiles_ci * splits)), dim3(256), lds, stream, q);
    int rc = lp_check_launch("wgrad1x1");
    if (rc) return rc;
    return launch_wreduce(p, 1, dw, nullptr, out_scale, sn_w, sn_dot, stream);
}


// ---- 3x3 weight gradient, DMA-staged (round 4) ------------------------------------------------------------------------------------------
// conv_wgrad_kernel stages both operands load -> register -> LDS -> barrier with nothing in flight while it multiplies, and a wave's
// 32(co) x 32(ci) x 9-tap block costs 40 transposing LDS reads per 36 MFMAs: 0.117 of the MFMA peak over the step's 3x3 layers for three
// rounds.  This kernel is the same contraction on the wgrad1x1_kernel template:
//   * a workgroup owns 64(co) x 64(ci) x 9 taps and a contiguous range of 8 x 16-pixel tiles (split-K over pixels); a wave owns ALL 64 output
//     channels x 16 input channels x 9 taps (36 accumulator tiles): the 4 dy fragments of a k-step are read once and used by all 9 taps,
//     each tap needs ONE input fragment -> 26 transposing reads per 36 MFMAs;
//   * the dy tile [128 pixels][64 co] and the input halo [10 x 18 (6 x 10 low-resolution with the fused x2 upsampling) pixels][64 ci] go
//     global -> LDS by DMA (1 KiB pieces of 8 pixel rows x 128 B), double buffered: the DMA of tile t+1 is in flight while tile t multiplies,
//     one barrier per tile; two workgroups per CU (78 KB of LDS, <= 256 registers each) overlap each other's waits;
//   * 128-byte rows: the 32-byte units of row r are XOR-swizzled with (r >> 1) & 3 -- with the row's parity that spreads any 8 CONSECUTIVE
//     rows over the 8 units of the 256-byte bank period, and the 8 pixels a half-wave's transposing read touches are consecutive halo rows for
//     every tap shift (halo rows are 18 | 10 = 2 mod 8 pixels apart: the key of pixel (row, col) is (row + (col >> 1)) & 3);
//   * the bias gradient (column sums of dy) comes from the matrix core as well: dy x ones, one extra MFMA per k-step and wave;
//   * XCD-aware block order as wgrad1x1_kernel: the (co, ci) tiles of one pixel range run on one XCD.
// Output: the [split][tap][CoP][CiP] slabs of the common reduction.  f16 / bf16 operands; bf16x3 since round 6 (one workgroup per CU: the doubled
// planes fill the LDS; 512 registers per lane, accumulators in AGPRs).
struct W3Params {
    const uint16_t* a_hi; const uint16_t* a_lo; const uint16_t* d_hi; const uint16_t* d_lo; float* part; float* bpart;
    int N, H, W, Hin, Win, C8, Co8, CoP, CiP;
    int splits, per, num_tiles, tiles_x, tiles_y, tiles_co, tiles_ci, xcd_map, diag;
};

// DIAGB (grouped convs, block-diagonal form: the workgroup of output channels [co0, co0 + 64) only meets input channels [co0, co0 + 64)):
//   0: dense 64 x 64 block (every wave all four 16-row fragments) | 16: groups of <= 16 channels -- only the 16 x 16 diagonal tiles are
//   non-zero, a wave keeps ONE row fragment (its own: 9 accumulator tiles, 1/4 of the matrix work and of the slab bytes) | 32: groups of 32
//   -- the two row fragments of the wave's 32-channel half.  The bf16x3 mode (hi + lo planes, 3 MFMAs) is built for the diagonal forms only:
//   their 9 / 18 accumulator tiles leave the registers for the second fragment set, and ONE workgroup per CU holds the doubled planes.
template <bool UPS, int PREC, int DIAGB>
__global__ __launch_bounds__(256, (PREC == LP_PREC_BF16X3 && DIAGB == 0) ? 1 : 2) void wgrad3_pipe_kernel(W3Params p) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    // (round 6) dense bf16x3: the doubled planes hold a CU to ONE workgroup anyway (156 KB of LDS), so the kernel is built for one wave per SIMD --
    // 512 registers: the 36 accumulator tiles go to AGPRs and the second fragment set fits without scratch
    static_assert(!UPS || DIAGB == 0, "grouped convs are not upsampled");
    constexpr int NT = DIAGB == 16 ? 1 : DIAGB == 32 ? 2 : 4;     // row (co) fragments per wave
    constexpr int HH = UPS ? 6 : 10, HW = UPS ? 10 : 18;
    constexpr int HALO_PX = HH * HW;
    constexpr int NAH = (HALO_PX + 7) / 8;                  // 1 KiB DMA pieces of the halo: 23 | 8
    constexpr int NAW = (NAH + 3) / 4;                      // per wave
    constexpr int HALO_B = NAH * 1024, DY_B = 128 * 128;
    constexpr int NP = SPLIT ? 2 : 1;                       // planes: stage = [halo hi][halo lo][dy hi][dy lo]
    constexpr int DY_O = HALO_B * NP, STAGE_B = (HALO_B + DY_B) * NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = p.tiles_co * p.tiles_ci;
    int tile, split;
    if (p.xcd_map) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; tile = j % ntile; split = (j / ntile) * 8 + x; }
    else { tile = blockIdx.x % ntile; split = blockIdx.x / ntile; }
    const int tco = tile / p.tiles_ci, tci = tile - tco * p.tiles_ci;
    const int co0 = tco * 64, ci0 = p.diag ? co0 : tci * 64;
    const int t_beg = split * p.per, t_end = min(p.num_tiles, t_beg + p.per);
    const uint16_t* zero16 = (const uint16_t*)w1_zero_page;

    // DMA piece q = i*4 + wave covers LDS rows 8q .. 8q+7 (a row = one pixel x 64 channels); lane -> row 8q + lane/8, 16-byte slot lane%8,
    // which holds the channel chunk slot ^ (((row >> 1) & 3) << 1)
    auto issue = [&](int t, int buf) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                        // re-derive the lane's piece coordinates per tile: hoisted out of the loop they cost
        const int lrow = ln >> 3, lslot = ln & 7;           // ~20 registers that the 144 + 48 accumulator / fragment registers do not leave
        int r = t;
        const int tx = r % p.tiles_x; r /= p.tiles_x;
        const int ty = r % p.tiles_y; const int n0 = r / p.tiles_y;
        const int y0 = ty * 8, x0 = tx * 16;
        const int oy = UPS ? (y0 >> 1) - 1 : y0 - 1, ox = UPS ? (x0 >> 1) - 1 : x0 - 1;
        const unsigned dst = (unsigned)(uintptr_t)(smem + buf * STAGE_B);
#pragma unroll
        for (int i = 0; i < NAW; ++i) {
            const int q = i * 4 + wave;
            if (q < NAH) {                                  // (wave-uniform)
                const int hp = q * 8 + lrow;
                const int hy = hp / HW, hx = hp - hy * HW;
                const int ch = ci0 + ((lslot ^ (((hp >> 1) & 3) << 1)) << 3);
                const int iy = oy + hy, ix = ox + hx;
                const bool ok = (hp < HALO_PX) && (ch < p.C8) && (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
                const size_t off = ok ? ((size_t)((n0 * p.Hin + iy) * p.Win + ix) * p.C8 + ch) : 0;
                lp_glds16(ok ? p.a_hi + off : zero16, dst + (unsigned)(q * 1024));
                if (SPLIT) lp_glds16(ok ? p.a_lo + off : zero16, dst + (unsigned)(HALO_B + q * 1024));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * 4 + wave;
            const int kp = q * 8 + lrow;
            const int ch = co0 + ((lslot ^ (((kp >> 1) & 3) << 1)) << 3);
            const int yy = y0 + (kp >> 4), xx = x0 + (kp & 15);
            const bool ok = (ch < p.Co8) && (yy < p.H) && (xx < p.W);
            const size_t off = ok ? ((size_t)((n0 * p.H + yy) * p.W + xx) * p.Co8 + ch) : 0;
            lp_glds16(ok ? p.d_hi + off : zero16, dst + (unsigned)(DY_O + q * 1024));
            if (SPLIT) lp_glds16(ok ? p.d_lo + off : zero16, dst + (unsigned)(DY_O + DY_B + q * 1024));
        }
    };

    f32x4_t acc[9][NT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[t][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t accb = (f32x4_t){0.f, 0.f, 0.f, 0.f};           // wave w: column sums of dy for co0 + 16 w .. + 15 (every column of the tile is the sum)
    const bool do_bias = (DIAGB == 0) && (p.bpart != nullptr) && (tci == 0);
    const short one16 = F16 ? (short)0x3C00 : (short)0x3F80;
    const s16x8_t ones = (s16x8_t){one16, one16, one16, one16, one16, one16, one16, one16};

    // source-lane role of the transposing reads (header of this file): pixel column pxl of the 16-pixel patch row, channels 4 sq .. 4 sq + 3
    const int G = lane >> 4, sj = (lane & 15) >> 2, sq = lane & 3;
    const int pxl = G * 4 + sj;
    const int swA = (pxl >> 1) & 3;                         // key of dy row 16 m + pxl
    const int mf0 = DIAGB == 16 ? wave : DIAGB == 32 ? (wave >> 1) * 2 : 0;      // first row fragment of this wave
    int aoff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) aoff[j] = pxl * 128 + (((mf0 + j) ^ swA) << 5) + sq * 8;
    int bl[3], bs[3];                                      // per tap column dx: byte offset of the lane's halo column, key part of that column
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int col = UPS ? ((pxl + dx - 1) >> 1) + 1 : pxl + dx;
        bl[dx] = col * 128 + sq * 8; bs[dx] = col >> 1;
    }

    auto compute = [&](int buf) {
        const unsigned char* Hb = smem + buf * STAGE_B;
        const unsigned char* Db = Hb + DY_O;
        s16x8_t fa[2][NT], fb[3], fal[2][NT], fbl[3];
        auto fetch_a = [&](int ks, int set) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const s16x4_t v0 = tr_read(Db + (ks * 32) * 128 + aoff[j]), v1 = tr_read(Db + (ks * 32 + 16) * 128 + aoff[j]);
                fa[set][j] = (s16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (SPLIT) {
                    const s16x4_t w0 = tr_read(Db + DY_B + (ks * 32) * 128 + aoff[j]), w1 = tr_read(Db + DY_B + (ks * 32 + 16) * 128 + aoff[j]);
                    fal[set][j] = (s16x8_t){w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                }
            }
        };
        auto fetch_b = [&](int g, int set) {
            const int ks = g / 9, tap = g - ks * 9;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int r0 = UPS ? ((2 * ks + dy - 1) >> 1) + 1 : 2 * ks + dy;
            const int r1 = UPS ? ((2 * ks + dy) >> 1) + 1 : 2 * ks + dy + 1;
            const int o0 = r0 * HW * 128 + bl[dx] + ((wave ^ ((bs[dx] + r0) & 3)) << 5);
            const int o1 = r1 * HW * 128 + bl[dx] + ((wave ^ ((bs[dx] + r1) & 3)) << 5);
            const s16x4_t v0 = tr_read(Hb + o0), v1 = tr_read(Hb + o1);
            fb[set] = (s16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (SPLIT) {
                const s16x4_t w0 = tr_read(Hb + HALO_B + o0), w1 = tr_read(Hb + HALO_B + o1);
                fbl[set] = (s16x8_t){w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            }
        };
        fetch_a(0, 0); fetch_b(0, 0); fetch_b(1, 1);
#pragma unroll
        for (int g = 0; g < 36; ++g) {
            const int ks = g / 9, tap = g - ks * 9;
            if (g + 2 < 36) fetch_b(g + 2, (g + 2) % 3);
            if (tap == 4 && ks + 1 < 4) fetch_a(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (SPLIT) {
                    acc[tap][j] = mfma16(fal[ks & 1][j], fb[g % 3], acc[tap][j]);
                    acc[tap][j] = mfma16(fa[ks & 1][j], fbl[g % 3], acc[tap][j]);
                }
                acc[tap][j] = mfma16t<F16>(fa[ks & 1][j], fb[g % 3], acc[tap][j]);
            }
            if constexpr (DIAGB == 0) {
                if (tap == 0 && do_bias) {                  // (scalar selects: no dynamic register indexing)
                    const s16x8_t fw = wave == 0 ? fa[ks & 1][0] : wave == 1 ? fa[ks & 1][1] : wave == 2 ? fa[ks & 1][2] : fa[ks & 1][3];
                    accb = mfma16t<F16>(fw, ones, accb);
                    if (SPLIT) {          // (dense bf16x3: dy = hi + lo)
                        const s16x8_t fwl = wave == 0 ? fal[ks & 1][0] : wave == 1 ? fal[ks & 1][1] : wave == 2 ? fal[ks & 1][2] : fal[ks & 1][3];
                        accb = mfma16(fwl, ones, accb);
                    }
                }
            }
        }
    };

    if (t_beg < t_end) issue(t_beg, 0);
    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        lp_wait_vm0();
        __syncthreads();                    // tile t has landed for every wave; everyone is done with the other buffer
        if (t + 1 < t_end) issue(t + 1, buf ^ 1);
        compute(buf);
    }

    // C layout of the 16x16 MFMA: row (co) = (lane >> 4) * 4 + r, column (ci) = lane & 15
    const int cib = (p.diag ? 0 : ci0) + wave * 16 + (lane & 15);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (mf0 + j) * 16 + (lane >> 4) * 4 + r;
                p.part[(((size_t)split * 9 + tap) * p.CoP + co) * p.CiP + cib] = acc[tap][j][r];
            }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p.bpart[(size_t)split * p.CoP + co0 + wave * 16 + (lane >> 4) * 4 + r] = accb[r];
    }
}

template <bool UPS, int PREC, int DIAGB = 0>
static int launch_wgrad3_pipe(WgradParams& p, float* dw, float* dbias, const float* out_scale, const float* sn_w, float* sn_dot, hipStream_t stream) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    static const int target_env = getenv("LP_WGRAD3_WGS") ? atoi(getenv("LP_WGRAD3_WGS")) : 0;
    W3Params q;
    q.a_hi = p.a_hi; q.a_lo = p.a_lo; q.d_hi = p.d_hi; q.d_lo = p.d_lo; q.part = p.part; q.bpart = p.bpart;
    q.N = p.N; q.H = p.H; q.W = p.W; q.Hin = p.Hin; q.Win = p.Win; q.C8 = p.C8; q.Co8 = p.Co8; q.CoP = p.CoP; q.CiP = p.CiP;
    q.diag = p.diag;
    q.tiles_x = (p.W + 15) / 16; q.tiles_y = (p.H + 7) / 8;
    q.num_tiles = q.tiles_x * q.tiles_y * p.N;
    q.tiles_co = p.CoP / 64; q.tiles_ci = p.diag ? 1 : p.CiP / 64;
    const int ntile = q.tiles_co * q.tiles_ci;
    const int target = target_env > 0 ? target_env : (SPLIT ? 512 : 1024);    // ~2 resident sets of two (bf16x3: one) workgroups per CU
    int splits = target / ntile; if (splits < 1) splits = 1;
    if (splits > p.splits) splits = p.splits;
    if (splits > q.num_tiles) splits = q.num_tiles;
    if (splits >= 8) splits = splits / 8 * 8;
    q.per = (q.num_tiles + splits - 1) / splits;
    splits = (q.num_tiles + q.per - 1) / q.per;                            // no empty split: every slab the reduction sums is written
    p.splits = q.splits = splits;
    q.xcd_map = (splits % 8 == 0);
    constexpr int HALO_PX = UPS ? 60 : 180;
    const size_t lds = (size_t)2 * (((HALO_PX + 7) / 8) * 1024 + 128 * 128) * (SPLIT ? 2 : 1);
    auto kern = wgrad3_pipe_kernel<UPS, PREC, DIAGB>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipGetDevice failed");
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_dev = dev;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntile * splits)), dim3(256), lds, stream, q);
    int rc = lp_check_launch("wgrad3_pipe");
    if (rc) return rc;
    if (p.diag) return LP_OK;                              // the grouped reduction is launched by lp_gconv16_wgrad
    return launch_wreduce(p, 9, dw, dbias, out_scale, sn_w, sn_dot, stream);
}

// LP_WGRAD3_PIPE = 0: conv_wgrad_kernel everywhere | 1 (default): the DMA-staged kernel for layers with >= LP_WGRAD3_MIN_TILES (32) pixel
// tiles and W >= 16 | 2: for every 3x3 layer of a one-plane mode (tests)
template <int PREC>
static bool wgrad3_pipe_wanted(const WgradParams& p, bool diag_form = false) {
    static const int x3 = getenv("LP_WGRAD3_X3") ? atoi(getenv("LP_WGRAD3_X3")) : 1;          // dense bf16x3 layers on this kernel (round 6: -4 .. -9 % per layer, profiles/r06_wgrad3_x3.txt; 0: conv_wgrad_kernel)
    if (PREC == LP_PREC_BF16X3 && !diag_form && !x3) return false;
    static const int mode = getenv("LP_WGRAD3_PIPE") ? atoi(getenv("LP_WGRAD3_PIPE")) : 1;
    static const int min_tiles = getenv("LP_WGRAD3_MIN_TILES") ? atoi(getenv("LP_WGRAD3_MIN_TILES")) : 32;
    if (mode == 0) return false;
    if (mode == 2) return true;
    return p.W >= 16 && p.H >= 8 && ((p.W + 15) / 16) * ((p.H + 7) / 8) * p.N >= min_tiles;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int lp_conv_wgrad_dot_blocks(int Cin, int Cout, int ksize) { return wred_blocks(Cin, Cout, ksize * ksize); }

extern "C" long long lp_conv_wgrad_workspace_bytes(int Cin, int Cout, int ksize, int splits) {
    // [splits][taps][CoP][CiP] weight-gradient slabs, then [splits][CoP] bias-gradient partials
    return (long long)splits * ksize * ksize * round_up(Cout, 64) * round_up(Cin, 64) * 4 + (long long)splits * round_up(Cout, 64) * 4;
}

template <int PREC>
static int dispatch_wgrad(WgradParams& p, float* dw, float* dbias, const float* out_scale, const float* sn_w, float* sn_dot, int ksize,
                          int upsample, hipStream_t s) {
    // 128 output channels per workgroup (8 waves) where the layer is wide enough; LP_WGRAD_COB = 64 | 128 overrides
    static const int cob_env = getenv("LP_WGRAD_COB") ? atoi(getenv("LP_WGRAD_COB")) : 0;
    const bool cob128 = cob_env ? (cob_env == 128) : true;      // measured: -9 % (bf16x3), -6 % (bf16) on the 64..512-channel layers
    if (ksize == 3 && wgrad3_pipe_wanted<PREC>(p))
        return upsample ? launch_wgrad3_pipe<true, PREC>(p, dw, dbias, out_scale, sn_w, sn_dot, s) : launch_wgrad3_pipe<false, PREC>(p, dw, dbias, out_scale, sn_w, sn_dot, s);
    if (cob128 && p.Cout >= 128 && ksize == 3)
        return upsample ? launch_wgrad<3, true, PREC, 128>(p, dw, dbias, out_scale, sn_w, sn_dot, s) : launch_wgrad<3, false, PREC, 128>(p, dw, dbias, out_scale, sn_w, sn_dot, s);
    static const bool w1_old = getenv("LP_WGRAD1X1_OLD") != nullptr;                              // A/B knob: the generic kernel for 1x1 layers
    static const int w1_kp = getenv("LP_W1_KP") ? atoi(getenv("LP_W1_KP")) : 0;                   // bf16x3 stage: 32 (default) | 64 pixels
    if (ksize == 1 && !upsample && !p.bpart && !w1_old) {
        const int kp = w1_kp ? w1_kp : (PREC == LP_PREC_BF16X3 ? 32 : 64);
        if (kp == 32) return launch_wgrad1x1<PREC, 32>(p, dw, out_scale, sn_w, sn_dot, s);
        return launch_wgrad1x1<PREC, 64>(p, dw, out_scale, sn_w, sn_dot, s);
    }
    static const int cob1_env = getenv("LP_WGRAD_COB1") ? atoi(getenv("LP_WGRAD_COB1")) : 0;        // 1x1 layers: 64 | 128 forces
    if ((cob1_env ? cob1_env == 128 : true) && p.Cout >= 128 && ksize == 1 && !upsample)
        return launch_wgrad<1, false, PREC, 128>(p, dw, dbias, out_scale, sn_w, sn_dot, s);
    if (ksize == 3 && !upsample) return launch_wgrad<3, false, PREC>(p, dw, dbias, out_scale, sn_w, sn_dot, s);
    if (ksize == 3 && upsample) return launch_wgrad<3, true, PREC>(p, dw, dbias, out_scale, sn_w, sn_dot, s);
    if (ksize == 1 && !upsample) return launch_wgrad<1, false, PREC>(p, dw, dbias, out_scale, sn_w, sn_dot, s);
    return lp_set_error(LP_ERR_UNSUPPORTED, "lp_conv16_wgrad: unsupported configuration");
}

extern "C" int lp_conv16_wgrad(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* dy_hi, const uint16_t* dy_lo, float* dw,
                               float* workspace, int N, int H, int W, int Cin, int Cout, int ksize, int upsample, int splits, int prec,
                               float* dbias, int dbias_accumulate, const float* out_scale, const float* sn_w_orig, float* sn_dot,
                               void* stream) {
    if (!a_hi || !dy_hi || !dw || !workspace) return lp_set_error(LP_ERR_ARG, "lp_conv16_wgrad: null pointer");
    if (!sn_w_orig != !sn_dot) return lp_set_error(LP_ERR_ARG, "lp_conv16_wgrad: sn_w_orig and sn_dot go together");
    if (prec == LP_PREC_BF16X3 && (!a_lo || !dy_lo)) return lp_set_error(LP_ERR_ARG, "lp_conv16_wgrad: bf16x3 needs the lo planes");
    if (splits < 1) return lp_set_error(LP_ERR_ARG, "lp_conv16_wgrad: splits must be >= 1");
    if (upsample && ((H | W) & 1)) return lp_set_error(LP_ERR_ARG, "lp_conv16_wgrad: upsampled dims must be even");
    WgradParams p;
    p.a_hi = a_hi; p.a_lo = a_lo; p.d_hi = dy_hi; p.d_lo = dy_lo; p.part = workspace;
    p.N = N; p.H = H; p.W = W; p.Hin = upsample ? H / 2 : H; p.Win = upsample ? W / 2 : W;
    p.Cin = Cin; p.Cout = Cout; p.C8 = (Cin + 7) & ~7; p.Co8 = (Cout + 7) & ~7; p.CoP = round_up(Cout, 64); p.CiP = round_up(Cin, 64);
    p.splits = splits; p.db_acc = dbias_accumulate; p.diag = 0;
    p.bpart = dbias ? workspace + (size_t)splits * ksize * ksize * p.CoP * p.CiP : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (prec == LP_PREC_BF16) return dispatch_wgrad<LP_PREC_BF16>(p, dw, dbias, out_scale, sn_w_orig, sn_dot, ksize, upsample, s);
    if (prec == LP_PREC_BF16X3) return dispatch_wgrad<LP_PREC_BF16X3>(p, dw, dbias, out_scale, sn_w_orig, sn_dot, ksize, upsample, s);
    if (prec == LP_PREC_F16) return dispatch_wgrad<LP_PREC_F16>(p, dw, dbias, out_scale, sn_w_orig, sn_dot, ksize, upsample, s);
    return lp_set_error(LP_ERR_ARG, "lp_conv16_wgrad: unknown precision");
}

// ---- grouped 3x3 conv (ResNeXt conv2): weight gradient of the block-diagonal formulation (see lp_gconv16_fwd) --------------------
// conv_wgrad_kernel with diag = 1 leaves slabs [split][9][C][64] (row = output channel, column = input channel inside the row's aligned
// 64-channel block); the reduction keeps the group's own columns: dw[co][j][t] = osc * sum_s slab[s][t][co][(co % 64) / cg * cg + j].
// (round 6) 32 columns x 8 split lanes per workgroup: lane r sums splits r, r + 8, ... on four accumulators, the lanes are folded in a fixed order
// through LDS -- one thread per column walked all S slabs (18 workgroups of dependent L2 loads for the layer-1 shapes: 20 us per layer)
__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int C, int cg,
                                                                  const float* __restrict__ out_scale) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, lr = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + col;                          // (co, t, j), j fastest: neighbouring lanes read neighbouring columns
    const bool live = idx < C * 9 * cg;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int j = 0, t = 0, co = 0;
    if (live) {
        j = idx % cg; const int r = idx / cg;
        t = r % 9; co = r / 9;
        const size_t slab = (size_t)9 * C * 64;
        const float* p = part + ((size_t)t * C + co) * 64 + ((co & 63) / cg) * cg + j;
        int k = lr;
        for (; k + 24 < S; k += 32) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += p[(size_t)(k + 8 * q) * slab];
        }
        for (; k < S; k += 8) a[0] += p[(size_t)k * slab];
    }
    red[lr][col] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (lr == 0 && live) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) sum += red[r][col];
        dw[((size_t)co * cg + j) * 9 + t] = sum * (out_scale ? out_scale[0] : 1.f);
    }
}

extern "C" long long lp_gconv_wgrad_workspace_bytes(int C, int splits) { return (long long)splits * 9 * C * 64 * 4; }

extern "C" int lp_gconv16_wgrad(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* dy_hi, const uint16_t* dy_lo, float* dw,
                                float* workspace, int N, int H, int W, int C, int group_size, int splits, int prec,
                                const float* out_scale, void* stream) {
    if (!a_hi || !dy_hi || !dw || !workspace) return lp_set_error(LP_ERR_ARG, "lp_gconv16_wgrad: null pointer");
    if (prec == LP_PREC_BF16X3 && (!a_lo || !dy_lo)) return lp_set_error(LP_ERR_ARG, "lp_gconv16_wgrad: bf16x3 needs the lo planes");
    if ((C & 63) || group_size < 1 || 64 % group_size || splits < 1) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_gconv16_wgrad: C % 64 == 0, group size dividing 64");
    WgradParams p;
    p.a_hi = a_hi; p.a_lo = a_lo; p.d_hi = dy_hi; p.d_lo = dy_lo; p.part = workspace;
    p.N = N; p.H = H; p.W = W; p.Hin = H; p.Win = W;
    p.Cin = C; p.Cout = C; p.C8 = C; p.Co8 = C; p.CoP = C; p.CiP = 64;
    p.splits = splits; p.db_acc = 0; p.diag = 1; p.bpart = nullptr;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // groups of <= 16 / 32 channels: the DMA-staged kernel on the non-zero diagonal tiles only (LP_GWGRAD_DIAG=0: the dense 64 x 64 block)
    static const bool diag_env = !(getenv("LP_GWGRAD_DIAG") && atoi(getenv("LP_GWGRAD_DIAG")) == 0);
    const int db = !diag_env ? 0 : (group_size <= 16 && 16 % group_size == 0) ? 16 : (group_size == 32 ? 32 : 0);
#define LP_GW3(PR) (db == 16 ? launch_wgrad3_pipe<false, PR, 16>(p, dw, nullptr, out_scale, nullptr, nullptr, s) \
                             : launch_wgrad3_pipe<false, PR, 32>(p, dw, nullptr, out_scale, nullptr, nullptr, s))
    if (db && prec == LP_PREC_BF16 && wgrad3_pipe_wanted<LP_PREC_BF16>(p, true)) rc = LP_GW3(LP_PREC_BF16);
    else if (db && prec == LP_PREC_F16 && wgrad3_pipe_wanted<LP_PREC_F16>(p, true)) rc = LP_GW3(LP_PREC_F16);
    else if (db && prec == LP_PREC_BF16X3 && wgrad3_pipe_wanted<LP_PREC_BF16X3>(p, true)) rc = LP_GW3(LP_PREC_BF16X3);
#undef LP_GW3
    else if (prec == LP_PREC_BF16 && wgrad3_pipe_wanted<LP_PREC_BF16>(p)) rc = launch_wgrad3_pipe<false, LP_PREC_BF16>(p, dw, nullptr, out_scale, nullptr, nullptr, s);
    else if (prec == LP_PREC_F16 && wgrad3_pipe_wanted<LP_PREC_F16>(p)) rc = launch_wgrad3_pipe<false, LP_PREC_F16>(p, dw, nullptr, out_scale, nullptr, nullptr, s);
    else if (prec == LP_PREC_BF16) rc = launch_wgrad<3, false, LP_PREC_BF16>(p, dw, nullptr, out_scale, nullptr, nullptr, s);
    else if (prec == LP_PREC_BF16X3) rc = launch_wgrad<3, false, LP_PREC_BF16X3>(p, dw, nullptr, out_scale, nullptr, nullptr, s);
    else if (prec == LP_PREC_F16) rc = launch_wgrad<3, false, LP_PREC_F16>(p, dw, nullptr, out_scale, nullptr, nullptr, s);
    else return lp_set_error(LP_ERR_ARG, "lp_gconv16_wgrad: unknown precision");
    if (rc) return rc;
    const int total = C * 9 * group_size;
    hipLaunchKernelGGL(gconv_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, s, workspace, dw, p.splits, C, group_size, out_scale);
    return lp_check_launch("gconv_wgrad_reduce");
}
