// Implicit-GEMM convolution for gfx950 whose two operands reach LDS by LDS-DMA only (no VGPR staging, no VALU work in the loop):
//
//   y[n,oy,ox,co] = alpha * alpha2 * sum_{tap,ci} a[n, oy+dy, ox+dx, ci] * w[tap][co][ci]  (+ bias[co]) (+ res[n,oy>>rs,ox>>rs,co])
//
// `a` is the ACTIVATED input of the conv -- relu(AdaIN(x)), relu(x), x or a gradient dY -- already stored in HBM as 16-bit
// operand planes [N][Hin][Win][C8] (hi [, lo]) by the producer (lp_act_pack, or the epilogue of the previous conv): the
// prologue of the pre-activation ResBlock (generators/common/blocks.py:70-88) is applied ONCE per tensor by a bandwidth-bound
// pass instead of once per consumer tile on the VALU of the matrix kernel, the conv reads 2 bytes per activation instead of 4,
// and the same planes are what the weight-gradient kernel multiplies (no prologue recomputation there either).  Nearest x2
// upsampling stays fused (the halo is staged at the low resolution, the fragment reads apply the >>1 index map); zero padding
// comes from a zero page that out-of-image DMA lanes read.
//
// Tiling: a group of WM*WN waves owns BM = WM*MR*16 output pixels (NB images x TH x TW patch, rows in LINEAR patch order) and
// BN = WN*NR*16 output channels.  K loop = 32-channel chunks; per chunk the halo ((TH+2) x (TW+2) pixels x 64 B) is staged once
// and reused by all KS*KS taps; the weights of one kernel row ([KS*BN][32] per stage) are double buffered.  Both images are
// lane-linear (wave-uniform base + lane*16, the LDS-DMA contract): slot s of row r holds the 8-channel group s ^ ((r>>1)&3),
// i.e. the XOR swizzle is applied to the per-lane SOURCE address and undone by the ds_read_b128 of the fragments -- conflict
// free for 16 consecutive rows in the bank model of MI355X_MICROARCH.md (scripts/lds_swizzle_sim.py: 4 LDS cycles per read; the
// key (r>>2)&3 of the round-1 kernel was 2-way conflicted).
//
// Ping-pong (PP): the workgroup has two 4-wave groups on adjacent M tiles sharing the weight stages; each stage slot has two
// phases separated by workgroup barriers: group 0 multiplies while group 1 issues its share of the DMA, then they swap, so
// every SIMD always has one wave in its MFMA phase while the other wave's DMA issue (~60-100 cycles per 1 KiB piece) runs
// beside it.  Small feature maps: split-K over gridDim.z (partial tiles into a caller workspace, finished by splitk_reduce_kernel)
// and 8-wave groups.
//
// Precision modes (MFMA 16x16x32, fp32 accumulate): bf16 | f16 operands, 1 MFMA per k-step; bf16x3 = hi+lo split, 3 MFMAs.
#include "lp_common.h"
#include "conv_common.h"
#include "lp_hip.h"
#include "lp_internal.h"
#include <stdlib.h>
#include <type_traits>

__device__ __attribute__((aligned(64))) unsigned int lp_zero_page[16];      // what out-of-image / out-of-channel DMA lanes read


// GH (grouped convs with groups of <= 32 channels): output channels [co0, co0 + 32) only contract with the chunk co0 .. co0 + 31 and
// [co0 + 32, co0 + 64) only with the second chunk -- the other half of every stage's MFMAs multiplies the zeros of the block-diagonal weight
// image and is skipped (half the matrix work: in the bf16x3 mode the layer-1 / layer-2 grouped convs were bound by it, not by their traffic)
template <int KS, bool UPS, int WM, int WN, int MR, int NR, int PREC, bool PP, int NBUF = 2, int CC = 32, bool GH = false>
#ifndef LP_PP_MINW
#define LP_PP_MINW 2          // min waves per SIMD of the ping-pong kernels in the 16-bit modes: 2 = one workgroup per CU, 4 = two (<= 128 VGPRs)
#endif
#ifndef LP_X3_1X1_MINW
#define LP_X3_1X1_MINW 3      // min waves per SIMD of the bf16x3 1x1 kernels: hipcc then fits them in 127 VGPRs (170 without the bound: two
                              // workgroups per CU); with ONE activation buffer (48 KB of LDS) three share a CU -- r04: 133 -> 120 us at 262 k pixels x
                              // 128 -> 256, 47.9 -> 40.6 us at 64 -> 128 (profiles/r04_conv1x1_x3_occupancy.txt); 1: the round-3 build
#endif
__global__ __launch_bounds__(WM * WN * 64 * (PP ? 2 : 1), (PP && PREC != LP_PREC_BF16X3) ? LP_PP_MINW : (!PP && NBUF == 2 && WM * WN == 4 && PREC != LP_PREC_BF16X3) ? 2 :
                             (KS == 1 && !PP && WM * WN == 4 && PREC == LP_PREC_BF16X3) ? LP_X3_1X1_MINW : 1)
void conv_dma_kernel(Conv16Params p) {
    constexpr int ROWB = CC * 2;                         // CC channels per chunk; bytes per halo pixel / weight row of a chunk
    constexpr int SL = CC / 8, RPI = 64 / SL;            // 16-byte slots per row; rows covered by one 1 KiB DMA piece
    constexpr int KK = CC / 32;                          // k-steps (MFMA 16x16x32) per tap and chunk
    static_assert(CC == 32 || CC == 64, "32- or 64-channel chunks");
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    constexpr int NWAVE = WM * WN;                       // waves of one group
    constexpr int NWD = PP ? 2 * NWAVE : NWAVE;          // waves sharing the weight DMA of a stage
    constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
    constexpr int B_STAGE = KS * BN * ROWB;              // bytes of one weight stage (hi)
    constexpr int B_BUF = B_STAGE * (SPLIT ? 2 : 1);
    constexpr int NQ = B_STAGE / 1024;                   // DMA instructions per stage (hi)
    constexpr int AIT = ((BM == 128) ? (NWAVE == 8 ? 3 : 5) : 6) * (CC / 32) - (CC == 64 ? 1 : 0);     // max halo DMA instructions per wave (host checks p.hit <= AIT)
    constexpr int NBW = ((NQ + NWD - 1) / NWD) * (SPLIT ? 2 : 1);     // weight DMA instructions one wave issues per stage
    static_assert(B_STAGE % 1024 == 0, "stage must be a whole number of 1 KiB DMA pieces");
    static_assert(NBUF == 2 || (NBUF == 3 && !PP && NQ % NWD == 0) || (NBUF == 1 && !PP && KS == 1),
                  "3-deep weight ring: single group, uniform DMA count per wave; NBUF = 1: pointwise layers, ONE weight stage + double-buffered activations");
    static_assert(!PP || ((KS == 3 || KS == 2) && NWAVE == 4), "ping-pong: two 4-wave groups, 3x3 (or the 2x2 phase form)");
    static_assert(!(UPS && KS != 3), "1x1 convs commute with nearest upsampling: run them at low resolution; KS = 2 IS the phase form of an upsampled 3x3 conv");
    static_assert(KS == 1 || KS == 2 || KS == 3, "kernel sizes");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int grp = PP ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    const int tid = PP ? ((int)threadIdx.x & 255) : (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_d = PP ? grp * NWAVE + wave : wave;
    const int wm = wave / WN, wn = wave % WN;
    const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;

    int tile_id, cob;
    if (!conv16_block(p, tile_id, cob)) return;          // (uniform: padding of the XCD-ordered grid)
    if (PP) tile_id = tile_id * 2 + grp;
    int t = tile_id;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; t /= p.tiles_y;
    int ph = -1;                                         // KS == 2: phase (a, b) = (ph >> 1, ph & 1) of the x2-upsampled output this tile writes
    const bool pdg = (KS == 2) && (p.phase == 2);       // phase form of the DATA GRADIENT: the K loop runs over (phase, channel chunk), output = low-res dx
    if (KS == 2 && !pdg) { ph = t & 3; t >>= 2; }
    const int ng = t;
    const int n0 = ng << p.lNB, y0 = ty << p.lTH, x0 = tx << p.lTW;
    const int co0 = cob * BN;

    int HH, HW, oy, ox;
    if (KS == 1) { HH = TH; HW = TW; oy = y0; ox = x0; }
    else if (KS == 2) { HH = TH + 1; HW = TW + 1; oy = y0 + (ph >> 1) - 1; ox = x0 + (ph & 1) - 1; }      // (tiles and halo on the LOW-resolution grid;
                                                                                                             //  data gradient: origin per phase, issue_a_pd)
    else if (UPS) { HH = (TH >> 1) + 2; HW = (TW >> 1) + 2; oy = (y0 >> 1) - 1; ox = (x0 >> 1) - 1; }
    else { HH = TH + 2; HW = TW + 2; oy = y0 - 1; ox = x0 - 1; }
    const int halo_px = NBv * HH * HW;
    const int a_bytes = p.hit * NWAVE * 1024;                          // one halo image (hi), rounded up to whole DMA pieces
    const int a_buf = a_bytes * (SPLIT ? 2 : 1);                       // [hi][lo]
    // LDS map: PP: [halo grp0][halo grp1][stage 0][stage 1];  else: [halo 0][halo 1 (a_dbuf)][stage 0][stage 1]
    unsigned char* H_base = smem + (PP ? grp * a_buf : 0);
    unsigned char* B_base = smem + ((PP || p.a_dbuf) ? 2 : 1) * a_buf;

    // ---- fragment row geometry (linear patch order)
    int a_nbbase[MR], a_py[MR], a_px[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        int m = wm * (MR * 16) + mr * 16 + (lane & 15);
        int nb, py, px;
        tile_row_linear(m, p.lTH, p.lTW, nb, py, px);
        a_nbbase[mr] = nb * HH * HW; a_py[mr] = py; a_px[mr] = px;
    }
    // XOR key of row r (halo pixel / weight row): 64-B rows (r >> 1) & 3, 128-B rows r & 7 (scripts/lds_swizzle_sim.py: conflict free)
    auto rkey = [](int r) { return CC == 32 ? ((r >> 1) & 3) : (r & 7); };
    const int kb = lane >> 4;
    const int bkey = rkey(lane & 15);                  // key of this lane's weight rows (row & 15 == lane & 15)
    int b_off[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) b_off[nr] = (wn * (NR * 16) + nr * 16 + (lane & 15)) * ROWB;

    f32x4_t acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- halo DMA descriptors (chunk independent).  Piece q = k*NWAVE + wave covers halo pixels RPI*q .. RPI*q+RPI-1; lane -> pixel
    // RPI*q + lane/SL, LDS slot lane%SL, which holds channel group g = slot ^ key(pixel); RPI*q is a multiple of 8, so the key only
    // depends on the lane.
    const int a_g8 = ((lane % SL) ^ rkey(lane / SL)) * 8;
    int a_off[AIT];                                    // element offset of (pixel, group) in the plane, or -1 (zero page)
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int hp = (k * NWAVE + wave) * RPI + lane / SL;
        const int hx = hp % HW, t2 = hp / HW;
        const int hy = t2 % HH, nb = t2 / HH;
        const int n = n0 + nb, iy = oy + hy, ix = ox + hx;
        const bool inb = (k < p.hit) && (hp < halo_px) && (n < p.N) && (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
        a_off[k] = inb ? (((n * p.Hin + iy) * p.Win + ix) * p.C8 + a_g8) : -1;
    }
    const uint16_t* zero16 = (const uint16_t*)lp_zero_page;
    auto issue_a = [&](int chunk, unsigned char* Hbuf) {
        const int c0 = chunk * CC + (p.grouped ? co0 : 0);
        const bool cok = (c0 + a_g8) < p.C8;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)Hbuf);          // (m0 operand: an SGPR)
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            if (k < p.hit) {
                const bool ok = cok && a_off[k] >= 0;
                const size_t off = (size_t)(ok ? a_off[k] + c0 : 0);
                const unsigned d = dst + (unsigned)((k * NWAVE + wave) * 1024);
                lp_glds16(ok ? (p.a_hi + off) : zero16, d);
                if (SPLIT) lp_glds16(ok ? (p.a_lo + off) : zero16, d + a_bytes);
            }
        }
    };
    // phase form of the data gradient (KS == 2, p.phase == 2): chunk = (phase, channel chunk of dy); the halo of phase (a, b) is the (TH + 1) x
    // (TW + 1) patch of the PHASE-SUBSAMPLED dy with origin (y0 - a, x0 - b): halo pixel (hy, hx) = dy pixel (2 (y0 - a + hy) + a, 2 (x0 - b + hx) + b)
    // of the [N][Hin = 2H][Win = 2W] planes -- a stride-2 gather by the DMA lanes; the per-lane geometry is re-derived per chunk (a few integer
    // operations per 1 KiB piece) instead of being held in registers for four phases
    const int nch1 = p.CinP / CC;                          // channel chunks (per phase)
    auto issue_a_pd = [&](int chunk, unsigned char* Hbuf) {
        const int phq = chunk / nch1, c0 = (chunk - phq * nch1) * CC;
        const int pa = phq >> 1, pb = phq & 1;
        const bool cok = (c0 + a_g8) < p.C8;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)Hbuf);          // (m0 operand: an SGPR)
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            if (k < p.hit) {
                const int hp = (k * NWAVE + wave) * RPI + lane / SL;
                const int hx = hp % HW, t2 = hp / HW;
                const int hy = t2 % HH, nb = t2 / HH;
                const int n = n0 + nb, iy = y0 - pa + hy, ix = x0 - pb + hx;          // low-resolution position inside phase (pa, pb)
                const bool ok = cok && (hp < halo_px) && (n < p.N) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
                const size_t off = ok ? ((size_t)((n * p.Hin + 2 * iy + pa) * p.Win + 2 * ix + pb) * p.C8 + a_g8 + c0) : 0;
                const unsigned d = dst + (unsigned)((k * NWAVE + wave) * 1024);
                lp_glds16(ok ? (p.a_hi + off) : zero16, d);
                if (SPLIT) lp_glds16(ok ? (p.a_lo + off) : zero16, d + a_bytes);
            }
        }
    };
    auto issue_halo = [&](int chunk, unsigned char* Hbuf) {
        if (KS == 2 && pdg) issue_a_pd(chunk, Hbuf); else issue_a(chunk, Hbuf);
    };
    // LDS-DMA of the weight tile of stage (chunk, ky) into stage buffer `buf`: piece q covers rows 16q .. 16q+15 of [KS*BN][32]
    auto issue_b = [&](int chunk, int ky, int buf) {
        const int phw = (KS == 2) ? (pdg ? chunk / nch1 : ph) : 0;          // phase whose 2 x 2 tap images this stage reads
        const int c0 = (KS == 2 && pdg ? chunk - phw * nch1 : chunk) * CC;
        const unsigned dst_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(B_base + buf * B_BUF));
        const int rr = lane / SL, slot = lane % SL;
        const int g8 = (slot ^ rkey(rr)) * 8;
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += NWD) {
            const int q = q0 + wave_d;
            if (NQ % NWD == 0 || q < NQ) {
                const int r = q * RPI + rr;                                          // row inside the stage: kx * BN + n
                const int kx = r / BN, n = r % BN;
                const size_t off = ((size_t)(((KS == 2 ? phw * 4 : 0) + ky * KS + kx) * p.CoutP + co0 + n) * p.CinP + c0 + g8);
                lp_glds16(p.w_hi + off, dst_lds + q * 1024);
                if (SPLIT) lp_glds16(p.w_lo + off, dst_lds + B_STAGE + q * 1024);
            }
        }
    };
    // MFMAs of one stage: kernel row ky, halo image Hbuf, weights in stage buffer bbuf
    // HC (GH only): integral constant = which half of the N fragments this chunk feeds (compile time: a run-time predicate inside the unrolled
    // MFMA loops cost more than the skipped work saved -- 79 -> 106 us per grouped launch, scripts/r04_call25.sh)
    auto compute_h = [&](int ky, const unsigned char* Hbuf, int bbuf, auto HC) {
        constexpr int half = decltype(HC)::value;
        const unsigned char* A_hi = Hbuf;
        const unsigned char* A_lo = Hbuf + a_bytes;
        const unsigned char* Bc = B_base + bbuf * B_BUF;
        s16x8_t fa[2][MR], fb[2][NR], fal[2][MR], fbl[2][NR];
        constexpr int STEPS = KS * KK;
        auto fetch = [&](int st, int set) {
            const int kx = st / KK, kk = st % KK;
            const int dy = (KS >= 2) ? ky : 0, dx = (KS >= 2) ? kx : 0;
            const unsigned char* Bk = Bc + kx * (BN * ROWB);
            const int grp8 = kk * 4 + kb;                                  // 8-channel group of this lane's fragment slice
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                int hy, hx;
                if (KS == 1) { hy = a_py[mr]; hx = a_px[mr]; }
                else if (UPS) { hy = ((a_py[mr] + dy - 1) >> 1) + 1; hx = ((a_px[mr] + dx - 1) >> 1) + 1; }
                else { hy = a_py[mr] + dy; hx = a_px[mr] + dx; }
                const int hp = a_nbbase[mr] + hy * HW + hx;
                const int off = hp * ROWB + ((grp8 ^ rkey(hp)) << 4);
                fa[set][mr] = *(const s16x8_t*)(A_hi + off);
                if (SPLIT) fal[set][mr] = *(const s16x8_t*)(A_lo + off);
            }
            const int bsl = (grp8 ^ bkey) << 4;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                if (GH && nr / (NR / 2) != half) continue;
                fb[set][nr] = *(const s16x8_t*)(Bk + b_off[nr] + bsl);
                if (SPLIT) fbl[set][nr] = *(const s16x8_t*)(Bk + B_STAGE + b_off[nr] + bsl);
            }
        };
        fetch(0, 0);
        // (measured null, round 3: s_setprio(1) around this MFMA cluster of the ping-pong kernel -- the SIMD's other wave issues LDS-DMA
        //  beside it -- changes the 12 layer classes of scripts/conv_micro.py by -2 .. +2 %; two ping-pong workgroups per CU at <= 128
        //  VGPRs (LP_PP_MINW=4) run 1.3 - 1.7x slower: profiles/r03_conv_dma_variants.md)
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int cur = st & 1;
            if (st + 1 < STEPS) fetch(st + 1, cur ^ 1);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    if (GH && nr / (NR / 2) != half) continue;
                    if (SPLIT) {
                        acc[mr][nr] = mfma16(fal[cur][mr], fb[cur][nr], acc[mr][nr]);
                        acc[mr][nr] = mfma16(fa[cur][mr], fbl[cur][nr], acc[mr][nr]);
                    }
                    acc[mr][nr] = mfma16t<F16>(fa[cur][mr], fb[cur][nr], acc[mr][nr]);
                }
        }
    };
    auto compute = [&](int ky, const unsigned char* Hbuf, int bbuf, int chunk_abs) {
        if constexpr (!GH) compute_h(ky, Hbuf, bbuf, std::integral_constant<int, -1>{});
        else if (chunk_abs & 1) compute_h(ky, Hbuf, bbuf, std::integral_constant<int, 1>{});
        else compute_h(ky, Hbuf, bbuf, std::integral_constant<int, 0>{});
    };

    const int nch_total = (p.CinP / CC) * ((KS == 2 && pdg) ? 4 : 1);
    const int per = (nch_total + p.ksplit - 1) / p.ksplit;
    const int cbeg = blockIdx.z * per;
    const int nch = min(nch_total, cbeg + per) - cbeg;        // chunks of this workgroup: [cbeg, cbeg + nch)
    if (nch <= 0) return;                                       // (uniform) nothing to contribute
    const int S = nch * KS;
    issue_b(cbeg, 0, 0);
    issue_halo(cbeg, H_base);

    if constexpr (PP) {
        // slot k = stage k of both groups.  Phase A(k): group 0 multiplies stage k, group 1 stages; phase B(k): swapped.  Staging
        // duties: the group's share of the weight DMA of stage k+1 (buffer (k+1)&1, last read in slot k-1) and, in the phase right
        // before its compute of a chunk's first stage, the halo of that chunk into the group's single halo buffer (the group itself
        // is its only reader and it is not multiplying now).  Every barrier is preceded by vmcnt(0): all DMA issued so far has
        // landed and is visible to the waves that pass the barrier.
        for (int chunk = 0; chunk < nch; ++chunk) {
            const bool has_next = chunk + 1 < nch;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int k = chunk * KS + ky;
                const int bbuf = k & 1;
                lp_wait_vm0();
                __syncthreads();                                   // ---- phase A(k)
                if (grp == 0) {
                    compute(ky, H_base, bbuf, cbeg + chunk);
                } else {
                    if (ky == 0 && chunk > 0) issue_halo(cbeg + chunk, H_base);
                    if (k + 1 < S) issue_b(cbeg + (k + 1) / KS, (k + 1) % KS, bbuf ^ 1);
                }
                lp_wait_vm0();
                __syncthreads();                                   // ---- phase B(k)
                if (grp == 1) {
                    compute(ky, H_base, bbuf, cbeg + chunk);
                } else {
                    if (ky == KS - 1 && has_next) issue_halo(cbeg + chunk + 1, H_base);
                    if (k + 1 < S) issue_b(cbeg + (k + 1) / KS, (k + 1) % KS, bbuf ^ 1);
                }
            }
        }
    } else {
        // one group: top of stage s: own DMA landed, barrier (everyone finished stage s-1); issue the weights of stage s+NBUF-1 and, in
        // a chunk's first stage, the halo of the NEXT chunk into the other halo buffer (it has the whole chunk to land); multiply.
        // NBUF == 3 (grids of <= 1 workgroup per CU, where nothing else hides the L2/HBM latency of a stage's weights): the DMA of
        // stage s+1 stays in flight across the barrier -- only what stage s needs (everything older, in issue order) is waited for.
        int abuf = 0;
        if constexpr (NBUF == 1) {
            // pointwise layers bound by their activation traffic (round 6): the ACTIVATION chunk -- the operand that comes from HBM -- is double
            // buffered and prefetched one chunk ahead; the weight stage -- L2 resident, a third of the latency -- has ONE buffer and is restaged
            // after the chunk's MFMAs.  Same LDS as one activation buffer + two weight stages (three workgroups per CU), but what a chunk now
            // waits for is an L2 hit instead of an HBM round trip.
            for (int chunk = 0; chunk < nch; ++chunk) {
                const bool has_next = chunk + 1 < nch;
                lp_wait_vm0();
                __syncthreads();
                if (has_next) issue_halo(cbeg + chunk + 1, H_base + __builtin_amdgcn_readfirstlane((abuf ^ 1) * a_buf));      // (the DMA's LDS base travels in m0: an SGPR)
                compute(0, H_base + abuf * a_buf, 0, cbeg + chunk);
                if (has_next) { __syncthreads(); issue_b(cbeg + chunk + 1, 0, 0); }
                abuf ^= 1;
            }
        } else {
        if (NBUF == 3 && S > 1) issue_b(cbeg + 1 / KS, 1 % KS, 1);
        for (int chunk = 0; chunk < nch; ++chunk) {
            const bool has_next = chunk + 1 < nch;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int s = chunk * KS + ky;
                // (one halo buffer: the next chunk's halo is issued AFTER this stage's weights -- it is the newest DMA, so only a full drain
                //  covers it; the counted wait is for the double-buffered halo, which is issued first)
                if (NBUF == 3 && s + 1 < S && p.a_dbuf) lp_wait_vm<NBW>(); else lp_wait_vm0();
                __syncthreads();
                // (halo first: it is older than the weights issued below, so the counted wait of the next stage covers it)
                if (p.a_dbuf && ky == 0 && has_next) issue_halo(cbeg + chunk + 1, H_base + (abuf ^ 1) * a_buf);
                const int sp = s + NBUF - 1;
                if (sp < S) issue_b(cbeg + sp / KS, sp % KS, sp % NBUF);
                compute(ky, H_base + abuf * a_buf, s % NBUF, cbeg + chunk);
                if (!p.a_dbuf && ky == KS - 1 && has_next) {      // (LDS-tight tiles) one halo buffer: restage it once everyone has read it
                    __syncthreads();
                    issue_halo(cbeg + chunk + 1, H_base);
                }
            }
            if (p.a_dbuf) abuf ^= 1;
        }
        }
    }

    // ---- epilogue (conv_common.h).  1x1 layers transpose 16 rows (one MFMA row block) at a time, 3x3 kernels the whole block
    conv16_epilogue<WM, WN, MR, NR, PREC, (KS == 1) ? 1 : MR>(p, acc, smem, wave_d, wm, wn, lane, n0, y0, x0, co0, NBv, tile_id, ph);
}

// Split-K finish: y = alpha * sum_s part[s] + bias + res, ReLU mask, 16-bit planes of the consumer -- the whole epilogue of the conv,
// once, on the summed tile (4 channels per thread; the slices are plain coalesced stores of the conv workgroups: no atomics, no
// memset of y, and the planes need no extra pack pass).
template <int PREC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(Conv16Params p) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    const int C4 = p.Cout >> 2;
    const size_t P = (size_t)p.N * p.H * p.W, items = P * C4, slice = P * p.Cout;
    float alpha = p.alpha ? *p.alpha : 1.f;
    if (p.alpha2) alpha *= *p.alpha2;
    float am = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % C4) * 4;
        const size_t pix = i / C4;
        float4 v = *(const float4*)(p.part + i * 4);
        for (int z = 1; z < p.ksplit; ++z) {
            const float4 q = *(const float4*)(p.part + z * slice + i * 4);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bv = *(const float4*)(p.bias + co);
        v.x = fmaf(v.x, alpha, bv.x); v.y = fmaf(v.y, alpha, bv.y); v.z = fmaf(v.z, alpha, bv.z); v.w = fmaf(v.w, alpha, bv.w);
        if (p.res) {
            const int ox = (int)(pix % p.W); const size_t t = pix / p.W;
            const int oy = (int)(t % p.H), n = (int)(t / p.H);
            const float4 rv = *(const float4*)(p.res + ((size_t)(n * (p.H >> p.res_shift) + (oy >> p.res_shift)) * (p.W >> p.res_shift)
                                                        + (ox >> p.res_shift)) * p.Cout + co);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (p.mask16) {
            const ushort4 mv = *(const ushort4*)(p.mask16 + pix * p.Co8 + co);
            v.x = (mv.x - 1u) < 0x7fffu ? v.x : 0.f; v.y = (mv.y - 1u) < 0x7fffu ? v.y : 0.f;
            v.z = (mv.z - 1u) < 0x7fffu ? v.z : 0.f; v.w = (mv.w - 1u) < 0x7fffu ? v.w : 0.f;
        }
        if (p.o_relu & 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (p.y) *(float4*)(p.y + pix * p.Cout + co) = v;
        am = lp_amax4(am, v);
        if (p.o_hi) {
            float o[4] = {v.x, v.y, v.z, v.w};
            ushort4 oh, ol;
            uint16_t* ohp = (uint16_t*)&oh; uint16_t* olp = (uint16_t*)&ol;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float q = (p.o_relu & 1) ? fmaxf(o[j], 0.f) : o[j];
                ohp[j] = lp_f32_to_op16<F16>(q);
                if (SPLIT) olp[j] = lp_f32_to_op16<false>(q - lp_op16_to_f32<false>(ohp[j]));
            }
            *(ushort4*)(p.o_hi + pix * p.Co8 + co) = oh;
            if (SPLIT) *(ushort4*)(p.o_lo + pix * p.Co8 + co) = ol;
        }
    }
    if (p.amax) lp_amax_commit(am, p.amax, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------------------
// host side: tile selection + dispatch
// ------------------------------------------------------------------------------------------------------------------
template <int KS, bool UPS, int WM, int WN, int MR, int NR, int PREC, bool PP, int NBUF = 2, int CC = 32, bool GH = false>
static int launch_v(Conv16Params& p, size_t lds, dim3 grid, hipStream_t stream) {
    auto kern = conv_dma_kernel<KS, UPS, WM, WN, MR, NR, PREC, PP, NBUF, CC, GH>;
    static thread_local int attr_dev = -1;                 // per host thread and device (main and autograd threads both launch)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipGetDevice failed");
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_dev = dev;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64 * (PP ? 2 : 1)), lds, stream, p);
    return lp_check_launch("conv_dma");
}

template <int KS, bool UPS, int WM, int WN, int MR, int NR, int PREC, int CC = 32, bool GH = false>
static int launch_conv16(Conv16Params& p, hipStream_t stream) {
    constexpr int BM = WM * MR * 16, BN = WN * NR * 16, NWAVE = WM * WN;
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    constexpr int ROWB = CC * 2, RPI = 64 / (CC / 8);                    // bytes per halo pixel / weight row of a chunk; rows per 1 KiB DMA piece
    constexpr size_t B_BUF = (size_t)KS * BN * ROWB * (SPLIT ? 2 : 1);
    constexpr size_t LDS_MAX = 160 * 1024;
    constexpr int AIT = ((BM == 128) ? (NWAVE == 8 ? 3 : 5) : 6) * (CC / 32) - (CC == 64 ? 1 : 0);
    if (p.CinP % CC) return lp_set_error(LP_ERR_ARG, "conv16: padded input channels must be a multiple of the chunk size");
    const bool phf = (KS == 2) && p.phase == 1, phd = (KS == 2) && p.phase == 2;          // phase forms: forward | data gradient
    const int GHt = phf ? p.Hin : p.H, GWt = phf ? p.Win : p.W;          // the grid the tiles cover (forward phase form: the low-resolution INPUT grid;
                                                                         //  data gradient: its low-resolution OUTPUT, p.H x p.W as always)
    choose_tile(BM, p.N, GHt, GWt, &p.lTH, &p.lTW, &p.lNB);
    int HH, HW, halo_px;
    for (;;) {       // fewer images per tile until the halo fits the per-wave descriptor budget (tiny feature maps)
        const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;
        if (KS == 1) { HH = TH; HW = TW; } else if (KS == 2) { HH = TH + 1; HW = TW + 1; } else if (UPS) { HH = TH / 2 + 2; HW = TW / 2 + 2; } else { HH = TH + 2; HW = TW + 2; }
        halo_px = NBv * HH * HW;
        if ((halo_px + RPI - 1) / RPI <= AIT * NWAVE || p.lNB == 0) break;
        --p.lNB;
    }
    const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;
    p.tiles_x = (GWt + TW - 1) / TW; p.tiles_y = (GHt + TH - 1) / TH;
    const int NH = (halo_px + RPI - 1) / RPI;
    p.hit = (NH + NWAVE - 1) / NWAVE;
    if (p.hit > AIT) return lp_set_error(LP_ERR_UNSUPPORTED, "conv16: halo exceeds the DMA descriptor budget");
    const size_t a_buf = (size_t)p.hit * NWAVE * 1024 * (SPLIT ? 2 : 1);
    static const int adbuf_env = getenv("LP_CONV_ADBUF") ? atoi(getenv("LP_CONV_ADBUF")) : -1;      // 0: one A buffer also where two fit (A/B knob)
    p.a_dbuf = (2 * a_buf + 2 * B_BUF <= LDS_MAX) ? 1 : 0;
    if (adbuf_env == 0 && KS == 1) p.a_dbuf = 0;
    // bf16x3 pointwise layers with short contractions (<= 8 chunks: the layers bound by their traffic): one activation buffer, i.e. a
    // third resident workgroup per CU, beats the double buffer (longer contractions lose 5 - 10 % with it and keep two)
    if (adbuf_env < 0 && KS == 1 && SPLIT && CC == 32 && p.CinP <= 256) p.a_dbuf = 0;
    // LP_CONV1X1_B1=1 (round 6 experiment, OFF): those layers with TWO activation buffers and ONE weight stage instead (NBUF = 1 schedule of the kernel):
    // the HBM round trip of every activation chunk is hidden behind the previous chunk's MFMAs -- and NOTHING moves (119.8 -> 117.4 us at 262 k pixels x
    // 128 -> 256, 107.2 -> 105.2 us at 256 -> 128; = 2: also the longer contractions, 5 - 20 % slower: profiles/r06_conv1x1_schedules.txt).  The layers'
    // times fit  bytes / 8.7 TB/s + MFMA work / 0.7 PF/s: what bounds them is the matrix pipe's stage loop (16 fragment reads per 48 MFMAs and a barrier
    // per 32-channel stage), not the exposed latency of their activation traffic.
    static const int b1_env = getenv("LP_CONV1X1_B1") ? atoi(getenv("LP_CONV1X1_B1")) : 0;
    const bool b1 = KS == 1 && CC == 32 && ((b1_env && !p.a_dbuf) || (b1_env >= 2 && SPLIT && BN == 128));          // 2: also the longer bf16x3 contractions
    if (b1) p.a_dbuf = 1;
    const size_t lds_halo = (p.a_dbuf ? 2 : 1) * a_buf;
    size_t lds = lds_halo + (b1 ? 1 : 2) * B_BUF;
    size_t epi = (size_t)NWAVE * ((KS == 1) ? 16 : MR * 16) * (NR * 16 + 4) * sizeof(float);      // LDS transpose of the coalesced epilogue (1x1: 16 rows per wave at a time)
    const int tiles = p.tiles_x * p.tiles_y * ((p.N + NBv - 1) / NBv) * (phf ? 4 : 1);          // (forward phase form: every tile position once per phase)
    // ping-pong: two adjacent M tiles per workgroup, when the paired grid still covers the CUs.  LP_CONV_PP = 0 | 1 overrides.
    constexpr bool pp_ok = (KS == 3 || KS == 2) && (NWAVE == 4);
    static const int pp_env = getenv("LP_CONV_PP") ? atoi(getenv("LP_CONV_PP")) : -1;
    const long long pp_wgs = (long long)((tiles + 1) / 2) * ((p.Cout + BN - 1) / BN);
    // (the two groups of a ping-pong workgroup share their weight stages: in the phase form a pair of tiles must not straddle two phases)
    const bool pp = pp_ok && (pp_env >= 0 ? pp_env != 0 : pp_wgs >= 200) && tiles >= 2 && p.a_dbuf && (!phf || ((p.tiles_x * p.tiles_y) & 1) == 0);
    if (pp) epi *= 2;
    if (lds < epi) lds = epi;
    if (lds > LDS_MAX) return lp_set_error(LP_ERR_UNSUPPORTED, "conv16 tile needs too much LDS");
    dim3 grid(pp ? (tiles + 1) / 2 : tiles, (p.Cout + BN - 1) / BN);
    p.xcd_map = 0; p.ntiles = grid.x; p.nco = grid.y;
    {   // split-K when the output tiling alone cannot fill the 256 CUs (4x4 ... 16x16 layers with K = 9*512): grid.z slices of the
        // contraction write partial tiles into the caller's workspace, splitk_reduce_kernel (launched by lp_conv16_fwd) finishes.
        // 1x1 convs are not split: their whole contraction is 16 stages, less than the cost of a second launch.
        static const int max_split = getenv("LP_CONV_KSPLIT") ? atoi(getenv("LP_CONV_KSPLIT")) : 8;
        static const int split_wgs = getenv("LP_CONV_SPLIT_WGS") ? atoi(getenv("LP_CONV_SPLIT_WGS")) : 256;   // split while <= this many workgroups result
        const int wgs = grid.x * grid.y, nch = (p.CinP / CC) * (phd ? 4 : 1);
        int ks = 1;
        if ((KS == 3 || KS == 2) && (p.Cout & 3) == 0 && p.part)
            while (ks < max_split && wgs * ks * 2 <= split_wgs && nch / (ks * 2) >= 2 &&
                   (long long)(ks * 2) * p.N * p.H * p.W * p.Cout * (long long)sizeof(float) <= p.part_bytes) ks *= 2;
        // every slice must own at least one chunk: the kernel gives slice z the chunks [z*per, (z+1)*per) with per = ceil(nch/ks), and
        // splitk_reduce_kernel sums ALL ks slices of the (uninitialised) workspace -- e.g. 18 chunks (Cin 576) over 8 slices would
        // leave slices 6 and 7 unwritten.  Shrink ks to the fixed point of ks = ceil(nch / ceil(nch / ks)).
        for (;;) { const int per = (nch + ks - 1) / ks, k2 = (nch + per - 1) / per; if (k2 == ks) break; ks = k2; }
        p.ksplit = ks;
        grid.z = ks;
    }
    if ((KS == 3 || KS == 2) && !p.grouped) grid = conv16_grid(p, (int)grid.x, (int)grid.y, (int)grid.z);      // XCD-aware order (dense 3x3: the halo is the shared operand)
    p.stats_rows = 0;
    if (p.stats) {
        // fused norm statistics need: full tiles (every wave's MR*16 rows inside ONE image, tiles covering the images exactly, image-major
        // row-block order), the coalesced epilogue (Cout % 4 == 0), no split-K (its finish kernel has no statistics), room in the buffer
        constexpr int WR = MR * 16;
        const long long rows = (long long)(pp ? 2 * ((tiles + 1) / 2) : tiles) * WM;
        // (forward phase form: a tile's row blocks are ordered (image group, phase, tile): with several images per tile the rows of ONE image would
        //  not be contiguous -- the partials are only offered for one image per tile, small maps take the statistics pass)
        const bool ok = (NBv * TH * TW == BM) && (TH * TW >= WR) && (GHt % TH == 0) && (GWt % TW == 0) && ((p.Cout & 3) == 0) && p.ksplit == 1 &&
                        rows * p.Cout * 3 <= p.stats_cap && (!phf || NBv == 1);
        if (ok) p.stats_rows = (p.H * p.W) / WR; else p.stats = nullptr;
    }
    // (64-channel chunks -- CC = 64, twice the MFMAs between two barriers -- were measured 7-15 % SLOWER on the 64^2..256^2 layers: their
    //  146 KB of LDS leave one workgroup per CU, while the 72 KB of the 32-channel kernel let two ping-pong workgroups share a CU)
    if constexpr (pp_ok) { if (pp) return launch_v<KS, UPS, WM, WN, MR, NR, PREC, true, 2, CC, GH>(p, lds, grid, stream); }
    if constexpr (KS == 1 && CC == 32 && !GH) { if (b1) return launch_v<KS, UPS, WM, WN, MR, NR, PREC, false, 1, CC, GH>(p, lds, grid, stream); }
    // 3-deep weight ring when the grid gives each CU at most ~one workgroup (its LDS would exclude a second one anyway) and it fits
    constexpr int NQ = KS * BN * ROWB / 1024;
    constexpr bool ring_ok = (NQ % NWAVE == 0);
    static const int nbuf_env = getenv("LP_CONV_NBUF") ? atoi(getenv("LP_CONV_NBUF")) : 0;      // 2 | 3 forces
    if constexpr (ring_ok) {
        const size_t lds3 = lds_halo + 3 * B_BUF;
        const long long total_wgs = (long long)grid.x * grid.y * grid.z;
        const bool ring = (nbuf_env ? nbuf_env == 3 : total_wgs <= 320) && lds3 <= LDS_MAX;
        if (ring) return launch_v<KS, UPS, WM, WN, MR, NR, PREC, false, 3, CC, GH>(p, lds3 > epi ? lds3 : epi, grid, stream);
    }
    return launch_v<KS, UPS, WM, WN, MR, NR, PREC, false, 2, CC, GH>(p, lds, grid, stream);
}

template <int PREC>
static int dispatch_conv16(Conv16Params& p, int ks, int ups, hipStream_t s) {
    if (ks == 3 && ups >= 2) {      // phase-decomposed x2-upsampled conv (Conv16Params::phase): 2 x 2 taps per phase on the low-resolution grid
        const int gh = ups == 2 ? p.Hin : p.H, gw = ups == 2 ? p.Win : p.W;
        if ((p.Cout & 3) || gh < 2 || gw < 2) return lp_set_error(LP_ERR_UNSUPPORTED, "phase-decomposed conv: Cout % 4 == 0, low-resolution grid >= 2 x 2");
        if (p.Cout <= 64 && gh * gw >= 256) return launch_conv16<2, false, 4, 1, 4, 4, PREC>(p, s);
        return launch_conv16<2, false, 2, 2, 4, 4, PREC>(p, s);
    }
    if (ks == 3) {      // the tap-pipelined kernel (conv_pipe.hip) takes the layers it covers: maps whose tiling fills the chip
        const int r = lp_conv_pipe_launch(p, ups, PREC, s);
        if (r) return r < 0 ? r : LP_OK;
    }
    if (ks == 1 && !ups) {      // the chunk-pipelined 1x1 kernel: layers with >= 256 workgroups
        const int r = lp_conv1x1_pipe_launch(p, PREC, s);
        if (r) return r < 0 ? r : LP_OK;
    }
    const bool big_img = p.H * p.W >= 256;     // a 256-pixel patch fits inside one image
    // 8-wave groups (two waves per SIMD, each 64 x 32 of the 128 x 128 tile) for the small feature maps (<= 16x16): those run
    // split-K with few stages per workgroup, so prologue and epilogue dominate.  LP_CONV_W8 = 0 | 1 forces it off / on.
    static const int w8_env = getenv("LP_CONV_W8") ? atoi(getenv("LP_CONV_W8")) : -1;
    const bool w8 = w8_env >= 0 ? (w8_env != 0) : (p.H * p.W <= 256);
    if (w8 && p.Cout > 64) {
        // bf16x3 small maps (round 6, LP_CONV_X3_BN64=0 restores the 128 x 128 tile): the 8-wave 128 x 128 tile's hi + lo weight stages (49 KB each) leave no room for the 3-deep
        // ring, so every stage drains vmcnt(0); 128 x 64 tiles of four waves (24.5 KB stages) keep the ring and double the workgroups per slice
        static const int x3bn64 = getenv("LP_CONV_X3_BN64") ? atoi(getenv("LP_CONV_X3_BN64")) : 1;      // measured (profiles/r06_x3_small_maps.txt): 4x4 23.2 -> 17.7 us, 8x8 25.4 -> 23.3, 8x8 upsampled 26.0 -> 21.9, 16x16 unchanged; step -0.35 ms
        if (PREC == LP_PREC_BF16X3 && x3bn64 && ks == 3) {
            if (!ups) return launch_conv16<3, false, 2, 2, 4, 2, PREC>(p, s);
            return launch_conv16<3, true, 2, 2, 4, 2, PREC>(p, s);
        }
        if (ks == 3 && !ups) return launch_conv16<3, false, 2, 4, 4, 2, PREC>(p, s);
        if (ks == 3 && ups) return launch_conv16<3, true, 2, 4, 4, 2, PREC>(p, s);
        if (ks == 1 && !ups) return launch_conv16<1, false, 2, 4, 4, 2, PREC>(p, s);
    }
    if (PREC == LP_PREC_BF16X3 && ks == 3 && p.Cout > 64) {
        // bf16x3 layers whose paired (ping-pong) grid would not cover the chip (e.g. 32 x 32 x 512 at N = 8: 256 workgroups of 128 x 128) fall to the
        // single-group kernel, whose doubled weight stages do not fit a 3-deep ring either: the same 128 x 64 four-wave tile as the small maps
        // (LP_CONV_X3_BN64=2 enables it; measured in profiles/r06_x3_small_maps.txt)
        static const int x3mid = getenv("LP_CONV_X3_BN64") ? atoi(getenv("LP_CONV_X3_BN64")) : 1;
        const long long tiles128 = ((long long)p.N * p.H * p.W + 127) / 128;
        if (x3mid >= 2 && ((tiles128 + 1) / 2) * ((p.Cout + 127) / 128) < 200) {
            if (!ups) return launch_conv16<3, false, 2, 2, 4, 2, PREC>(p, s);
            return launch_conv16<3, true, 2, 2, 4, 2, PREC>(p, s);
        }
    }
    if (ks == 3 && !ups) {
        if (p.Cout <= 16 && big_img) return launch_conv16<3, false, 4, 1, 4, 1, PREC>(p, s);
        if (p.Cout <= 64 && big_img) return launch_conv16<3, false, 4, 1, 4, 4, PREC>(p, s);
        return launch_conv16<3, false, 2, 2, 4, 4, PREC>(p, s);
    }
    if (ks == 3 && ups) {
        if (p.Cout <= 64 && big_img) return launch_conv16<3, true, 4, 1, 4, 4, PREC>(p, s);
        return launch_conv16<3, true, 2, 2, 4, 4, PREC>(p, s);
    }
    if (ks == 1 && !ups) {
        // 1x1: one tap per chunk, so a 32-channel stage is 16 MFMAs per wave between two barriers.  With >= 512 input channels and single-plane
        // operands the stage is a 64-channel chunk (32 MFMAs, half the barriers): +3..12% on the embedder's K >= 512 layers, nothing below
        // (those are bound by the fp32 output store), and SLOWER with bf16x3's doubled planes (LDS per workgroup doubles again, occupancy
        // drops) -- profiles/r03_conv1x1_cc64.txt.  LP_CONV_CC1 = 32 | 64 forces.
        static const int cc_env = getenv("LP_CONV_CC1") ? atoi(getenv("LP_CONV_CC1")) : 0;
        const bool cc64 = (cc_env ? cc_env == 64 : (p.Cin >= 512 && PREC != LP_PREC_BF16X3)) && (p.CinP % 64 == 0);
        // LP_CONV1X1_TILE=256: 256 x 128 output tiles (8 waves, 64 x 64 each) for the big pointwise layers -- 25 % fewer operand bytes staged
        // L2 -> LDS per flop than the 128 x 128 tile.  Measured NULL (profiles/r03_conv1x1_tile256.txt: every layer within +-5 %, the step
        // 39.4 vs 39.3 ms), so the K >= 512 layers' ~470-600 TF/s is not an L2-bandwidth limit but the one-stage-ahead DMA + barrier per stage
        // of this loop; the default stays 128.
        static const int tile_env = getenv("LP_CONV1X1_TILE") ? atoi(getenv("LP_CONV1X1_TILE")) : 0;
        const long long pix = (long long)p.N * p.H * p.W;
        const bool tall = (tile_env ? tile_env == 256 : false) && p.Cout >= 128 && p.H * p.W >= 256 &&
                          (pix / 256) * ((p.Cout + 127) / 128) >= 512;
        if (cc64) {
            if (p.Cout <= 64 && big_img) return launch_conv16<1, false, 4, 1, 4, 4, PREC, 64>(p, s);
            if (tall) return launch_conv16<1, false, 4, 2, 4, 4, PREC, 64>(p, s);
            return launch_conv16<1, false, 2, 2, 4, 4, PREC, 64>(p, s);
        }
        if (p.Cout <= 64 && big_img) return launch_conv16<1, false, 4, 1, 4, 4, PREC>(p, s);
        if (tall) return launch_conv16<1, false, 4, 2, 4, 4, PREC>(p, s);
        // (round 6, measured and removed: 128 x 64 tiles for the bf16x3 layers with short contractions -- 41.6 -> 54.8 us, 119.8 -> 154.7 us,
        //  profiles/r06_conv1x1_schedules.txt)
        return launch_conv16<1, false, 2, 2, 4, 4, PREC>(p, s);
    }
    return lp_set_error(LP_ERR_UNSUPPORTED, "unsupported conv configuration");
}

// upper bound of the split-K workspace: the split factor never exceeds 8 nor 256 / (workgroups of the largest tile, 256 x 128)
extern "C" long long lp_conv16_fwd_workspace_bytes(int N, int H, int W, int Cout, int ksize) {
    if (ksize != 3 || (Cout & 3)) return 0;
    const long long P = (long long)N * H * W;
    const long long wgs_lb = ((P + 255) / 256) * ((Cout + 127) / 128);
    static const int max_split = getenv("LP_CONV_KSPLIT") ? atoi(getenv("LP_CONV_KSPLIT")) : 8;         // (the knobs of launch_conv16)
    static const int split_wgs = getenv("LP_CONV_SPLIT_WGS") ? atoi(getenv("LP_CONV_SPLIT_WGS")) : 256;
    long long ks = split_wgs / wgs_lb; if (ks > max_split) ks = max_split;
    if (ks < 2) return 0;
    return ks * P * Cout * (long long)sizeof(float);
}

extern "C" int lp_conv16_fwd(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                             const float* bias, const float* res, const float* alpha, const float* alpha2,
                             int N, int H, int W, int Cin, int Cout, int CinP, int CoutP,
                             int ksize, int upsample, int res_shift, int prec, const uint16_t* relu_mask16,
                             uint16_t* out_hi, uint16_t* out_lo, int out_relu, float* workspace, long long workspace_bytes, float* amax_slots,
                             void* stream) {
    if (!y) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: null pointer");
    return lp_conv16_fwd_stats(a_hi, a_lo, w_hi, w_lo, y, bias, res, alpha, alpha2, N, H, W, Cin, Cout, CinP, CoutP, ksize, upsample, res_shift, prec,
                               relu_mask16, out_hi, out_lo, out_relu, workspace, workspace_bytes, amax_slots, nullptr, 0, nullptr, stream);
}

// lp_conv16_fwd + (a) y may be NULL when the operand planes out_hi (, out_lo) are requested and Cout % 8 == 0 (the consumer only reads the
// planes: no fp32 store); (b) stats [rows][Cout][3] | NULL: {count, mean, M2} of the written values per (row block, channel), merged by
// lp_norm_stats_finalize -- *stats_rows (HOST int, out) = partial rows per image, 0 when this geometry is not covered (small / ragged
// maps, split-K) and the caller has to run lp_instnorm_stats / lp_bn_train_stats on y instead.
extern "C" int lp_conv16_fwd_stats(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                                   const float* bias, const float* res, const float* alpha, const float* alpha2,
                                   int N, int H, int W, int Cin, int Cout, int CinP, int CoutP,
                                   int ksize, int upsample, int res_shift, int prec, const uint16_t* relu_mask16,
                                   uint16_t* out_hi, uint16_t* out_lo, int out_relu, float* workspace, long long workspace_bytes, float* amax_slots,
                                   float* stats, long long stats_capacity_floats, int* stats_rows, void* stream) {
    if (stats_rows) *stats_rows = 0;
    if (!a_hi || !w_hi) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: null pointer");
    if (!y && (!out_hi || (Cout & 7))) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: y may only be NULL when out_hi is given and Cout % 8 == 0");
    if (stats && !stats_rows) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: stats needs stats_rows");
    if (prec == LP_PREC_BF16X3 && (!w_lo || !a_lo)) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: bf16x3 needs the lo planes");
    if (prec == LP_PREC_BF16X3 && out_hi && !out_lo) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: bf16x3 needs out_lo");
    if ((upsample == 1 || upsample == 2) && ((H | W) & 1)) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: upsampled output dims must be even");
    if (upsample < 0 || upsample > 3) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: upsample = 0 | 1 | 2 (phase form) | 3 (phase form of the data gradient)");
    if (CinP % 32 || CinP < Cin || CoutP % 128 || CoutP < Cout) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: bad padded dims");
    if (H < 2 || W < 2) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_conv16_fwd: H,W must be >= 2");
    Conv16Params p;
    p.a_hi = a_hi; p.a_lo = a_lo; p.w_hi = w_hi; p.w_lo = w_lo; p.y = y; p.bias = bias; p.res = res; p.alpha = alpha; p.alpha2 = alpha2;
    p.mask16 = relu_mask16; p.o_relu = out_relu; p.part = workspace; p.part_bytes = workspace ? workspace_bytes : 0; p.amax = amax_slots;
    p.o_hi = (Cout & 7) ? nullptr : out_hi; p.o_lo = (Cout & 7) ? nullptr : out_lo;   // (pad channels: the pack pass writes them)
    p.N = N; p.H = H; p.W = W; p.Hin = upsample ? H / 2 : H; p.Win = upsample ? W / 2 : W;
    p.Cin = Cin; p.C8 = (Cin + 7) & ~7; p.Cout = Cout; p.Co8 = (Cout + 7) & ~7; p.CinP = CinP; p.CoutP = CoutP; p.res_shift = res_shift;
    p.grouped = 0; p.stats = stats; p.stats_cap = stats ? stats_capacity_floats : 0; p.stats_rows = 0;
    p.phase = (upsample == 2) ? 1 : (upsample == 3) ? 2 : 0;
    if (upsample >= 2 && ksize != 3) return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: upsample = 2 | 3 (phase-decomposed forms) are 3x3 forms");
    if (upsample == 3) { p.Hin = 2 * H; p.Win = 2 * W; }          // data gradient: a = dy planes [N][2H][2W][C8], output = the low-resolution dx [N][H][W][Cout]
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (prec == LP_PREC_BF16) rc = dispatch_conv16<LP_PREC_BF16>(p, ksize, upsample, s);
    else if (prec == LP_PREC_BF16X3) rc = dispatch_conv16<LP_PREC_BF16X3>(p, ksize, upsample, s);
    else if (prec == LP_PREC_F16) rc = dispatch_conv16<LP_PREC_F16>(p, ksize, upsample, s);
    else return lp_set_error(LP_ERR_ARG, "lp_conv16_fwd: unknown precision mode");
    if (rc) return rc;
    if (p.ksplit > 1) {
        const long long items = (long long)N * H * W * (Cout >> 2);
        long long blocks = (items + 255) / 256; if (blocks > 2048) blocks = 2048;
        if (prec == LP_PREC_BF16) hipLaunchKernelGGL(splitk_reduce_kernel<LP_PREC_BF16>, dim3((unsigned)blocks), dim3(256), 0, s, p);
        else if (prec == LP_PREC_BF16X3) hipLaunchKernelGGL(splitk_reduce_kernel<LP_PREC_BF16X3>, dim3((unsigned)blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(splitk_reduce_kernel<LP_PREC_F16>, dim3((unsigned)blocks), dim3(256), 0, s, p);
        rc = lp_check_launch("splitk_reduce");
        if (rc) return rc;
    }
    if (stats_rows) *stats_rows = p.stats_rows;
    // channel counts the epilogue writes element-wise (Cout % 8 != 0: padding channels) get their 16-bit planes from a
    // bandwidth-bound pass over the finished y instead (same stream)
    if (out_hi && !p.o_hi)
        return lp_act_pack(y, nullptr, nullptr, (out_relu & 1) ? 2 : 0, out_hi, out_lo, N, H * W, Cout, prec, nullptr, nullptr, 0, 1, nullptr, stream);
    return LP_OK;
}

// Grouped 3x3 conv (ResNeXt conv2, embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26 = torchvision resnext50_32x4d:
// 32 groups of 4 / 8 / 16 / 32 channels) as a BLOCK-DIAGONAL dense conv: every group lies inside one aligned 64-channel block, so the
// workgroup that owns output channels [co0, co0 + 64) contracts over input channels [co0, co0 + 64) only -- two 32-channel chunks of the
// same LDS-DMA / MFMA pipeline, with a weight image [9][CP][64] whose off-group entries are zero (lp_pack_grouped).  The matrix pipe
// does 64 / group-size times the algorithmic work, which is irrelevant: the layer is bound by its activation traffic (2 B read +
// 4 B written per element).  Stride 2 and the data gradient of stride 2 are expressed by the caller (full-resolution conv +
// lp_subsample2 / lp_zero_stuff2_16).  With the mode-1 pack and a = dY this is the data-gradient kernel.
extern "C" int lp_gconv16_fwd(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                              const float* alpha2, int N, int H, int W, int C, int CP, int prec, float* amax_slots, void* stream) {
    return lp_gconv16_fwd_stats(a_hi, a_lo, w_hi, w_lo, y, alpha2, N, H, W, C, CP, prec, amax_slots, nullptr, 0, nullptr, stream);
}

extern "C" int lp_gconv16_fwd_stats(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                                    const float* alpha2, int N, int H, int W, int C, int CP, int prec, float* amax_slots,
                                    float* stats, long long stats_capacity_floats, int* stats_rows, void* stream) {
    if (!y) return lp_set_error(LP_ERR_ARG, "lp_gconv16_fwd: null pointer");
    return lp_gconv16_fwd_planes(a_hi, a_lo, w_hi, w_lo, y, nullptr, nullptr, alpha2, N, H, W, C, CP, 0, prec, amax_slots, stats,
                                 stats_capacity_floats, stats_rows, stream);
}

// y (fp32, |NULL) and / or the operand planes of y (o_hi [, o_lo], |NULL): the fp16 mode keeps the embedder's conv outputs 16-bit resident
extern "C" int lp_gconv16_fwd_planes(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                                     uint16_t* o_hi, uint16_t* o_lo, const float* alpha2, int N, int H, int W, int C, int CP, int group_size,
                                     int prec, float* amax_slots, float* stats, long long stats_capacity_floats, int* stats_rows,
                                     void* stream) {
    if (stats_rows) *stats_rows = 0;
    if (stats && !stats_rows) return lp_set_error(LP_ERR_ARG, "lp_gconv16_fwd: stats needs stats_rows");
    if (!a_hi || !w_hi || (!y && !o_hi)) return lp_set_error(LP_ERR_ARG, "lp_gconv16_fwd: null pointer");
    if (o_hi && prec == LP_PREC_BF16X3 && !o_lo) return lp_set_error(LP_ERR_ARG, "lp_gconv16_fwd: bf16x3 output planes need lo");
    if (prec == LP_PREC_BF16X3 && (!w_lo || !a_lo)) return lp_set_error(LP_ERR_ARG, "lp_gconv16_fwd: bf16x3 needs the lo planes");
    if ((C & 63) || CP % 128 || CP < C) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_gconv16_fwd: C must be a multiple of 64, CP of 128");
    if (H < 2 || W < 2) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_gconv16_fwd: H,W must be >= 2");
    Conv16Params p;
    p.a_hi = a_hi; p.a_lo = a_lo; p.w_hi = w_hi; p.w_lo = w_lo; p.y = y; p.bias = nullptr; p.res = nullptr; p.alpha = nullptr; p.alpha2 = alpha2;
    p.mask16 = nullptr; p.o_relu = 0; p.part = nullptr; p.part_bytes = 0; p.amax = amax_slots; p.o_hi = o_hi; p.o_lo = o_lo;
    p.N = N; p.H = H; p.W = W; p.Hin = H; p.Win = W;
    p.Cin = C; p.C8 = C; p.Cout = C; p.Co8 = C; p.CinP = 64; p.CoutP = CP; p.res_shift = 0; p.grouped = 1; p.phase = 0;
    p.stats = stats; p.stats_cap = stats ? stats_capacity_floats : 0; p.stats_rows = 0;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // group_size (0: unknown) in 1 .. 32: the two 32-channel halves of a 64-channel block do not interact -- the kernel variant that skips
    // the zero half of every stage (LP_GCONV_HALF=0: the full block-diagonal product, the A/B baseline)
    static const bool half_env = !(getenv("LP_GCONV_HALF") && atoi(getenv("LP_GCONV_HALF")) == 0);
    const bool gh = half_env && group_size >= 1 && group_size <= 32 && (32 % group_size) == 0;
    if (prec == LP_PREC_BF16) rc = gh ? launch_conv16<3, false, 4, 1, 4, 4, LP_PREC_BF16, 32, true>(p, s) : launch_conv16<3, false, 4, 1, 4, 4, LP_PREC_BF16>(p, s);
    else if (prec == LP_PREC_BF16X3) rc = gh ? launch_conv16<3, false, 4, 1, 4, 4, LP_PREC_BF16X3, 32, true>(p, s) : launch_conv16<3, false, 4, 1, 4, 4, LP_PREC_BF16X3>(p, s);
    else if (prec == LP_PREC_F16) rc = gh ? launch_conv16<3, false, 4, 1, 4, 4, LP_PREC_F16, 32, true>(p, s) : launch_conv16<3, false, 4, 1, 4, 4, LP_PREC_F16>(p, s);
    else return lp_set_error(LP_ERR_ARG, "lp_gconv16_fwd: unknown precision mode");
    if (stats_rows) *stats_rows = p.stats_rows;
    return rc;
}

// upper bound of the statistics buffer of lp_conv16_fwd_stats / lp_gconv16_fwd_stats in floats: one row block per 64 output pixels
// (+ one tile of slack for the paired ping-pong grid) x Cout x 3
extern "C" long long lp_conv16_stats_floats(int N, int H, int W, int Cout) {
    return ((long long)N * H * W / 64 + 16) * Cout * 3;
}
