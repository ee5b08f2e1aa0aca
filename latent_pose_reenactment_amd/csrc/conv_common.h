// Shared pieces of the implicit-GEMM convolution kernels (conv_dma.hip: vmcnt(0) + barrier ping-pong / ring loops; conv_pipe.hip: the
// counted-vmcnt tap pipeline): the launch parameter block and the epilogue that turns a wave's accumulator block into the outputs.
#pragma once
#include "lp_common.h"
#include <stdlib.h>

struct Conv16Params {
    const uint16_t* a_hi; const uint16_t* a_lo; const uint16_t* w_hi; const uint16_t* w_lo;
    float* y;
    const float* bias; const float* res; const float* alpha; const float* alpha2;
    const uint16_t* mask16;       // epilogue: y = 0 where this 16-bit activation plane [N][H][W][Co8] is <= 0 (fused ReLU backward)
    uint16_t* o_hi; uint16_t* o_lo;   // optional: 16-bit planes [N][H][W][Co8] of (o_relu ? relu(y) : y) for the consumer conv
    int o_relu;                   // bit 0: the emitted planes hold relu(y); bit 1 (round 6): y itself is stored as relu(y) (conv + pool + the next block's in-place ReLU)
    float* amax;                            // LP_AMAX_SLOTS pre-zeroed floats | NULL: fold max|y| in (y will become an fp16 gradient operand)
    float* part; long long part_bytes;      // split-K partial sums [ksplit][N*H*W][Cout] (caller's workspace)
    int N, H, W, Hin, Win, Cin, C8, Cout, Co8, CinP, CoutP;
    int res_shift;
    int lTH, lTW, lNB, tiles_x, tiles_y;
    int hit;                      // halo DMA instructions per wave and chunk
    int a_dbuf;                   // single-group schedule: halo double buffered (1) or one buffer + an extra barrier per chunk (0)
    int ksplit;
    float* stats;                 // NULL | per-(row block, channel) {count, mean, M2} of the values this launch writes: [rows][Cout][3], row block
                                  // = (tile index * WM + wave row) -- what lp_norm_stats_finalize merges (instance / batch norm of y with no
                                  // extra pass over it).  Host-checked: every wave's rows lie in one image, tiles cover the images exactly.
    long long stats_cap;          // (host) capacity of `stats` in floats
    int stats_rows;               // (host, out) partial rows per image the launch wrote (0: none -- geometry not covered, caller runs the stats pass)
    int xcd_map, ntiles, nco;     // xcd_map: 1-D grid of 8 * ceil(ntiles / 8) * nco workgroups; workgroup id -> XCD id & 7 (round-robin dispatch), and on
                                  // that XCD the Cout blocks of ONE pixel tile run back to back, the XCD's tiles being a contiguous range: the halo
                                  // a tile's nco workgroups share (and the rows neighbouring tiles share) is fetched from HBM once, into one L2
    int phase;                    // 1: PHASE-DECOMPOSED x2-upsampled 3x3 conv (round 6; KS = 2 kernels): output pixel (2y + a, 2x + b) of the conv over the
                                  // nearest-upsampled input only meets the 2 x 2 low-resolution pixels (y + a - 1 + i, x + b - 1 + j), with the
                                  // 3 x 3 taps that fall on the same pixel pre-summed -- 4/9 of the matrix work, exact algebra.  Tiles cover the
                                  // LOW-resolution grid (Hin x Win) once per phase (tile index: x, y, phase, image); the weight image holds
                                  // [4 phases][2 x 2 taps][CoutP][CinP]; the epilogue writes to the phase's position of the H x W output
    int grouped;                  // block-diagonal (grouped) conv: the workgroup's 64 output channels only see input channels co0 .. co0+63;
                                  // the weight image then has 64 columns (CinP = 64) and the activation channel offset is co0
};

static inline int ilog2_floor(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }

// workgroup id -> (pixel tile | pair of tiles, Cout block); false: the workgroup lies in the padding of the XCD-ordered grid
__device__ __forceinline__ bool conv16_block(const Conv16Params& p, int& tile, int& cob) {
    if (!p.xcd_map) { tile = (int)blockIdx.x; cob = (int)blockIdx.y; return true; }
    const int id = (int)blockIdx.x, x = id & 7, j = id >> 3;
    if (p.xcd_map == 2) { tile = id % p.ntiles; cob = id / p.ntiles; return cob < p.nco; }      // (debug: 1-D grid, plain order)
    const int per = (p.ntiles + 7) >> 3;          // tiles per XCD
    cob = j % p.nco;
    const int k = j / p.nco;
    tile = x * per + k;
    return k < per && tile < p.ntiles;
}

static inline dim3 conv16_grid(Conv16Params& p, int ntiles, int nco, int gz = 1) {
    static const int on = getenv("LP_CONV_XCD") ? atoi(getenv("LP_CONV_XCD")) : 1;
    p.ntiles = ntiles; p.nco = nco;
    p.xcd_map = (on && nco > 1 && ntiles >= 16) ? on : 0;
    if (!p.xcd_map) return dim3(ntiles, nco, gz);
    return dim3(((ntiles + 7) / 8) * 8 * nco, 1, gz);
}

// choose NB x TH x TW = BM with TH<=H, TW<=W (powers of two), preferring wide patches (TW up to 16)
static inline void choose_tile(int BM, int N, int H, int W, int* lTH, int* lTW, int* lNB) {
    int lbm = ilog2_floor(BM);
    int ltw = ilog2_floor(W); if (ltw > 4) ltw = 4;
    int lth = ilog2_floor(H); if (lth > lbm - ltw) lth = lbm - ltw;
    if (lth < 1) lth = 1;
    if (ltw < 1) ltw = 1;
    int lnb = lbm - ltw - lth; if (lnb < 0) lnb = 0;
    while (lnb > 0 && (1 << (lnb - 1)) >= N) --lnb;      // no more images per tile than exist (keeps the LDS halo small)
    *lTH = lth; *lTW = ltw; *lNB = lnb;
}

// conv_pipe.hip: the tap-pipelined 3x3 kernel.  -> 1: launched; 0: layer not covered (run conv_dma_kernel); < 0: error
int lp_conv_pipe_launch(Conv16Params& p, int ups, int prec, hipStream_t s);
int lp_conv1x1_pipe_launch(Conv16Params& p, int prec, hipStream_t s);      // the chunk-pipelined 1x1 kernel

// Epilogue of a conv workgroup.  acc[MR][NR]: the wave's (MR*16) x (NR*16) block in the MFMA 16x16 C layout (col = lane&15 = channel,
// row = (lane>>4)*4 + reg = tile row); wave (wm, wn) of a WM x WN group whose tile starts at image n0, pixel (y0, x0), channel co0; `wave_d`
// = the wave's index among ALL waves of the workgroup (its slice of the LDS transpose scratch); `tile_index` = linear tile index of the
// group (row block of the statistics partials = tile_index * WM + wm).  MRP = MFMA row blocks transposed through LDS at a time (the scratch
// is waves x MRP*16 x (NR*16+4) floats at the start of `smem`, which must be dead: the function begins with a workgroup barrier).
// y = acc*alpha*alpha2 + bias + res[n, y>>rs, x>>rs], ReLU-mask from 16-bit planes, optional fp32 y, operand planes of (relu?)(y), amax,
// {count, mean, M2} partials, or the raw split-K partial tile.  Must be called by every wave of the workgroup.
template <int WM, int WN, int MR, int NR, int PREC, int MRP>
__device__ __forceinline__ void conv16_epilogue(const Conv16Params& p, f32x4_t (&acc)[MR][NR], unsigned char* smem, int wave_d, int wm, int wn,
                                                int lane, int n0, int y0, int x0, int co0, int NBv, int tile_index, int ph = -1) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    float alpha = p.alpha ? *p.alpha : 1.f;
    if (p.alpha2) alpha *= *p.alpha2;
    if ((p.Cout & 3) == 0) {
        // Coalesced path: every wave transposes its (MR*16) x (NR*16) accumulator block through LDS (the staging buffers are dead
        // now) and writes whole pixel rows -- NR*64 contiguous bytes per pixel, 16 B per lane; bias, residual, mask likewise.
        // 1x1 layers transpose 16 rows (one MFMA row block) at a time: their stage buffers are small, and the full-block scratch
        // (17 KB per wave) was what limited a CU to two workgroups (+10..20 % on the K <= 256 pointwise layers); the 3x3 kernels keep
        // the whole block in one piece (their stage buffers are larger than the scratch, and one long run of independent LDS reads
        // and stores measured ~1.5 % faster per step than four short ones) -- profiles/r03_conv_epilogue_lds.txt.
        constexpr int WR = MR * 16, WC = NR * 16, LDW = WC + 4;
        __syncthreads();                                                  // all waves are done with the halo / weight buffers
        float* tile = (float*)smem + wave_d * (MRP * 16 * LDW);
        constexpr int C4 = WC / 4;                 // float4 columns per row
        constexpr int RPP = 64 / C4;               // rows per pass of the wave
        const int c4 = lane % C4, rsub = lane / C4;
        const int co = co0 + wn * WC + c4 * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && co < p.Cout) bv = *(const float4*)(p.bias + co);
        float am = 0.f;
        float st_n = 0.f, st_ref[4] = {0.f, 0.f, 0.f, 0.f}, st_d[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mp = 0; mp < MR; mp += MRP) {
#pragma unroll
            for (int mr = mp; mr < mp + MRP; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        tile[((mr - mp) * 16 + (lane >> 4) * 4 + r) * LDW + nr * 16 + (lane & 15)] = acc[mr][nr][r];
            // (the wave reads back what its own lanes wrote: LDS operations of one wave execute in order)
#pragma unroll 4
        for (int r0 = mp * 16; r0 < (mp + MRP) * 16; r0 += RPP) {
            const int row = r0 + rsub;
            const int m = wm * WR + row;
            int nb, py, px;
            tile_row_linear(m, p.lTH, p.lTW, nb, py, px);
            const int n = n0 + nb;
            int oyy = y0 + py, oxx = x0 + px;
            bool inside = nb < NBv && n < p.N && co < p.Cout;
            if (ph >= 0) { inside = inside && oyy < p.Hin && oxx < p.Win; oyy = 2 * oyy + (ph >> 1); oxx = 2 * oxx + (ph & 1); }      // (phase mode: tile rows are low-resolution positions)
            else inside = inside && oyy < p.H && oxx < p.W;
            if (inside) {
                float4 v = *(const float4*)(tile + (row - mp * 16) * LDW + c4 * 4);
                if (p.ksplit > 1) {        // split-K: the raw partial tile; splitk_reduce_kernel sums the slices and applies the epilogue
                    const size_t pixs = (size_t)(n * p.H + oyy) * p.W + oxx;
                    *(float4*)(p.part + ((size_t)blockIdx.z * p.N * p.H * p.W + pixs) * p.Cout + co) = v;
                    continue;
                }
                v.x = fmaf(v.x, alpha, bv.x); v.y = fmaf(v.y, alpha, bv.y); v.z = fmaf(v.z, alpha, bv.z); v.w = fmaf(v.w, alpha, bv.w);
                if (p.res) {
                    const float4 rv = *(const float4*)(p.res + ((size_t)(n * (p.H >> p.res_shift) + (oyy >> p.res_shift)) * (p.W >> p.res_shift)
                                                                + (oxx >> p.res_shift)) * p.Cout + co);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                const size_t pix = (size_t)(n * p.H + oyy) * p.W + oxx;
                if (p.mask16) {
                    const ushort4 mv = *(const ushort4*)(p.mask16 + pix * p.Co8 + co);      // > 0  <=>  sign clear and magnitude non-zero
                    v.x = (mv.x - 1u) < 0x7fffu ? v.x : 0.f; v.y = (mv.y - 1u) < 0x7fffu ? v.y : 0.f;
                    v.z = (mv.z - 1u) < 0x7fffu ? v.z : 0.f; v.w = (mv.w - 1u) < 0x7fffu ? v.w : 0.f;
                }
                if (p.o_relu & 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (p.y) *(float4*)(p.y + pix * p.Cout + co) = v;
                am = lp_amax4(am, v);
                if (p.stats) {                     // shifted sums (reference = the lane's first value): no cancellation at large |mean| / std
                    const float o4[4] = {v.x, v.y, v.z, v.w};
                    if (st_n == 0.f) { st_ref[0] = v.x; st_ref[1] = v.y; st_ref[2] = v.z; st_ref[3] = v.w; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d = o4[j] - st_ref[j]; st_d[j] += d; st_q[j] = fmaf(d, d, st_q[j]); }
                    st_n += 1.f;
                }
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
        }
        }
        if (p.amax && p.ksplit == 1) lp_amax_commit(am, p.amax, blockIdx.x + blockIdx.y * 7u);
        if (p.stats && p.ksplit == 1) {
            // lane -> (count, mean, M2) of its 4 channels over its rows; the RPP lanes that share the channel quad (lane bits above C4)
            // merge pairwise (Chan et al.); rsub == 0 writes the wave's partial: row block = tile * WM + wm
            float mean[4], m2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mean[j] = 0.f; m2[j] = 0.f;
                if (st_n > 0.f) { mean[j] = st_ref[j] + st_d[j] / st_n; m2[j] = fmaxf(st_q[j] - st_d[j] * st_d[j] / st_n, 0.f); }
            }
#pragma unroll
            for (int o = C4; o < 64; o <<= 1) {
                const float nb_ = __shfl_xor(st_n, o, 64);
                float mb[4], qb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { mb[j] = __shfl_xor(mean[j], o, 64); qb[j] = __shfl_xor(m2[j], o, 64); }
                const float nn = st_n + nb_;
                if (nn > 0.f) {
                    const float fb = nb_ / nn, fab = st_n * fb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d = mb[j] - mean[j]; mean[j] = fmaf(d, fb, mean[j]); m2[j] += qb[j] + d * d * fab; }
                }
                st_n = nn;
            }
            if (rsub == 0 && co < p.Cout) {
                float* o = p.stats + ((size_t)(tile_index * WM + wm) * p.Cout + co) * 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j * 3 + 0] = st_n; o[j * 3 + 1] = mean[j]; o[j * 3 + 2] = m2[j]; }
            }
        }
        return;
    }
    float am_s = 0.f;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = wm * (MR * 16) + mr * 16 + (lane >> 4) * 4 + r;
            int nb, py, px;
            tile_row_linear(m, p.lTH, p.lTW, nb, py, px);
            const int n = n0 + nb, oyy = y0 + py, oxx = x0 + px;
            if (nb >= NBv || n >= p.N || oyy >= p.H || oxx >= p.W) continue;
            const size_t pixi = (size_t)(n * p.H + oyy) * p.W + oxx;
            const size_t pix = pixi * p.Cout;
            size_t rpix = 0;
            if (p.res) rpix = ((size_t)(n * (p.H >> p.res_shift) + (oyy >> p.res_shift)) * (p.W >> p.res_shift) + (oxx >> p.res_shift)) * p.Cout;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int co = co0 + wn * (NR * 16) + nr * 16 + (lane & 15);
                if (co < p.Cout) {
                    float v = acc[mr][nr][r] * alpha;      // (element-wise path: channel counts that are no multiple of 4; never split-K)
                    if (p.bias) v += p.bias[co];
                    if (p.res) v += p.res[rpix + co];
                    if (p.mask16 && !((unsigned)(p.mask16[pixi * p.Co8 + co] - 1u) < 0x7fffu)) v = 0.f;
                    if (p.y) p.y[pix + co] = v;
                    am_s = fmaxf(am_s, fabsf(v));
                    if (p.o_hi) {
                        const float q = (p.o_relu & 1) ? fmaxf(v, 0.f) : v;
                        const uint16_t h = lp_f32_to_op16<F16>(q);
                        p.o_hi[pixi * p.Co8 + co] = h;
                        if (SPLIT) p.o_lo[pixi * p.Co8 + co] = lp_f32_to_op16<false>(q - lp_op16_to_f32<false>(h));
                    }
                }
            }
        }
    }
    if (p.amax) lp_amax_commit(am_s, p.amax, blockIdx.x + blockIdx.y * 7u);
}
