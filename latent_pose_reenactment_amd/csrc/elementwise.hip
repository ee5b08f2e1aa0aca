// HBM-bound helper kernels of the generator path (gfx950): weight packing, instance-norm statistics, AdaIN/ReLU
// backward, 2x2 block sums, generator head.  All NHWC fp32, float4-vectorised, coalesced along channels.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------------------------
// modes 2 / 3 (round 6): the PHASE form of a x2-upsampled 3x3 conv (T = 9 in, 16 taps out: t = phase * 4 + i * 2 + j, phase = 2a + b).  Output pixel
// (2y + a, 2x + b) of conv3x3(nearest_up2(x)) reads the low-resolution pixels (y + a - 1 + i, x + b - 1 + j); the 3x3 taps that meet the same
// pixel are summed here, in fp32, before the 16-bit split: rows {0} | {1, 2} for a = 0, {0, 1} | {2} for a = 1 (columns alike).
//   mode 2 (forward):        out[t][co][ci] = sum of w[co][ci][dy][dx] over the taps of (phase, i, j)
//   mode 3 (data gradient):  out[t][ci][co] = the (phase, 1 - i, 1 - j) sum  (dx_lo[y] = sum_a sum_i Wp[a][i]^T dy_a[y - (a - 1 + i)]: a 2 x 2
//                            conv per phase over the phase-subsampled dy with origin y - a and flipped taps)
__device__ __forceinline__ float lp_phase_tap_sum(const float* __restrict__ w9, int t16, bool flip) {
    const int ph = t16 >> 2, a = ph >> 1, b = ph & 1;
    int i = (t16 >> 1) & 1, j = t16 & 1;
    if (flip) { i = 1 - i; j = 1 - j; }
    const int r0 = (a == 0) ? (i == 0 ? 0 : 1) : (i == 0 ? 0 : 2), r1 = (a == 0) ? (i == 0 ? 0 : 2) : (i == 0 ? 1 : 2);
    const int c0 = (b == 0) ? (j == 0 ? 0 : 1) : (j == 0 ? 0 : 2), c1 = (b == 0) ? (j == 0 ? 0 : 2) : (j == 0 ? 1 : 2);
    float v = 0.f;
    for (int r = r0; r <= r1; ++r)
        for (int c = c0; c <= c1; ++c) v += w9[r * 3 + c];
    return v;
}

// modes 4 / 5 (round 6): a 3x3 conv FOLLOWED BY AvgPool2d(2) (the critic's down blocks, discriminators/no_landmarks.py:52-81 via blocks.py:76-90) is a
// 4x4 stride-2 conv: pooled[y] = 1/4 sum_{a in {0,1}} conv[2y + a] = sum_{u = -1..2} W4[u] in[2y + u],  W4[u] = 1/4 sum of w[dy] over dy = u - a + 1 in [0, 2]
// (rows; columns alike) -- 16 taps per pooled output instead of 4 x 9: 4/9 of the matrix work, and the un-pooled conv output never exists.
// It runs on the phase kernels: hi-res offset u = 2 ky - pa of (phase pa, tap ky) in the gather form (lp_conv16_fwd upsample = 3), and its
// data gradient on the scatter-free phase-forward form (upsample = 2): hi-res pixel 2y + a collects W4[u_a(i)]^T d_pooled[y + a - 1 + i],
// u_0 = (2, 0), u_1 = (1, -1).
//   mode 4 (forward):        out[t = (pa, pb, ky, kx)][co][ci] = W4[2 ky - pa][2 kx - pb]
//   mode 5 (data gradient):  out[t = (a, b, i, j)][ci][co]     = W4[u_a(i)][u_b(j)]
__device__ __forceinline__ float lp_pool_tap_sum(const float* __restrict__ w9, int u, int v) {
    float s = 0.f;
    for (int dy = (u > 0 ? u : 0); dy <= (u + 1 < 2 ? u + 1 : 2); ++dy)
        for (int dx = (v > 0 ? v : 0); dx <= (v + 1 < 2 ? v + 1 : 2); ++dx) s += w9[dy * 3 + dx];
    return 0.25f * s;
}

__device__ __forceinline__ float lp_pack_value(const float* __restrict__ w, int Cout, int Cin, int T, int mode, int t, int row, int col) {
    if (mode == 0) return (row < Cout && col < Cin) ? w[((size_t)row * Cin + col) * T + t] : 0.f;                 // [t][co][ci]
    if (mode == 1) return (row < Cin && col < Cout) ? w[((size_t)col * Cin + row) * T + (T - 1 - t)] : 0.f;     // [T-1-t][ci][co]
    if (mode == 2) return (row < Cout && col < Cin) ? lp_phase_tap_sum(w + ((size_t)row * Cin + col) * 9, t, false) : 0.f;
    if (mode == 3) return (row < Cin && col < Cout) ? lp_phase_tap_sum(w + ((size_t)col * Cin + row) * 9, t, true) : 0.f;
    if (mode == 4) {
        const int pa = (t >> 3) & 1, pb = (t >> 2) & 1, ky = (t >> 1) & 1, kx = t & 1;
        return (row < Cout && col < Cin) ? lp_pool_tap_sum(w + ((size_t)row * Cin + col) * 9, 2 * ky - pa, 2 * kx - pb) : 0.f;
    }
    const int a = (t >> 3) & 1, b = (t >> 2) & 1, i = (t >> 1) & 1, j = t & 1;
    return (row < Cin && col < Cout) ? lp_pool_tap_sum(w + ((size_t)col * Cin + row) * 9, (a ? 1 : 2) - 2 * i, (b ? 1 : 2) - 2 * j) : 0.f;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                    int Cout, int Cin, int T, int RowsP, int ColsP, int mode, int f16) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)(mode >= 2 ? 16 : T) * RowsP * ColsP;
    if (idx >= total) return;
    int col = (int)(idx % ColsP);
    int row = (int)((idx / ColsP) % RowsP);
    int t = (int)(idx / ((long long)ColsP * RowsP));
    const float v = lp_pack_value(w, Cout, Cin, T, mode, t, row, col);
    if (f16) { hi[idx] = lp_f32_to_op16<true>(v); return; }
    __bf16 h = (__bf16)v;
    hi[idx] = __builtin_bit_cast(uint16_t, h);
    if (lo) { __bf16 l = (__bf16)(v - (float)h); lo[idx] = __builtin_bit_cast(uint16_t, l); }
}

// batched variant: blockIdx.y = entry of a device table (one per (weight, orientation)); one launch re-packs every conv weight
// of a module after an optimizer step
// (a flat 1-D grid of 1024-element chunks: entry sizes differ by 1000x, so a (blocks, entries) grid either starves the big
//  entries or floods the scheduler with empty blocks; chunk0 = first chunk of the entry, found by binary search)
struct PackDesc { const float* w; uint16_t* hi; uint16_t* lo; int Cout, Cin, T, RowsP, ColsP, mode, chunk0, f16; };

__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PackDesc* __restrict__ table, int num_entries) {
    int lo_e = 0, hi_e = num_entries - 1;
    while (lo_e < hi_e) {                                  // last entry whose chunk0 <= blockIdx.x   (block-uniform)
        const int mid = (lo_e + hi_e + 1) >> 1;
        if (table[mid].chunk0 <= (int)blockIdx.x) lo_e = mid; else hi_e = mid - 1;
    }
    const PackDesc d = table[lo_e];
    const long long total = (long long)(d.mode >= 2 ? 16 : d.T) * d.RowsP * d.ColsP;
    const long long base = (long long)((int)blockIdx.x - d.chunk0) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long idx = base + k * 256 + threadIdx.x;
        if (idx >= total) break;
        int col = (int)(idx % d.ColsP);
        int row = (int)((idx / d.ColsP) % d.RowsP);
        int t = (int)(idx / ((long long)d.ColsP * d.RowsP));
        const float v = lp_pack_value(d.w, d.Cout, d.Cin, d.T, d.mode, t, row, col);
        if (d.f16) { d.hi[idx] = lp_f32_to_op16<true>(v); continue; }
        __bf16 h = (__bf16)v;
        d.hi[idx] = __builtin_bit_cast(uint16_t, h);
        if (d.lo) { __bf16 l = (__bf16)(v - (float)h); d.lo[idx] = __builtin_bit_cast(uint16_t, l); }
    }
}

// Both orientations of a conv weight from ONE coalesced read: a block stages a 32 (co) x 32 (ci) x T tile of W [Cout][Cin][T] in LDS
// (each co row is a contiguous 32*T-float run) and writes the forward pack [t][co][ci] (ci fastest) and the data-gradient pack
// [T-1-t][ci][co] (co fastest) as 64-byte runs, zero padding included.  The per-element kernels above read W with a 36-byte stride
// (forward) or a Cin*T-float stride (data gradient): 0.34 ms per step for the generator's 27 M weights; this form is bandwidth bound.
struct PackPairDesc {
    const float* w; uint16_t* hi0; uint16_t* lo0; uint16_t* hi1; uint16_t* lo1;
    int Cout, Cin, T, RowsP0, ColsP0, RowsP1, ColsP1, tile0, tiles_ci, f16;
};

__global__ __launch_bounds__(256) void pack_pair_kernel(const PackPairDesc* __restrict__ table, int num_entries) {
    __shared__ float tile[32][32 * 9 + 1];
    int lo_e = 0, hi_e = num_entries - 1;
    while (lo_e < hi_e) {                                  // last entry whose tile0 <= blockIdx.x   (block-uniform)
        const int mid = (lo_e + hi_e + 1) >> 1;
        if (table[mid].tile0 <= (int)blockIdx.x) lo_e = mid; else hi_e = mid - 1;
    }
    const PackPairDesc d = table[lo_e];
    const int tl = (int)blockIdx.x - d.tile0;
    const int co0 = (tl / d.tiles_ci) * 32, ci0 = (tl % d.tiles_ci) * 32;
    const int T = d.T, row = 32 * T;
    for (int e = threadIdx.x; e < 32 * row; e += 256) {
        const int co_l = e / row, r = e - co_l * row;
        const int ci_l = r / T;
        float v = 0.f;
        if (co0 + co_l < d.Cout && ci0 + ci_l < d.Cin) v = d.w[((size_t)(co0 + co_l) * d.Cin + ci0) * T + r];
        tile[co_l][r] = v;
    }
    __syncthreads();
    auto put = [&](uint16_t* hi, uint16_t* lo, size_t idx, float v) {
        if (d.f16) { hi[idx] = lp_f32_to_op16<true>(v); return; }
        const __bf16 h = (__bf16)v;
        hi[idx] = __builtin_bit_cast(uint16_t, h);
        if (lo) { const __bf16 l = (__bf16)(v - (float)h); lo[idx] = __builtin_bit_cast(uint16_t, l); }
    };
    for (int e = threadIdx.x; e < 32 * row; e += 256) {    // forward pack: [t][co][ci], ci fastest
        const int ci_l = e & 31, co_l = (e >> 5) & 31, t = e >> 10;
        if (t < T && co0 + co_l < d.RowsP0 && ci0 + ci_l < d.ColsP0)
            put(d.hi0, d.lo0, ((size_t)t * d.RowsP0 + co0 + co_l) * d.ColsP0 + ci0 + ci_l, tile[co_l][ci_l * T + t]);
    }
    for (int e = threadIdx.x; e < 32 * row; e += 256) {    // data-gradient pack: [T-1-t][ci][co], co fastest
        const int co_l = e & 31, ci_l = (e >> 5) & 31, t = e >> 10;
        if (t < T && ci0 + ci_l < d.RowsP1 && co0 + co_l < d.ColsP1)
            put(d.hi1, d.lo1, ((size_t)(T - 1 - t) * d.RowsP1 + ci0 + ci_l) * d.ColsP1 + co0 + co_l, tile[co_l][ci_l * T + t]);
    }
}

extern "C" int lp_pack_pair_desc_bytes(void) { return (int)sizeof(PackPairDesc); }

extern "C" int lp_pack_weights_pairs(const void* table, int num_entries, long long total_tiles, void* stream) {
    if (!table || num_entries <= 0 || total_tiles < 1) return lp_set_error(LP_ERR_ARG, "lp_pack_weights_pairs: bad arguments");
    hipLaunchKernelGGL(pack_pair_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const PackPairDesc*)table, num_entries);
    return lp_check_launch("pack_weights_pairs");
}

extern "C" int lp_pack_desc_bytes(void) { return (int)sizeof(PackDesc); }

extern "C" int lp_pack_weights_batch(const void* table, int num_entries, long long total_chunks, void* stream) {
    if (!table || num_entries <= 0) return lp_set_error(LP_ERR_ARG, "lp_pack_weights_batch: bad arguments");
    if (total_chunks < 1) return lp_set_error(LP_ERR_ARG, "lp_pack_weights_batch: no chunks");
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)table,
                       num_entries);
    return lp_check_launch("pack_weights_batch");
}

extern "C" int lp_pack_weights(const float* w, uint16_t* hi, uint16_t* lo, int Cout, int Cin, int T, int RowsP, int ColsP, int mode,
                               int f16, void* stream) {
    if (!w || !hi) return lp_set_error(LP_ERR_ARG, "lp_pack_weights: null pointer");
    if (mode < 0 || mode > 5 || (mode >= 2 && T != 9)) return lp_set_error(LP_ERR_ARG, "lp_pack_weights: mode 0 | 1, or 2 .. 5 (phase / conv + pool forms of a 3x3 weight: T = 9)");
    int rows = (mode == 0 || mode == 2 || mode == 4) ? Cout : Cin, cols = (mode == 0 || mode == 2 || mode == 4) ? Cin : Cout;
    if (RowsP < rows || ColsP < cols || (ColsP & 7)) return lp_set_error(LP_ERR_ARG, "lp_pack_weights: bad padded dims");
    long long total = (long long)(mode >= 2 ? 16 : T) * RowsP * ColsP;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, hi, lo, Cout, Cin, T,
                       RowsP, ColsP, mode, f16);
    return lp_check_launch("pack_weights");
}

// ------------------------------------------------------------------------------------------------------------------
// instance-norm statistics -> AdaIN scale/shift
//   stage 1: grid (splits, C/64 blocks, N); block = 16 channel-quads x 16 pixel lanes; per-thread shifted sums,
//            merged with Chan's parallel-variance formula (no E[x^2]-E[x]^2 cancellation).
//   stage 2: one thread per (n,c) merges the splits in fp64 and emits mean, rstd, scale, shift.
// ------------------------------------------------------------------------------------------------------------------
// pixels per stage-1 block: lp_stat_split_pix (lp_common.h) -- 1024 for the 256^2 maps (4096 blocks), down to 64 for the small ones

__device__ __forceinline__ void chan_merge(float& n_a, float& mean_a, float& m2_a, float n_b, float mean_b, float m2_b) {
    if (n_b == 0.f) return;
    if (n_a == 0.f) { n_a = n_b; mean_a = mean_b; m2_a = m2_b; return; }
    float n = n_a + n_b, d = mean_b - mean_a;
    mean_a += d * (n_b / n);
    m2_a += m2_b + d * d * (n_a * n_b / n);
    n_a = n;
}

__global__ __launch_bounds__(256) void instnorm_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                               int S, int PB) {
    __shared__ float sh[3][16][64];
    const int n = blockIdx.z, cb = blockIdx.y, s = blockIdx.x;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = cb * 64 + cq * 4;
    const int p0 = s * PB, p1 = min(HW, p0 + PB);
    float cnt = 0.f, ref[4] = {0, 0, 0, 0}, sd[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    if (c < C) {
        const float* base = x + (size_t)n * HW * C + c;
        const bool vec = (C & 3) == 0;
#pragma unroll 4
        for (int pix = p0 + pl; pix < p1; pix += 16) {
            float v[4] = {0, 0, 0, 0};
            if (vec) { float4 q = *(const float4*)(base + (size_t)pix * C); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
            else { for (int j = 0; j < 4; ++j) if (c + j < C) v[j] = base[(size_t)pix * C + j]; }
            if (cnt == 0.f) { for (int j = 0; j < 4; ++j) ref[j] = v[j]; }
            for (int j = 0; j < 4; ++j) { float d = v[j] - ref[j]; sd[j] += d; sq[j] += d * d; }
            cnt += 1.f;
        }
    }
    for (int j = 0; j < 4; ++j) {
        float mean = 0.f, m2 = 0.f;
        if (cnt > 0.f) { mean = ref[j] + sd[j] / cnt; m2 = sq[j] - sd[j] * sd[j] / cnt; }
        sh[0][pl][cq * 4 + j] = cnt; sh[1][pl][cq * 4 + j] = mean; sh[2][pl][cq * 4 + j] = m2;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        int ch = threadIdx.x;
        float na = 0.f, ma = 0.f, qa = 0.f;
        for (int k = 0; k < 16; ++k) chan_merge(na, ma, qa, sh[0][k][ch], sh[1][k][ch], sh[2][k][ch]);
        int cg = cb * 64 + ch;
        if (cg < C) {
            float* o = part + (((size_t)n * S + s) * C + cg) * 3;
            o[0] = na; o[1] = ma; o[2] = qa;
        }
    }
}

// one WAVE per (n, c): the lanes split the S partials (a BatchNorm over 10^6 positions has 1024 of them), merge pairwise in fp64.
// WPC = 4 (round 6, S >= 256: the train-mode BatchNorms of the identity encoder's stem / first two stages leave 512 .. 8192 partials per channel
// and have 64 .. 512 channels): one WORKGROUP per (n, c), its four waves' sums folded in order through LDS -- these launches sit between every
// conv and its consumer and were a few dozen waves walking 32 .. 128 rounds of strided loads each (8 us mean, 90 us for the stem).
template <int WPC = 1>
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         int ab_stride, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                         float* __restrict__ scale, float* __restrict__ shift, int N, int C, int S,
                                         float* __restrict__ run_mean = nullptr, float* __restrict__ run_var = nullptr, float momentum = 0.f) {
    const int idx = WPC == 4 ? (int)blockIdx.x : blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = WPC == 4 ? (int)threadIdx.x : (threadIdx.x & 63), LANES = 64 * WPC;
    if (idx >= N * C) return;
    int n = idx / C, c = idx % C;
    // {count, mean, M2} partials -> shifted sums about ONE reference (the first partial's mean, the same for all lanes):
    //   N = sum n_s,  S1 = sum n_s (mean_s - ref),  S2 = sum [M2_s + n_s (mean_s - ref)^2]   (fp64 FMAs, no division per partial)
    //   mean = ref + S1 / N,  M2 = S2 - S1^2 / N.   |mean_s - ref| is a few standard deviations at most, so the final subtraction is benign
    //   in fp64; the lanes' sums simply add (6 shuffle steps).  (The pairwise Chan merge this replaces spent two fp64 divisions per partial:
    //   11 us for the 4096 partials per channel of the embedder's first stage.)
    const double ref = part[((size_t)n * S * C + c) * 3 + 1];
    double na = 0, s1 = 0, s2 = 0;
#pragma unroll 4
    for (int s = lane; s < S; s += LANES) {
        const float* q = part + (((size_t)n * S + s) * C + c) * 3;
        const double nb = q[0], d = (double)q[1] - ref;
        na += nb; s1 = fma(nb, d, s1); s2 += (double)q[2] + nb * d * d;
    }
    for (int o = 32; o > 0; o >>= 1) { na += __shfl_down(na, o, 64); s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
    if (WPC == 4) {          // the four waves' sums, in wave order
        __shared__ double wsum[4][3];
        if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6][0] = na; wsum[threadIdx.x >> 6][1] = s1; wsum[threadIdx.x >> 6][2] = s2; }
        __syncthreads();
        na = ((wsum[0][0] + wsum[1][0]) + wsum[2][0]) + wsum[3][0];
        s1 = ((wsum[0][1] + wsum[1][1]) + wsum[2][1]) + wsum[3][1];
        s2 = ((wsum[0][2] + wsum[1][2]) + wsum[2][2]) + wsum[3][2];
    }
    const double ma = ref + s1 / na;
    double qa = s2 - s1 * s1 / na;
    if (qa < 0) qa = 0;
    if (lane != 0) return;
    float var = (float)(qa / na);
    float m = (float)ma, r = 1.0f / sqrtf(var + eps);
    mean[idx] = m; rstd[idx] = r;
    if (run_mean) {        // nn.BatchNorm2d: running_mean / running_var (UNBIASED batch variance) <- (1 - momentum) * old + momentum * batch
        const float unb = na > 1 ? (float)(qa / (na - 1)) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
    }
    if (scale) {
        float g = gamma ? gamma[(size_t)n * ab_stride + c] : 1.f, b = beta ? beta[(size_t)n * ab_stride + c] : 0.f;
        float sc = r * g;
        scale[idx] = sc; shift[idx] = b - m * sc;
    }
}

extern "C" long long lp_instnorm_workspace_bytes(int N, int HW, int C) {
    const int PB = lp_stat_split_pix(N, HW, C);
    long long S = (HW + PB - 1) / PB;
    return (long long)N * S * C * 3 * 4;
}

extern "C" int lp_instnorm_stats(const float* x, const float* gamma, const float* beta, int ab_stride, float eps, float* mean,
                                 float* rstd, float* scale, float* shift, float* workspace, int N, int HW, int C, void* stream) {
    if (!x || !mean || !rstd || !workspace) return lp_set_error(LP_ERR_ARG, "lp_instnorm_stats: null pointer");
    const int PB = lp_stat_split_pix(N, HW, C);
    int S = (HW + PB - 1) / PB;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_partial_kernel, dim3(S, (C + 63) / 64, N), dim3(256), 0, st, x, workspace, HW, C, S, PB);
    int rc = lp_check_launch("instnorm_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(instnorm_finalize_kernel<1>, dim3(cdiv((long long)N * C, 4)), dim3(256), 0, st, workspace, gamma, beta, ab_stride,
                       eps, mean, rstd, scale, shift, N, C, S, nullptr, nullptr, 0.f);
    return lp_check_launch("instnorm_finalize");
}

// Finish of the statistics a conv launch left behind (lp_conv16_fwd_stats): part [N][S][C][3] = {count, mean, M2} per (row block, channel)
// -> mean, rstd (biased variance), scale = gamma*rstd, shift = beta - mean*scale per (n, c); N = 1 with running_mean/var: a train-mode
// BatchNorm (momentum update, unbiased variance).  gamma/beta [N][C] with row stride ab_stride.
extern "C" int lp_norm_stats_finalize(const float* part, int S, const float* gamma, const float* beta, int ab_stride, float eps, float momentum,
                                      float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                                      int N, int C, void* stream) {
    if (!part || !mean || !rstd || S < 1) return lp_set_error(LP_ERR_ARG, "lp_norm_stats_finalize: null pointer");
    if (!running_mean != !running_var || (running_mean && N != 1)) return lp_set_error(LP_ERR_ARG, "lp_norm_stats_finalize: running statistics need N == 1");
    if (S >= 256 && (long long)N * C <= 65535)
        hipLaunchKernelGGL(instnorm_finalize_kernel<4>, dim3(N * C), dim3(256), 0, (hipStream_t)stream, part, gamma, beta, ab_stride,
                           eps, mean, rstd, scale, shift, N, C, S, running_mean, running_var, momentum);
    else
        hipLaunchKernelGGL(instnorm_finalize_kernel<1>, dim3(cdiv((long long)N * C, 4)), dim3(256), 0, (hipStream_t)stream, part, gamma, beta, ab_stride,
                           eps, mean, rstd, scale, shift, N, C, S, running_mean, running_var, momentum);
    return lp_check_launch("norm_stats_finalize");
}

// train-mode nn.BatchNorm2d over y [P][C]: batch statistics (biased variance for the normalisation), scale = gamma*rstd,
// shift = beta - mean*scale, and the momentum update of the running statistics -- the instance-norm kernels with N = 1, HW = P.
extern "C" long long lp_bn_train_stats_workspace_bytes(long long P, int C) { return lp_instnorm_workspace_bytes(1, (int)P, C); }

extern "C" int lp_bn_train_stats(const float* y, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                 float* running_var, float* mean, float* rstd, float* scale, float* shift, float* workspace,
                                 long long P, int C, void* stream) {
    if (!y || !gamma || !beta || !mean || !rstd || !scale || !shift || !workspace) return lp_set_error(LP_ERR_ARG, "lp_bn_train_stats: null pointer");
    if (!running_mean != !running_var) return lp_set_error(LP_ERR_ARG, "lp_bn_train_stats: running_mean and running_var go together");
    if (P < 1 || P >= (1ll << 30)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_train_stats: 1 <= P < 2^30");
    const int HW = (int)P;
    const int PB = lp_stat_split_pix(1, HW, C);
    const int S = (HW + PB - 1) / PB;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_partial_kernel, dim3(S, (C + 63) / 64, 1), dim3(256), 0, st, y, workspace, HW, C, S, PB);
    int rc = lp_check_launch("bn_stats_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(instnorm_finalize_kernel<1>, dim3(cdiv(C, 4)), dim3(256), 0, st, workspace, gamma, beta, C, eps, mean, rstd, scale, shift,
                       1, C, S, running_mean, running_var, momentum);
    return lp_check_launch("bn_stats_finalize");
}

// ------------------------------------------------------------------------------------------------------------------
// backward of relu(AdaIN(x)) [+ x2 nearest upsample]
//   pass 1: g = sum2x2?(dA) * mask  -> written into dx (as temporary), partial sums S1 = sum g, S2 = sum g*xhat
//   pass 2: coefficients; pass 3: dx = ca*g + cb*x + cc (+ add)
// ------------------------------------------------------------------------------------------------------------------
// mask_mode 0: the activation's own pattern 0 < x*scale+shift < act_hi (ReLU: act_hi = inf; ReLU6: 6);  1: no mask (a plain norm);
// 2: mask_src > 0 (the ReLU sits behind a residual add: mask_src = the block output).  g_copy (|NULL): the masked gradient g is also
// written there (the identity branch of the residual block receives exactly that).
// X16: x is a 16-BIT-RESIDENT conv output (the unscaled fp16 plane [N][HW][C] the conv epilogue left instead of fp32 y; C % 8 == 0)
template <bool X16> __device__ __forceinline__ float4 lp_ldx4(const void* x, size_t e) {
    if (X16) {
        const ushort4 h = *(const ushort4*)((const uint16_t*)x + e);
        return make_float4(lp_op16_to_f32<true>(h.x), lp_op16_to_f32<true>(h.y), lp_op16_to_f32<true>(h.z), lp_op16_to_f32<true>(h.w));
    }
    return *(const float4*)((const float*)x + e);
}

// four consecutive channels of a gradient tensor as bf16 operand planes: hi = RNE bf16, lo = bf16(v - hi) (|NULL: plain bf16 mode)
__device__ __forceinline__ void lp_store_bf16_planes4(const float4& o, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, size_t e) {
    const float v[4] = {o.x, o.y, o.z, o.w};
    ushort4 h, l;
    uint16_t* hp = (uint16_t*)&h; uint16_t* lp = (uint16_t*)&l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hp[j] = lp_f32_to_op16<false>(v[j]);
        lp[j] = lp_f32_to_op16<false>(v[j] - lp_op16_to_f32<false>(hp[j]));
    }
    *(ushort4*)(hi + e) = h;
    if (lo) *(ushort4*)(lo + e) = l;
}

template <bool X16 = false>
__global__ __launch_bounds__(256) void adain_bwd_partial_kernel(const float* __restrict__ dA, const void* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                float* __restrict__ g_out, float* __restrict__ part, int H, int W, int C,
                                                                int ups, int S, int PB, int mask_mode = 0, const float* __restrict__ mask_src = nullptr,
                                                                float* __restrict__ g_copy = nullptr, float act_hi = 3.0e38f) {
    __shared__ float sh[2][16][64];
    const int n = blockIdx.z, cb = blockIdx.y, s = blockIdx.x;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = cb * 64 + cq * 4;
    const int HW = H * W;
    const int p0 = s * PB, p1 = min(HW, p0 + PB);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (c < C) {                                       // C % 4 == 0 required (checked on host)
        float4 mu = *(const float4*)(mean + (size_t)n * C + c), rs = *(const float4*)(rstd + (size_t)n * C + c);
        float4 sc = *(const float4*)(scale + (size_t)n * C + c), sf = *(const float4*)(shift + (size_t)n * C + c);
#pragma unroll 2
        for (int pix = p0 + pl; pix < p1; pix += 16) {
            float4 xv = lp_ldx4<X16>(x, ((size_t)n * HW + pix) * C + c);
            float4 g;
            if (ups) {
                int yy = pix / W, xx = pix % W;
                const float* b = dA + (((size_t)n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
                float4 a0 = *(const float4*)b, a1 = *(const float4*)(b + C), a2 = *(const float4*)(b + (size_t)2 * W * C),
                       a3 = *(const float4*)(b + (size_t)2 * W * C + C);
                g.x = (a0.x + a1.x) + (a2.x + a3.x); g.y = (a0.y + a1.y) + (a2.y + a3.y);
                g.z = (a0.z + a1.z) + (a2.z + a3.z); g.w = (a0.w + a1.w) + (a2.w + a3.w);
            } else g = *(const float4*)(dA + ((size_t)n * HW + pix) * C + c);
            if (mask_mode == 0) {
                const float a0 = fmaf(xv.x, sc.x, sf.x), a1 = fmaf(xv.y, sc.y, sf.y), a2 = fmaf(xv.z, sc.z, sf.z), a3 = fmaf(xv.w, sc.w, sf.w);
                g.x = (a0 > 0.f && a0 < act_hi) ? g.x : 0.f; g.y = (a1 > 0.f && a1 < act_hi) ? g.y : 0.f;
                g.z = (a2 > 0.f && a2 < act_hi) ? g.z : 0.f; g.w = (a3 > 0.f && a3 < act_hi) ? g.w : 0.f;
            } else if (mask_mode == 2) {
                const float4 mk = *(const float4*)(mask_src + ((size_t)n * HW + pix) * C + c);
                g.x = mk.x > 0.f ? g.x : 0.f; g.y = mk.y > 0.f ? g.y : 0.f; g.z = mk.z > 0.f ? g.z : 0.f; g.w = mk.w > 0.f ? g.w : 0.f;
            }
            *(float4*)(g_out + ((size_t)n * HW + pix) * C + c) = g;
            if (g_copy) *(float4*)(g_copy + ((size_t)n * HW + pix) * C + c) = g;
            s1[0] += g.x; s1[1] += g.y; s1[2] += g.z; s1[3] += g.w;
            s2[0] += g.x * ((xv.x - mu.x) * rs.x); s2[1] += g.y * ((xv.y - mu.y) * rs.y);
            s2[2] += g.z * ((xv.z - mu.z) * rs.z); s2[3] += g.w * ((xv.w - mu.w) * rs.w);
        }
    }
    for (int j = 0; j < 4; ++j) { sh[0][pl][cq * 4 + j] = s1[j]; sh[1][pl][cq * 4 + j] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
        float a = 0.f;
        for (int k = 0; k < 16; ++k) a += sh[which][k][ch];
        int cg = cb * 64 + ch;
        if (cg < C) part[(((size_t)n * S + s) * C + cg) * 2 + which] = a;
    }
}

__global__ __launch_bounds__(256) void adain_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, int ab_stride,
                                          const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, float* __restrict__ coef, int N, int C, int S, float inv_hw,
                                          int frozen = 0) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;          // one wave per (n, c)
    if (idx >= N * C) return;
    int n = idx / C, c = idx % C;
    double a1 = 0, a2 = 0;
    for (int s = lane; s < S; s += 64) { const float* q = part + (((size_t)n * S + s) * C + c) * 2; a1 += q[0]; a2 += q[1]; }
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64); }
    if (lane != 0) return;
    float S1 = (float)a1, S2 = (float)a2;
    if (dgamma) dgamma[(size_t)n * ab_stride + c] = S2;
    if (dbeta) dbeta[(size_t)n * ab_stride + c] = S1;
    float g = gamma ? gamma[(size_t)n * ab_stride + c] : 1.f;
    float r = rstd[idx], m = mean[idx];
    float ca = g * r;
    float cb = frozen ? 0.f : -ca * r * S2 * inv_hw;          // frozen: the statistics are constants (running statistics, eval mode)
    float cc = frozen ? 0.f : -ca * S1 * inv_hw - cb * m;
    coef[(size_t)idx * 3 + 0] = ca; coef[(size_t)idx * 3 + 1] = cb; coef[(size_t)idx * 3 + 2] = cc;
}

// o_hi (, o_lo) | NULL: the result ALSO goes out as bf16 (hi + lo) operand planes [N][HW][C] (C % 8 == 0) -- what lp_act_pack(grad) would write in
// the bf16 / bf16x3 modes (no gradient scale there) -- and with keep_dx == 0 ONLY as planes (round 6: the generator's backward in its bf16x3 default;
// a conv's gradient operand then costs no fp32 round trip and no pack launch)
template <bool X16 = false>
__global__ void adain_bwd_apply_kernel(float* __restrict__ dx /* holds g */, const void* __restrict__ x, const float* __restrict__ add,
                                       const float* __restrict__ coef, long long total4, int HW, int C, float* __restrict__ amax,
                                       uint16_t* __restrict__ o_hi = nullptr, uint16_t* __restrict__ o_lo = nullptr, int keep_dx = 1) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C4 = C >> 2;
    float am = 0.f;
    for (; i < total4; i += stride) {
        int c = (int)(i % C4) * 4;
        int n = (int)(i / ((long long)C4 * HW));
        const float* cf = coef + ((size_t)n * C + c) * 3;
        float4 g = ((const float4*)dx)[i], xv = lp_ldx4<X16>(x, (size_t)i * 4), o;
        o.x = fmaf(cf[0], g.x, fmaf(cf[1], xv.x, cf[2]));
        o.y = fmaf(cf[3], g.y, fmaf(cf[4], xv.y, cf[5]));
        o.z = fmaf(cf[6], g.z, fmaf(cf[7], xv.z, cf[8]));
        o.w = fmaf(cf[9], g.w, fmaf(cf[10], xv.w, cf[11]));
        if (add) { float4 a = ((const float4*)add)[i]; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
        if (keep_dx) ((float4*)dx)[i] = o;
        if (o_hi) lp_store_bf16_planes4(o, o_hi, o_lo, (size_t)i * 4);
        am = lp_amax4(am, o);
    }
    if (amax) lp_amax_commit(am, amax, blockIdx.x);
}

extern "C" long long lp_adain_bwd_workspace_bytes(int N, int HW, int C) {
    const int PB = lp_stat_split_pix(N, HW, C);
    long long S = (HW + PB - 1) / PB;
    return (long long)N * S * C * 2 * 4 + (long long)N * C * 3 * 4;
}

extern "C" int lp_adain_relu_bwd(const float* dA, const float* x, const float* add, const float* gamma, int ab_stride, const float* mean,
                                 const float* rstd, const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta,
                                 float* workspace, int N, int H, int W, int C, int upsample, float* amax_slots, void* stream) {
    return lp_norm_act_bwd(dA, x, add, gamma, ab_stride, mean, rstd, scale, shift, dx, dgamma, dbeta, workspace, N, H, W, C, upsample, 0,
                           nullptr, nullptr, 0.f, 0, amax_slots, stream);
}

static int norm_act_bwd_impl(const float* dA, const void* x, int x16, const float* add, const float* gamma, int ab_stride, const float* mean,
                             const float* rstd, const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta,
                             float* workspace, int N, int H, int W, int C, int upsample, int mask_mode, const float* mask_src,
                             float* g_copy, float act_hi, int frozen_stats, float* amax_slots, void* stream,
                             uint16_t* o_hi = nullptr, uint16_t* o_lo = nullptr, int keep_dx = 1);

// lp_adain_relu_bwd whose result goes out as bf16 (out_lo == NULL) / bf16 hi + lo operand planes [N][H][W][C] (C % 8 == 0) -- the gradient operand
// of the conv below, in the bf16 / bf16x3 modes (no gradient scale) -- and, with keep_dx != 0, also as fp32 dx.  dx must always be given: it holds
// the masked gradient between the passes.
extern "C" int lp_adain_relu_bwd_planes(const float* dA, const float* x, const float* add, const float* gamma, int ab_stride, const float* mean,
                                        const float* rstd, const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta,
                                        float* workspace, int N, int H, int W, int C, int upsample, uint16_t* out_hi, uint16_t* out_lo,
                                        int keep_dx, void* stream) {
    if (!out_hi) return lp_set_error(LP_ERR_ARG, "lp_adain_relu_bwd_planes: null pointer");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_adain_relu_bwd_planes: C must be a multiple of 8");
    return norm_act_bwd_impl(dA, x, 0, add, gamma, ab_stride, mean, rstd, scale, shift, dx, dgamma, dbeta, workspace, N, H, W, C, upsample, 0,
                             nullptr, nullptr, 0.f, 0, nullptr, stream, out_hi, out_lo, keep_dx);
}

// lp_adain_relu_bwd with x held as a 16-bit-resident conv output (fp16 plane [N][H][W][C], C % 8 == 0): the generator's fp16 mode keeps
// no fp32 copy of its conv outputs (round 5); x-hat and the ReLU pattern are recomputed from the SAME fp16 values the forward normalised
extern "C" int lp_adain_relu_bwd16(const float* dA, const uint16_t* x16, const float* add, const float* gamma, int ab_stride, const float* mean,
                                   const float* rstd, const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta,
                                   float* workspace, int N, int H, int W, int C, int upsample, float* amax_slots, void* stream) {
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_adain_relu_bwd16: C must be a multiple of 8");
    return norm_act_bwd_impl(dA, x16, 1, add, gamma, ab_stride, mean, rstd, scale, shift, dx, dgamma, dbeta, workspace, N, H, W, C, upsample, 0,
                             nullptr, nullptr, 0.f, 0, amax_slots, stream);
}

extern "C" int lp_norm_act_bwd(const float* dA, const float* x, const float* add, const float* gamma, int ab_stride, const float* mean,
                               const float* rstd, const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta,
                               float* workspace, int N, int H, int W, int C, int upsample, int mask_mode, const float* mask_src,
                               float* g_copy, float act_hi, int frozen_stats, float* amax_slots, void* stream) {
    return norm_act_bwd_impl(dA, x, 0, add, gamma, ab_stride, mean, rstd, scale, shift, dx, dgamma, dbeta, workspace, N, H, W, C, upsample, mask_mode,
                             mask_src, g_copy, act_hi, frozen_stats, amax_slots, stream);
}

static int norm_act_bwd_impl(const float* dA, const void* x, int x16, const float* add, const float* gamma, int ab_stride, const float* mean,
                             const float* rstd, const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta,
                             float* workspace, int N, int H, int W, int C, int upsample, int mask_mode, const float* mask_src,
                             float* g_copy, float act_hi, int frozen_stats, float* amax_slots, void* stream,
                             uint16_t* o_hi, uint16_t* o_lo, int keep_dx) {
    if (!dA || !x || !mean || !rstd || !scale || !shift || !dx || !workspace) return lp_set_error(LP_ERR_ARG, "lp_norm_act_bwd: null pointer");
    if (C & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_norm_act_bwd: C must be a multiple of 4");
    if (mask_mode < 0 || mask_mode > 2 || (mask_mode == 2 && !mask_src)) return lp_set_error(LP_ERR_ARG, "lp_norm_act_bwd: bad mask mode");
    if ((long long)H * W >= (1ll << 30)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_norm_act_bwd: H*W < 2^30");
    const int HW = H * W;
    const int PB = lp_stat_split_pix(N, HW, C);
    int S = (HW + PB - 1) / PB;
    float* part = workspace;
    float* coef = workspace + (size_t)N * S * C * 2;
    hipStream_t st = (hipStream_t)stream;
    if (x16) hipLaunchKernelGGL(adain_bwd_partial_kernel<true>, dim3(S, (C + 63) / 64, N), dim3(256), 0, st, dA, x, mean, rstd, scale, shift, dx, part,
                                H, W, C, upsample, S, PB, mask_mode, mask_src, g_copy, act_hi > 0.f ? act_hi : 3.0e38f);
    else hipLaunchKernelGGL(adain_bwd_partial_kernel<false>, dim3(S, (C + 63) / 64, N), dim3(256), 0, st, dA, x, mean, rstd, scale, shift, dx, part,
                            H, W, C, upsample, S, PB, mask_mode, mask_src, g_copy, act_hi > 0.f ? act_hi : 3.0e38f);
    int rc = lp_check_launch("adain_bwd_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(adain_bwd_finalize_kernel, dim3(cdiv((long long)N * C, 4)), dim3(256), 0, st, part, gamma, ab_stride, mean, rstd,
                       dgamma, dbeta, coef, N, C, S, 1.0f / (float)HW, frozen_stats);
    rc = lp_check_launch("adain_bwd_finalize");
    if (rc) return rc;
    long long total4 = (long long)N * HW * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 4096) blocks = 4096;
    if (x16) hipLaunchKernelGGL(adain_bwd_apply_kernel<true>, dim3(blocks), dim3(256), 0, st, dx, x, add, coef, total4, HW, C, amax_slots, o_hi, o_lo, keep_dx);
    else hipLaunchKernelGGL(adain_bwd_apply_kernel<false>, dim3(blocks), dim3(256), 0, st, dx, x, add, coef, total4, HW, C, amax_slots, o_hi, o_lo, keep_dx);
    return lp_check_launch("adain_bwd_apply");
}

// ------------------------------------------------------------------------------------------------------------------
// 2x2 block sum (adjoint of nearest x2 upsampling)
// ------------------------------------------------------------------------------------------------------------------
__global__ void sum2x2_kernel(const float* __restrict__ in, float* __restrict__ out, long long total4, int H, int W, int C,
                              float* __restrict__ amax, uint16_t* __restrict__ o_hi = nullptr, uint16_t* __restrict__ o_lo = nullptr) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C4 = C >> 2;
    float am = 0.f;
    for (; i < total4; i += stride) {
        int c = (int)(i % C4) * 4;
        long long pix = i / C4;
        int xx = (int)(pix % W); long long t = pix / W;
        int yy = (int)(t % H); int n = (int)(t / H);
        const float* b = in + (((size_t)n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
        float4 a0 = *(const float4*)b, a1 = *(const float4*)(b + C), a2 = *(const float4*)(b + (size_t)2 * W * C),
               a3 = *(const float4*)(b + (size_t)2 * W * C + C), o;
        o.x = (a0.x + a1.x) + (a2.x + a3.x); o.y = (a0.y + a1.y) + (a2.y + a3.y);
        o.z = (a0.z + a1.z) + (a2.z + a3.z); o.w = (a0.w + a1.w) + (a2.w + a3.w);
        if (out) ((float4*)out)[i] = o;
        if (o_hi) lp_store_bf16_planes4(o, o_hi, o_lo, (size_t)i * 4);
        am = lp_amax4(am, o);
    }
    if (amax) lp_amax_commit(am, amax, blockIdx.x);
}

// lp_sum2x2 straight to bf16 (out_lo == NULL) / bf16 hi + lo operand planes [N][H][W][C] (C % 8 == 0); out (fp32) | NULL
extern "C" int lp_sum2x2_planes(const float* in, float* out, uint16_t* out_hi, uint16_t* out_lo, int N, int H, int W, int C, void* stream) {
    if (!in || !out_hi) return lp_set_error(LP_ERR_ARG, "lp_sum2x2_planes: null pointer");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_sum2x2_planes: C must be a multiple of 8");
    long long total4 = (long long)N * H * W * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum2x2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, total4, H, W, C, (float*)nullptr, out_hi, out_lo);
    return lp_check_launch("sum2x2");
}

extern "C" int lp_sum2x2(const float* in, float* out, int N, int H, int W, int C, float* amax_slots, void* stream) {
    if (!in || !out) return lp_set_error(LP_ERR_ARG, "lp_sum2x2: null pointer");
    if (C & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_sum2x2: C must be a multiple of 4");
    long long total4 = (long long)N * H * W * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum2x2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, total4, H, W, C, amax_slots);
    return lp_check_launch("sum2x2");
}

// ------------------------------------------------------------------------------------------------------------------
// generator head: tanh + range shift + RGB x mask compositing (noBottleneck.py:170-181)
// ------------------------------------------------------------------------------------------------------------------
__global__ void head_fwd_kernel(const float* __restrict__ z, float* __restrict__ t, float* __restrict__ rgbs, float* __restrict__ segm,
                                int N, int HW) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * HW) return;
    int n = (int)(i / HW), pix = (int)(i % HW);
    float4 zv = ((const float4*)z)[i], tv;
    tv.x = tanhf(zv.x); tv.y = tanhf(zv.y); tv.z = tanhf(zv.z); tv.w = tanhf(zv.w);
    if (t) ((float4*)t)[i] = tv;
    float sg = tv.w * 0.5f + 0.5f;
    float* r = rgbs + (size_t)n * 3 * HW + pix;
    r[0] = (tv.x * 0.75f + 0.5f) * sg; r[HW] = (tv.y * 0.75f + 0.5f) * sg; r[2 * (size_t)HW] = (tv.z * 0.75f + 0.5f) * sg;
    segm[i] = sg;
}

__global__ void head_bwd_kernel(const float* __restrict__ t, const float* __restrict__ d_rgbs, const float* __restrict__ d_segm,
                                float* __restrict__ dz, int N, int HW, float* __restrict__ amax) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float am = 0.f;
    if (i < (long long)N * HW) {
        int n = (int)(i / HW), pix = (int)(i % HW);
        float4 tv = ((const float4*)t)[i];
        float sg = tv.w * 0.5f + 0.5f;
        const float* dr = d_rgbs + (size_t)n * 3 * HW + pix;
        float d0 = dr[0], d1 = dr[HW], d2 = dr[2 * (size_t)HW];
        float dsg = d0 * (tv.x * 0.75f + 0.5f) + d1 * (tv.y * 0.75f + 0.5f) + d2 * (tv.z * 0.75f + 0.5f);
        if (d_segm) dsg += d_segm[i];
        float4 o;
        o.x = d0 * sg * 0.75f * (1.f - tv.x * tv.x);
        o.y = d1 * sg * 0.75f * (1.f - tv.y * tv.y);
        o.z = d2 * sg * 0.75f * (1.f - tv.z * tv.z);
        o.w = dsg * 0.5f * (1.f - tv.w * tv.w);
        ((float4*)dz)[i] = o;
        am = lp_amax4(0.f, o);
    }
    if (amax) lp_amax_commit(am, amax, blockIdx.x);       // (all lanes: the wave reduction is a cross-lane shuffle)
}

extern "C" int lp_head_fwd(const float* z, float* t, float* fake_rgbs, float* fake_segm, int N, int H, int W, void* stream) {
    if (!z || !fake_rgbs || !fake_segm) return lp_set_error(LP_ERR_ARG, "lp_head_fwd: null pointer");
    long long total = (long long)N * H * W;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, z, t, fake_rgbs, fake_segm, N, H * W);
    return lp_check_launch("head_fwd");
}

extern "C" int lp_head_bwd(const float* t, const float* d_rgbs, const float* d_segm, float* dz, int N, int H, int W, float* amax_slots,
                           void* stream) {
    if (!t || !d_rgbs || !dz) return lp_set_error(LP_ERR_ARG, "lp_head_bwd: null pointer");
    long long total = (long long)N * H * W;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, t, d_rgbs, d_segm, dz, N, H * W, amax_slots);
    return lp_check_launch("head_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// discriminator / VGG helpers: ReLU backward, (ReLU +) 2x2 average pool forward/backward, L1 between ReLU'd features
// ------------------------------------------------------------------------------------------------------------------
__global__ void relu_bwd_kernel(const float4* __restrict__ dA, const float4* __restrict__ x, float4* __restrict__ dx, long long total4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total4; i += stride) {
        float4 g = dA[i], v = x[i];
        g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
        dx[i] = g;
    }
}

extern "C" int lp_relu_bwd(const float* dA, const float* x, float* dx, long long numel, void* stream) {
    if (!dA || !x || !dx) return lp_set_error(LP_ERR_ARG, "lp_relu_bwd: null pointer");
    if (numel & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_relu_bwd: numel must be a multiple of 4");
    long long total4 = numel / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)dA, (const float4*)x, (float4*)dx, total4);
    return lp_check_launch("relu_bwd");
}

// y[n,Y,X,c] = 0.25 * sum_{2x2} act(x[n,2Y+i,2X+j,c]),  act = relu if relu_in else identity.  H, W = OUTPUT dims.
// o16 != NULL (C % 8 == 0): also the 16-bit operand planes [N][H][W][C] of y for the conv that follows (fp16 when f16, else bf16)
// relu_out: y = relu(pool(.)) -- the ReLU the reference's NEXT block applies in place to its input (generators/common/blocks.py:71-73), so that
// every consumer of the pooled tensor (skip branch, feature list, next conv) sees relu(y): pool + ReLU + operand planes in ONE pass
__global__ void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long total4, int H, int W, int C, int relu_in,
                                    uint16_t* __restrict__ o16, int f16, int relu_out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C4 = C >> 2;
    for (; i < total4; i += stride) {
        int c = (int)(i % C4) * 4;
        long long pix = i / C4;
        int xx = (int)(pix % W); long long t = pix / W;
        int yy = (int)(t % H); int n = (int)(t / H);
        const float* b = x + (((size_t)n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
        float4 a0 = *(const float4*)b, a1 = *(const float4*)(b + C), a2 = *(const float4*)(b + (size_t)2 * W * C),
               a3 = *(const float4*)(b + (size_t)2 * W * C + C), o;
        if (relu_in) {
            a0.x = fmaxf(a0.x, 0.f); a0.y = fmaxf(a0.y, 0.f); a0.z = fmaxf(a0.z, 0.f); a0.w = fmaxf(a0.w, 0.f);
            a1.x = fmaxf(a1.x, 0.f); a1.y = fmaxf(a1.y, 0.f); a1.z = fmaxf(a1.z, 0.f); a1.w = fmaxf(a1.w, 0.f);
            a2.x = fmaxf(a2.x, 0.f); a2.y = fmaxf(a2.y, 0.f); a2.z = fmaxf(a2.z, 0.f); a2.w = fmaxf(a2.w, 0.f);
            a3.x = fmaxf(a3.x, 0.f); a3.y = fmaxf(a3.y, 0.f); a3.z = fmaxf(a3.z, 0.f); a3.w = fmaxf(a3.w, 0.f);
        }
        o.x = 0.25f * ((a0.x + a1.x) + (a2.x + a3.x)); o.y = 0.25f * ((a0.y + a1.y) + (a2.y + a3.y));
        o.z = 0.25f * ((a0.z + a1.z) + (a2.z + a3.z)); o.w = 0.25f * ((a0.w + a1.w) + (a2.w + a3.w));
        if (relu_out) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        ((float4*)y)[i] = o;
        if (o16) {
            ushort4 h;
            if (f16) { h.x = lp_f32_to_op16<true>(o.x); h.y = lp_f32_to_op16<true>(o.y); h.z = lp_f32_to_op16<true>(o.z); h.w = lp_f32_to_op16<true>(o.w); }
            else { h.x = lp_f32_to_op16<false>(o.x); h.y = lp_f32_to_op16<false>(o.y); h.z = lp_f32_to_op16<false>(o.z); h.w = lp_f32_to_op16<false>(o.w); }
            ((ushort4*)o16)[i] = h;
        }
    }
}

// the same on operand planes (no-grad chains that never need the fp32 tensors: the VGG stacks over the TARGET image): x16 [N][2H][2W][C] ->
// o16 [N][H][W][C], 8 channels per thread; the sum is taken in fp32 and rounded once
template <bool F16>
__global__ void avgpool2_fwd16_kernel(const uint16_t* __restrict__ x16, uint16_t* __restrict__ o16, long long total8, int H, int W, int C) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C8 = C >> 3;
    for (; i < total8; i += stride) {
        const int c = (int)(i % C8) * 8;
        long long pix = i / C8;
        const int xx = (int)(pix % W); long long t = pix / W;
        const int yy = (int)(t % H); const int n = (int)(t / H);
        const uint16_t* b = x16 + (((size_t)n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
        const s16x8_t a0 = *(const s16x8_t*)b, a1 = *(const s16x8_t*)(b + C), a2 = *(const s16x8_t*)(b + (size_t)2 * W * C),
                      a3 = *(const s16x8_t*)(b + (size_t)2 * W * C + C);
        s16x8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = 0.25f * ((lp_op16_to_f32<F16>((uint16_t)a0[j]) + lp_op16_to_f32<F16>((uint16_t)a1[j])) +
                                     (lp_op16_to_f32<F16>((uint16_t)a2[j]) + lp_op16_to_f32<F16>((uint16_t)a3[j])));
            o[j] = (short)lp_f32_to_op16<F16>(v);
        }
        *(s16x8_t*)(o16 + (size_t)i * 8) = o;
    }
}

extern "C" int lp_avgpool2_fwd16(const uint16_t* x_hi, uint16_t* out_hi, int N, int H, int W, int C, int prec, void* stream) {
    if (!x_hi || !out_hi) return lp_set_error(LP_ERR_ARG, "lp_avgpool2_fwd16: null pointer");
    if ((C & 7) || prec == LP_PREC_BF16X3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_avgpool2_fwd16: C % 8 == 0 and a one-plane precision mode");
    const long long total8 = (long long)N * H * W * C / 8;
    int blocks = (int)((total8 + 255) / 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    if (prec == LP_PREC_F16) hipLaunchKernelGGL(avgpool2_fwd16_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x_hi, out_hi, total8, H, W, C);
    else hipLaunchKernelGGL(avgpool2_fwd16_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x_hi, out_hi, total8, H, W, C);
    return lp_check_launch("avgpool2_fwd16");
}

// dx[n,y,x,c] = 0.25 * dy[n,y>>1,x>>1,c] * (relu_in ? [x>0] : 1).  H, W = INPUT (full-res) dims; one thread per 2x2 block.
// relu_out (flag bit 1 of the C entry point): the forward returned relu(pool(x)); `x` then points at that OUTPUT y [N][H/2][W/2][C] and dy is
// masked by [y > 0] first
__global__ void avgpool2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long long total4,
                                    int H, int W, int C, int relu_in, float* __restrict__ amax, int relu_out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    float am = 0.f;
    for (; i < total4; i += stride) {
        int c = (int)(i % C4) * 4;
        long long pix = i / C4;
        int xx = (int)(pix % Wo); long long t = pix / Wo;
        int yy = (int)(t % Ho); int n = (int)(t / Ho);
        float4 g = ((const float4*)dy)[i];
        g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
        if (relu_out) {
            const float4 yv = ((const float4*)x)[i];
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        size_t base = (((size_t)n * H + 2 * yy) * W + 2 * xx) * C + c;
        const size_t offs[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 o = g;
            if (relu_in) {
                float4 v = *(const float4*)(x + base + offs[k]);
                o.x = v.x > 0.f ? o.x : 0.f; o.y = v.y > 0.f ? o.y : 0.f; o.z = v.z > 0.f ? o.z : 0.f; o.w = v.w > 0.f ? o.w : 0.f;
            }
            *(float4*)(dx + base + offs[k]) = o;
            am = lp_amax4(am, o);
        }
    }
    if (amax) lp_amax_commit(am, amax, blockIdx.x);
}

// dx = 0.25 * dy[.., y>>1, x>>1, ..] * [m16 > 0]: the ReLU mask read from the operand planes of relu(x) (no fp32 x exists in the planes-only chains)
__global__ void avgpool2_bwd_m16_kernel(const float* __restrict__ dy, const uint16_t* __restrict__ m16, float* __restrict__ dx, long long total4,
                                        int H, int W, int C, float* __restrict__ amax) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    float am = 0.f;
    for (; i < total4; i += stride) {
        int c = (int)(i % C4) * 4;
        long long pix = i / C4;
        int xx = (int)(pix % Wo); long long t = pix / Wo;
        int yy = (int)(t % Ho); int n = (int)(t / Ho);
        float4 g = ((const float4*)dy)[i];
        g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
        size_t base = (((size_t)n * H + 2 * yy) * W + 2 * xx) * C + c;
        const size_t offs[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const ushort4 mv = *(const ushort4*)(m16 + base + offs[k]);          // (v - 1) < 0x7fff: a positive, non-zero 16-bit float
            float4 o;
            o.x = (mv.x - 1u) < 0x7fffu ? g.x : 0.f; o.y = (mv.y - 1u) < 0x7fffu ? g.y : 0.f;
            o.z = (mv.z - 1u) < 0x7fffu ? g.z : 0.f; o.w = (mv.w - 1u) < 0x7fffu ? g.w : 0.f;
            *(float4*)(dx + base + offs[k]) = o;
            am = lp_amax4(am, o);
        }
    }
    if (amax) lp_amax_commit(am, amax, blockIdx.x);
}

extern "C" int lp_avgpool2_bwd_m16(const float* dy, const uint16_t* mask_hi, float* dx, int N, int H, int W, int C, float* amax_slots, void* stream) {
    if (!dy || !dx || !mask_hi) return lp_set_error(LP_ERR_ARG, "lp_avgpool2_bwd_m16: null pointer");
    if ((C & 3) || (H & 1) || (W & 1)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_avgpool2_bwd_m16: C%4, H%2, W%2 must be 0");
    long long total4 = (long long)N * (H / 2) * (W / 2) * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(avgpool2_bwd_m16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, mask_hi, dx, total4, H, W, C, amax_slots);
    return lp_check_launch("avgpool2_bwd_m16");
}

extern "C" int lp_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, int relu_in, uint16_t* out_hi, int prec, void* stream) {
    if (out_hi && ((C & 7) || prec == LP_PREC_BF16X3)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_avgpool2_fwd: planes need C % 8 == 0 and a one-plane precision mode");
    if (!x || !y) return lp_set_error(LP_ERR_ARG, "lp_avgpool2_fwd: null pointer");
    if (C & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_avgpool2_fwd: C must be a multiple of 4");
    long long total4 = (long long)N * H * W * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, total4, H, W, C, relu_in & 1, out_hi,
                       prec == LP_PREC_F16 ? 1 : 0, (relu_in >> 1) & 1);
    return lp_check_launch("avgpool2_fwd");
}

extern "C" int lp_avgpool2_bwd(const float* dy, const float* x, float* dx, int N, int H, int W, int C, int relu_in, float* amax_slots,
                               void* stream) {
    if (!dy || !dx || (relu_in && !x)) return lp_set_error(LP_ERR_ARG, "lp_avgpool2_bwd: null pointer");
    if (relu_in == 3) return lp_set_error(LP_ERR_ARG, "lp_avgpool2_bwd: relu_in and relu_out masks are exclusive (one mask pointer)");
    if ((C & 3) || (H & 1) || (W & 1)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_avgpool2_bwd: C%4, H%2, W%2 must be 0");
    long long total4 = (long long)N * (H / 2) * (W / 2) * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, dx, total4, H, W, C, relu_in & 1, amax_slots,
                       (relu_in >> 1) & 1);
    return lp_check_launch("avgpool2_bwd");
}

// partial[b] = sum over this block's elements of |relu?(a) - relu?(b)|   (criterions/common/perceptual_loss.py:104-108 L1 taps)
// sgn != NULL: also the backward's sign pattern, one int8 per element: sign(relu?(a) - relu?(b)) * (relu_in ? [a > 0] : 1) -- the backward
// then reads 1 byte per element instead of a and b again (8 bytes).
// BMODE 0: b fp32 | 1: b = fp16 operand planes | 2: bf16 operand planes (the taps of the target image kept 16-bit, round 4)
// AMODE: the same for a (1 | 2: a = the operand planes of relu(a_true): the ReLU is already applied, [a_true > 0] == [plane > 0])
template <int BMODE, int AMODE = 0>
__global__ __launch_bounds__(256) void l1_partial_kernel(const void* __restrict__ av, const void* __restrict__ bv,
                                                         float* __restrict__ part, long long total4, int relu_in, char4* __restrict__ sgn) {
    __shared__ float sh[4];
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < total4; i += stride) {
        float4 u, v;
        if (AMODE == 0) u = ((const float4*)av)[i];
        else {
            const ushort4 q = ((const ushort4*)av)[i];
            u = make_float4(lp_op16_to_f32<AMODE == 1>(q.x), lp_op16_to_f32<AMODE == 1>(q.y), lp_op16_to_f32<AMODE == 1>(q.z), lp_op16_to_f32<AMODE == 1>(q.w));
        }
        if (BMODE == 0) v = ((const float4*)bv)[i];
        else {
            const ushort4 q = ((const ushort4*)bv)[i];
            v = make_float4(lp_op16_to_f32<BMODE == 1>(q.x), lp_op16_to_f32<BMODE == 1>(q.y), lp_op16_to_f32<BMODE == 1>(q.z), lp_op16_to_f32<BMODE == 1>(q.w));
        }
        if (relu_in) {
            u.x = fmaxf(u.x, 0.f); u.y = fmaxf(u.y, 0.f); u.z = fmaxf(u.z, 0.f); u.w = fmaxf(u.w, 0.f);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        s += (fabsf(u.x - v.x) + fabsf(u.y - v.y)) + (fabsf(u.z - v.z) + fabsf(u.w - v.w));
        if (sgn) {      // (with relu_in, u == 0 wherever a <= 0: then u > v is false and u < v needs the mask)
            char4 q;
            q.x = (char)((u.x > v.x) - ((u.x < v.x) && (!relu_in || u.x > 0.f)));
            q.y = (char)((u.y > v.y) - ((u.y < v.y) && (!relu_in || u.y > 0.f)));
            q.z = (char)((u.z > v.z) - ((u.z < v.z) && (!relu_in || u.z > 0.f)));
            q.w = (char)((u.w > v.w) - ((u.w < v.w) && (!relu_in || u.w > 0.f)));
            sgn[i] = q;
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// out[0] = coef * sum(part[0..n)) in a fixed order: the finished loss term (a device-wide "last block" ticket inside the partial
// kernel was tried instead of this second launch: 1024 same-address atomics made the partial kernel 3x slower)
__global__ __launch_bounds__(256) void l1_finalize_kernel(const float* __restrict__ part, int n, float coef, float* __restrict__ out) {
    __shared__ float sh[4];
    float t = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) t += part[j];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = coef * ((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// da = coef * g[0] * sign(relu?(a) - relu?(b)) * (relu_in ? [a>0] : 1)  (+ add: the gradient arriving at `a` from its other consumer)
__global__ void l1_bwd_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float* __restrict__ g, float coef,
                              const float4* __restrict__ add, float4* __restrict__ da, long long total4, int relu_in,
                              float* __restrict__ amax) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float k = coef * g[0];
    float am = 0.f;
    for (; i < total4; i += stride) {
        float4 u = a[i], v = b[i], o;
        float ux = relu_in ? fmaxf(u.x, 0.f) : u.x, uy = relu_in ? fmaxf(u.y, 0.f) : u.y, uz = relu_in ? fmaxf(u.z, 0.f) : u.z,
              uw = relu_in ? fmaxf(u.w, 0.f) : u.w;
        float vx = relu_in ? fmaxf(v.x, 0.f) : v.x, vy = relu_in ? fmaxf(v.y, 0.f) : v.y, vz = relu_in ? fmaxf(v.z, 0.f) : v.z,
              vw = relu_in ? fmaxf(v.w, 0.f) : v.w;
        o.x = (ux > vx ? k : (ux < vx ? -k : 0.f)); o.y = (uy > vy ? k : (uy < vy ? -k : 0.f));
        o.z = (uz > vz ? k : (uz < vz ? -k : 0.f)); o.w = (uw > vw ? k : (uw < vw ? -k : 0.f));
        if (relu_in) { o.x = u.x > 0.f ? o.x : 0.f; o.y = u.y > 0.f ? o.y : 0.f; o.z = u.z > 0.f ? o.z : 0.f; o.w = u.w > 0.f ? o.w : 0.f; }
        if (add) { const float4 e = add[i]; o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
        da[i] = o;
        am = lp_amax4(am, o);
    }
    if (amax) lp_amax_commit(am, amax, blockIdx.x);
}

// the same from the sign pattern the forward launch left: da = coef * g[0] * sgn (+ add)
__global__ void l1_bwd_sgn_kernel(const char4* __restrict__ sgn, const float* __restrict__ g, float coef, const float4* __restrict__ add,
                                  float4* __restrict__ da, long long total4, float* __restrict__ amax) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float k = coef * g[0];
    float am = 0.f;
    for (; i < total4; i += stride) {
        const char4 q = sgn[i];
        float4 o = make_float4(k * (float)q.x, k * (float)q.y, k * (float)q.z, k * (float)q.w);
        if (add) { const float4 e = add[i]; o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
        da[i] = o;
        am = lp_amax4(am, o);
    }
    if (amax) lp_amax_commit(am, amax, blockIdx.x);
}

#define L1_BLOCKS 1024
extern "C" int lp_l1_partial_blocks(void) { return L1_BLOCKS; }

extern "C" int lp_l1_fwd(const float* a, const float* b, float* partial, long long numel, int relu_in, float coef, float* out,
                         int8_t* sign_out, void* stream) {
    if (!a || !b || !partial) return lp_set_error(LP_ERR_ARG, "lp_l1_fwd: null pointer");
    if (numel & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_l1_fwd: numel must be a multiple of 4");
    hipLaunchKernelGGL(l1_partial_kernel<0>, dim3(L1_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const void*)a, (const void*)b, partial,
                       numel / 4, relu_in, (char4*)sign_out);
    if (out) hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, L1_BLOCKS, coef, out);
    return lp_check_launch("l1_fwd");
}

// the same with b given as 16-bit operand planes (same element order as a: channel counts that are multiples of 8)
extern "C" int lp_l1_fwd_b16(const float* a, const uint16_t* b_hi, int prec, float* partial, long long numel, int relu_in, float coef, float* out,
                             int8_t* sign_out, void* stream) {
    if (!a || !b_hi || !partial) return lp_set_error(LP_ERR_ARG, "lp_l1_fwd_b16: null pointer");
    if (numel & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_l1_fwd_b16: numel must be a multiple of 4");
    if (prec == LP_PREC_BF16X3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_l1_fwd_b16: one-plane precision modes only");
    if (prec == LP_PREC_F16)
        hipLaunchKernelGGL(l1_partial_kernel<1>, dim3(L1_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const void*)a, (const void*)b_hi, partial,
                           numel / 4, relu_in, (char4*)sign_out);
    else
        hipLaunchKernelGGL(l1_partial_kernel<2>, dim3(L1_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const void*)a, (const void*)b_hi, partial,
                           numel / 4, relu_in, (char4*)sign_out);
    if (out) hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, L1_BLOCKS, coef, out);
    return lp_check_launch("l1_fwd_b16");
}

// both operands as 16-bit operand planes of relu(.) (relu_in semantics: the sign pattern carries the [a > 0] mask)
extern "C" int lp_l1_fwd_ab16(const uint16_t* a_hi, const uint16_t* b_hi, int prec, float* partial, long long numel, float coef, float* out,
                              int8_t* sign_out, void* stream) {
    if (!a_hi || !b_hi || !partial) return lp_set_error(LP_ERR_ARG, "lp_l1_fwd_ab16: null pointer");
    if (numel & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_l1_fwd_ab16: numel must be a multiple of 4");
    if (prec == LP_PREC_BF16X3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_l1_fwd_ab16: one-plane precision modes only");
    if (prec == LP_PREC_F16)
        hipLaunchKernelGGL((l1_partial_kernel<1, 1>), dim3(L1_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const void*)a_hi, (const void*)b_hi, partial,
                           numel / 4, 1, (char4*)sign_out);
    else
        hipLaunchKernelGGL((l1_partial_kernel<2, 2>), dim3(L1_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const void*)a_hi, (const void*)b_hi, partial,
                           numel / 4, 1, (char4*)sign_out);
    if (out) hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, L1_BLOCKS, coef, out);
    return lp_check_launch("l1_fwd_ab16");
}

extern "C" int lp_l1_bwd(const float* a, const float* b, const float* grad_out, float coef, const float* add, float* da, long long numel,
                         int relu_in, const int8_t* sign, float* amax_slots, void* stream) {
    if ((!sign && (!a || !b)) || !grad_out || !da) return lp_set_error(LP_ERR_ARG, "lp_l1_bwd: null pointer");
    if (numel & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_l1_bwd: numel must be a multiple of 4");
    long long total4 = numel / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 8192) blocks = 8192;
    if (sign) {
        hipLaunchKernelGGL(l1_bwd_sgn_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char4*)sign, grad_out, coef, (const float4*)add,
                           (float4*)da, total4, amax_slots);
        return lp_check_launch("l1_bwd_sgn");
    }
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b, grad_out, coef,
                       (const float4*)add, (float4*)da, total4, relu_in, amax_slots);
    return lp_check_launch("l1_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// fused multi-tensor optimizers + EMA.  One launch updates every parameter tensor of an optimizer:
// `table` is a device array of MtDesc, blockIdx.y = tensor, blockIdx.x strides over its elements.  The step counter
// lives on the device (int64 *step, incremented by a 1-thread kernel) so a captured hipGraph replays correctly.
//   RAdam  -- utils/radam.py:29-95 (degenerated_to_sgd=True, weight_decay=0)
//   Adam   -- torch.optim.Adam (no amsgrad / weight decay), as used by runners/holycow.py:34-41 with betas=(beta1, 0.999), eps=1e-5
//   EMA    -- runners/holycow.py:99-109: avg = avg*alpha + cur*(1-alpha)
// ------------------------------------------------------------------------------------------------------------------
struct MtDesc { float* p; const float* g; float* m; float* v; long long n; };

__global__ void mt_step_inc_kernel(long long* step) { step[0] += 1; }

__global__ void mt_radam_kernel(const MtDesc* __restrict__ table, const long long* __restrict__ step_ptr, float lr, float beta1,
                                float beta2, float eps) {
    const MtDesc d = table[blockIdx.y];
    const double step = (double)step_ptr[0];
    // rectification term, evaluated in double like the Python reference
    const double beta2_t = pow((double)beta2, step);
    const double n_max = 2.0 / (1.0 - (double)beta2) - 1.0;
    const double n_sma = n_max - 2.0 * step * beta2_t / (1.0 - beta2_t);
    const double bc1 = 1.0 - pow((double)beta1, step);
    const bool rect = n_sma >= 5.0;
    const float step_size = rect ? (float)(sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / bc1 * lr)
                                 : (float)(1.0 / bc1 * lr);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += stride) {
        float g = d.g[i];
        float v = d.v[i] * beta2 + (1.f - beta2) * g * g;
        float m = d.m[i] * beta1 + (1.f - beta1) * g;
        d.v[i] = v; d.m[i] = m;
        float p = d.p[i];
        p -= rect ? step_size * m / (sqrtf(v) + eps) : step_size * m;
        d.p[i] = p;
    }
}

__global__ void mt_adam_kernel(const MtDesc* __restrict__ table, const long long* __restrict__ step_ptr, float lr, float beta1,
                               float beta2, float eps) {
    const MtDesc d = table[blockIdx.y];
    const double step = (double)step_ptr[0];
    const float bc1 = (float)(1.0 - pow((double)beta1, step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    const float step_size = lr / bc1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += stride) {
        float g = d.g[i];
        float m = d.m[i] * beta1 + (1.f - beta1) * g;
        float v = d.v[i] * beta2 + (1.f - beta2) * g * g;
        d.m[i] = m; d.v[i] = v;
        d.p[i] -= step_size * m / (sqrtf(v) / bc2_sqrt + eps);
    }
}

// table[k].p = running average, table[k].g = current value; m/v unused
__global__ void mt_ema_kernel(const MtDesc* __restrict__ table, float alpha, int copy_only) {
    const MtDesc d = table[blockIdx.y];
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float om = 1.f - alpha;
    float* avg = d.p;
    const float* cur = d.g;          // NOT __restrict__: may alias avg (see below)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += stride) {
        if (copy_only) { avg[i] = cur[i]; continue; }
        // Two in-place steps exactly like runners/holycow.py:104-106 (`p_avg *= alpha; p_avg += p * (1 - alpha)`).  This matters:
        // in fine-tuning the reference hands the SAME tensor to generator.enable_finetuning and to the EMA generator
        // (train.py:263-272), so both identity_embedding Parameters alias one storage and every iteration multiplies it by
        // alpha + (1 - alpha) * alpha.  Re-reading cur[i] after the store reproduces that.
        avg[i] = avg[i] * alpha;
        avg[i] = avg[i] + cur[i] * om;
    }
}

extern "C" int lp_mt_desc_bytes(void) { return (int)sizeof(MtDesc); }

extern "C" int lp_mt_optimizer_step(const void* table, int num_tensors, long long max_numel, long long* step, int kind, float lr,
                                    float beta1, float beta2, float eps, void* stream) {
    if (!table || !step || num_tensors <= 0) return lp_set_error(LP_ERR_ARG, "lp_mt_optimizer_step: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mt_step_inc_kernel, dim3(1), dim3(1), 0, st, step);
    int bx = (int)((max_numel + 2047) / 2048); if (bx < 1) bx = 1; if (bx > 1024) bx = 1024;     // large tensors need the whole chip
    dim3 grid(bx, num_tensors);
    if (kind == 0) hipLaunchKernelGGL(mt_radam_kernel, grid, dim3(256), 0, st, (const MtDesc*)table, step, lr, beta1, beta2, eps);
    else if (kind == 1) hipLaunchKernelGGL(mt_adam_kernel, grid, dim3(256), 0, st, (const MtDesc*)table, step, lr, beta1, beta2, eps);
    else return lp_set_error(LP_ERR_ARG, "lp_mt_optimizer_step: kind must be 0 (RAdam) or 1 (Adam)");
    return lp_check_launch("mt_optimizer_step");
}

extern "C" int lp_mt_ema(const void* table, int num_tensors, long long max_numel, float alpha, int copy_only, void* stream) {
    if (!table || num_tensors <= 0) return lp_set_error(LP_ERR_ARG, "lp_mt_ema: bad arguments");
    int bx = (int)((max_numel + 2047) / 2048); if (bx < 1) bx = 1; if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(mt_ema_kernel, dim3(bx, num_tensors), dim3(256), 0, (hipStream_t)stream, (const MtDesc*)table, alpha, copy_only);
    return lp_check_launch("mt_ema");
}

// ------------------------------------------------------------------------------------------------------------------
// Loss reductions (criterions/dice.py:20-39, criterions/adversarial.py:34-57)
// ------------------------------------------------------------------------------------------------------------------
// Dice: fake [B][Cf][HW], real [B][Cr][HW] (NCHW), Cf = 1 (broadcast over real's channels: the reference's 1-vs-3 channel quirk) or
// Cf = Cr.  overlap = sum 2*f*r over the BROADCAST shape, energy = sum f^2 (over fake's own elements) + sum r^2;
// loss = -log(overlap / energy) * weight.  Pass 1: DICE_BLOCKS block partials (3 sums each); pass 2: one block finishes, writes the
// loss and keeps {overlap, energy} for the backward:  d loss / d f = -weight * (2 * sum_c r / overlap - 2 f / energy) * grad_out.
#define DICE_BLOCKS 256
__global__ __launch_bounds__(256) void dice_partial_kernel(const float* __restrict__ fake, const float* __restrict__ real, float* __restrict__ part,
                                                           int B, int Cf, int Cr, int HW) {
    __shared__ float sh[3][4];
    float ov = 0.f, ef = 0.f, er = 0.f;
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i % HW);
        for (int c = 0; c < Cr; ++c) {
            const float r = real[((size_t)b * Cr + c) * HW + p];
            const float f = fake[((size_t)b * Cf + (Cf == 1 ? 0 : c)) * HW + p];
            ov = fmaf(2.f * f, r, ov); er = fmaf(r, r, er);
            if (Cf != 1 || c == 0) ef = fmaf(f, f, ef);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { ov += __shfl_down(ov, o, 64); ef += __shfl_down(ef, o, 64); er += __shfl_down(er, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = ov; sh[1][threadIdx.x >> 6] = ef; sh[2][threadIdx.x >> 6] = er; }
    __syncthreads();
    if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}
__global__ __launch_bounds__(64) void dice_finalize_kernel(const float* __restrict__ part, int nblocks, float weight, float* __restrict__ out,
                                                           float* __restrict__ sums) {
    double ov = 0, ef = 0, er = 0;
    for (int b = threadIdx.x; b < nblocks; b += 64) { ov += part[b * 3]; ef += part[b * 3 + 1]; er += part[b * 3 + 2]; }
    for (int o = 32; o > 0; o >>= 1) { ov += __shfl_down(ov, o, 64); ef += __shfl_down(ef, o, 64); er += __shfl_down(er, o, 64); }
    if (threadIdx.x == 0) {
        const float o_ = (float)ov, e_ = (float)(ef + er);
        sums[0] = o_; sums[1] = e_;
        out[0] = -logf(o_ / e_) * weight;
    }
}
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ fake, const float* __restrict__ real, const float* __restrict__ sums,
                                                       const float* __restrict__ gout, float weight, float* __restrict__ dfake, int B, int Cf,
                                                       int Cr, int HW) {
    const float io = 1.f / sums[0], ie = 1.f / sums[1], g = -weight * gout[0];
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i % HW);
        if (Cf == 1) {
            float rs = 0.f;
            for (int c = 0; c < Cr; ++c) rs += real[((size_t)b * Cr + c) * HW + p];
            const float f = fake[(size_t)b * HW + p];
            dfake[(size_t)b * HW + p] = g * (2.f * rs * io - 2.f * f * ie);
        } else {
            for (int c = 0; c < Cr; ++c) {
                const size_t o = ((size_t)b * Cr + c) * HW + p;
                dfake[o] = g * (2.f * real[o] * io - 2.f * fake[o] * ie);
            }
        }
    }
}

extern "C" int lp_dice_partial_blocks(void) { return DICE_BLOCKS; }

extern "C" int lp_reduce_dice(const float* fake, const float* real, float* partial, float* out, float* sums, int B, int Cf, int Cr, int HW,
                              float weight, void* stream) {
    if (!fake || !real || !partial || !out || !sums) return lp_set_error(LP_ERR_ARG, "lp_reduce_dice: null pointer");
    if (Cf != 1 && Cf != Cr) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_reduce_dice: fake must have 1 channel or as many as real");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_partial_kernel, dim3(DICE_BLOCKS), dim3(256), 0, st, fake, real, partial, B, Cf, Cr, HW);
    int rc = lp_check_launch("dice_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, st, partial, DICE_BLOCKS, weight, out, sums);
    return lp_check_launch("dice_finalize");
}

extern "C" int lp_reduce_dice_bwd(const float* fake, const float* real, const float* sums, const float* grad_out, float* dfake, int B, int Cf,
                                  int Cr, int HW, float weight, void* stream) {
    if (!fake || !real || !sums || !grad_out || !dfake) return lp_set_error(LP_ERR_ARG, "lp_reduce_dice_bwd: null pointer");
    if (Cf != 1 && Cf != Cr) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_reduce_dice_bwd: fake must have 1 channel or as many as real");
    const long long total = (long long)B * HW;
    long long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dice_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, fake, real, sums, grad_out, weight, dfake, B, Cf, Cr, HW);
    return lp_check_launch("dice_bwd");
}

// Hinge GAN losses on B critic scores (gan_type 'gan'): out[0] = loss_G = -mean(fake_G); out[1] = loss_D = mean(relu(1 - real)) +
// mean(relu(1 + fake_D)).  Backward: d_fake_G = -gG / B;  d_real = -gD / B * [1 - real > 0];  d_fake_D = gD / B * [1 + fake_D > 0].
__global__ __launch_bounds__(64) void hinge_fwd_kernel(const float* __restrict__ real, const float* __restrict__ fake_d, const float* __restrict__ fake_g,
                                                       float* __restrict__ out, int B) {
    float g = 0.f, d = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) { g -= fake_g[i]; d += fmaxf(1.f - real[i], 0.f) + fmaxf(1.f + fake_d[i], 0.f); }
    for (int o = 32; o > 0; o >>= 1) { g += __shfl_down(g, o, 64); d += __shfl_down(d, o, 64); }
    if (threadIdx.x == 0) { out[0] = g / (float)B; out[1] = d / (float)B; }
}
__global__ __launch_bounds__(64) void hinge_bwd_kernel(const float* __restrict__ real, const float* __restrict__ fake_d, const float* __restrict__ gG,
                                                       const float* __restrict__ gD, float* __restrict__ d_real, float* __restrict__ d_fake_d,
                                                       float* __restrict__ d_fake_g, int B) {
    const float ib = 1.f / (float)B;
    for (int i = threadIdx.x; i < B; i += 64) {
        if (d_fake_g) d_fake_g[i] = -gG[0] * ib;
        if (d_real) d_real[i] = (1.f - real[i] > 0.f) ? -gD[0] * ib : 0.f;
        if (d_fake_d) d_fake_d[i] = (1.f + fake_d[i] > 0.f) ? gD[0] * ib : 0.f;
    }
}
extern "C" int lp_reduce_hinge(const float* real, const float* fake_d, const float* fake_g, float* out, int B, void* stream) {
    if (!real || !fake_d || !fake_g || !out || B < 1) return lp_set_error(LP_ERR_ARG, "lp_reduce_hinge: null pointer");
    hipLaunchKernelGGL(hinge_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, real, fake_d, fake_g, out, B);
    return lp_check_launch("hinge_fwd");
}
extern "C" int lp_reduce_hinge_bwd(const float* real, const float* fake_d, const float* grad_G, const float* grad_D, float* d_real,
                                   float* d_fake_d, float* d_fake_g, int B, void* stream) {
    if (!real || !fake_d || B < 1) return lp_set_error(LP_ERR_ARG, "lp_reduce_hinge_bwd: null pointer");
    if ((d_fake_g && !grad_G) || ((d_real || d_fake_d) && !grad_D)) return lp_set_error(LP_ERR_ARG, "lp_reduce_hinge_bwd: missing upstream gradient");
    hipLaunchKernelGGL(hinge_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, real, fake_d, grad_G, grad_D, d_real, d_fake_d, d_fake_g, B);
    return lp_check_launch("hinge_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Projection head of the critic (discriminators/no_landmarks.py:100-108 of the reference): out = relu(out); pooled = out.sum(dim = (2, 3));
// score = linear(pooled) + (pooled * embed).sum(1).  Forward: pooled [N][C] and dot[n] = <pooled[n], embed[n]> in one launch (one workgroup per
// sample; the linear layer stays lp_linear_fwd on pooled).  Backward: d_out[n,p,c] = (g_pooled[n,c] + g_dot[n] embed[n,c]) [out > 0] and
// d_embed[n,c] = g_dot[n] pooled[n,c] in one launch.  (round 6: replaces relu + sum + mul + sum and their five autograd launches per pass)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proj_score_fwd_kernel(const float* __restrict__ out, const float* __restrict__ embed, float* __restrict__ pooled,
                                                             float* __restrict__ dot, int HW, int C) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += fmaxf(out[((size_t)n * HW + p) * C + c], 0.f);
        pooled[(size_t)n * C + c] = s;
        if (embed) acc = fmaf(s, embed[(size_t)n * C + c], acc);
    }
    if (!dot) return;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dot[n] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void proj_score_bwd_kernel(const float* __restrict__ out, const float* __restrict__ embed, const float* __restrict__ pooled,
                                                             const float* __restrict__ g_pooled, const float* __restrict__ g_dot, float* __restrict__ d_out,
                                                             float* __restrict__ d_embed, int HW, int C) {
    const int n = blockIdx.x;
    const float gd = g_dot ? g_dot[n] : 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const size_t nc = (size_t)n * C + c;
        float g = g_pooled ? g_pooled[nc] : 0.f;
        if (embed) g = fmaf(gd, embed[nc], g);
        if (d_embed) d_embed[nc] = gd * pooled[nc];
        if (d_out) for (int p = 0; p < HW; ++p) { const size_t i = ((size_t)n * HW + p) * C + c; d_out[i] = out[i] > 0.f ? g : 0.f; }
    }
}
extern "C" int lp_proj_score_fwd(const float* out, const float* embed, float* pooled, float* dot, int N, int HW, int C, void* stream) {
    if (!out || !pooled || (dot && !embed)) return lp_set_error(LP_ERR_ARG, "lp_proj_score_fwd: null pointer");
    if (N < 1) return LP_OK;
    hipLaunchKernelGGL(proj_score_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, out, embed, pooled, dot, HW, C);
    return lp_check_launch("proj_score_fwd");
}
extern "C" int lp_proj_score_bwd(const float* out, const float* embed, const float* pooled, const float* g_pooled, const float* g_dot, float* d_out,
                                 float* d_embed, int N, int HW, int C, void* stream) {
    if (!out || (!d_out && !d_embed) || (d_embed && (!pooled || !g_dot)) || (g_dot && !embed)) return lp_set_error(LP_ERR_ARG, "lp_proj_score_bwd: null pointer");
    if (N < 1) return LP_OK;
    hipLaunchKernelGGL(proj_score_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, out, embed, pooled, g_pooled, g_dot, d_out, d_embed, HW, C);
    return lp_check_launch("proj_score_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Input side of the VGG criterions (criterions/common/perceptual_loss.py:72-80,86-93 of the reference): x in [-1, 1], NCHW [N][3][HW] ->
// ((x + 1) / 2 - mean[c]) / std[c] as NHWC [N][HW][3] -- the same fp32 operations in the same order, and the layout change, in one launch
// (round 6: add + div + sub + div + the NCHW -> NHWC copy were five launches per image batch); backward dx[n,c,p] = g[n,p,c] / std[c] / 2.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void image_prep_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                             float* __restrict__ out, long long total, int HW) {
    const float m0 = mean[0], m1 = mean[1], m2 = mean[2], s0 = stdv[0], s1 = stdv[1], s2 = stdv[2];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / HW; const int p = (int)(i - n * HW);
        const float* xn = x + (size_t)n * 3 * HW + p;
        float* o = out + (size_t)i * 3;
        o[0] = ((xn[0] + 1.f) / 2.f - m0) / s0; o[1] = ((xn[HW] + 1.f) / 2.f - m1) / s1; o[2] = ((xn[2 * (size_t)HW] + 1.f) / 2.f - m2) / s2;
    }
}
__global__ __launch_bounds__(256) void image_prep_bwd_kernel(const float* __restrict__ g, const float* __restrict__ stdv, float* __restrict__ dx,
                                                             long long total, int HW) {
    const float s0 = stdv[0], s1 = stdv[1], s2 = stdv[2];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / HW; const int p = (int)(i - n * HW);
        const float* gi = g + (size_t)i * 3;
        float* d = dx + (size_t)n * 3 * HW + p;
        d[0] = gi[0] / s0 / 2.f; d[HW] = gi[1] / s1 / 2.f; d[2 * (size_t)HW] = gi[2] / s2 / 2.f;
    }
}
extern "C" int lp_image_prep_fwd(const float* x, const float* mean, const float* stdv, float* out, int N, int HW, void* stream) {
    if (!x || !mean || !stdv || !out) return lp_set_error(LP_ERR_ARG, "lp_image_prep_fwd: null pointer");
    const long long total = (long long)N * HW;
    if (total == 0) return LP_OK;
    long long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(image_prep_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mean, stdv, out, total, HW);
    return lp_check_launch("image_prep_fwd");
}
extern "C" int lp_image_prep_bwd(const float* g, const float* stdv, float* dx, int N, int HW, void* stream) {
    if (!g || !stdv || !dx) return lp_set_error(LP_ERR_ARG, "lp_image_prep_bwd: null pointer");
    const long long total = (long long)N * HW;
    if (total == 0) return LP_OK;
    long long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(image_prep_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, stdv, dx, total, HW);
    return lp_check_launch("image_prep_bwd");
}
