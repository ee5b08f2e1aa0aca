// HBM-bound helper kernels of the generator path (gfx950): weight packing, instance-norm statistics, AdaIN/ReLU
// backward, 2x2 block sums, generator head.  All NHWC fp32, float4-vectorised, coalesced along channels.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                    int Cout, int Cin, int T, int RowsP, int ColsP, int mode) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)T * RowsP * ColsP;
    if (idx >= total) return;
    int col = (int)(idx % ColsP);
    int row = (int)((idx / ColsP) % RowsP);
    int t = (int)(idx / ((long long)ColsP * RowsP));
    float v = 0.f;
    if (mode == 0) { if (row < Cout && col < Cin) v = w[((size_t)row * Cin + col) * T + t]; }                 // [t][co][ci]
    else { if (row < Cin && col < Cout) v = w[((size_t)col * Cin + row) * T + (T - 1 - t)]; }                  // [T-1-t][ci][co]
    __bf16 h = (__bf16)v;
    hi[idx] = __builtin_bit_cast(uint16_t, h);
    if (lo) { __bf16 l = (__bf16)(v - (float)h); lo[idx] = __builtin_bit_cast(uint16_t, l); }
}

extern "C" int lp_pack_weights(const float* w, uint16_t* hi, uint16_t* lo, int Cout, int Cin, int T, int RowsP, int ColsP, int mode,
                               void* stream) {
    if (!w || !hi) return lp_set_error(LP_ERR_ARG, "lp_pack_weights: null pointer");
    int rows = mode == 0 ? Cout : Cin, cols = mode == 0 ? Cin : Cout;
    if (RowsP < rows || ColsP < cols || (ColsP & 7)) return lp_set_error(LP_ERR_ARG, "lp_pack_weights: bad padded dims");
    long long total = (long long)T * RowsP * ColsP;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, hi, lo, Cout, Cin, T,
                       RowsP, ColsP, mode);
    return lp_check_launch("pack_weights");
}

// ------------------------------------------------------------------------------------------------------------------
// instance-norm statistics -> AdaIN scale/shift
//   stage 1: grid (splits, C/64 blocks, N); block = 16 channel-quads x 16 pixel lanes; per-thread shifted sums,
//            merged with Chan's parallel-variance formula (no E[x^2]-E[x]^2 cancellation).
//   stage 2: one thread per (n,c) merges the splits in fp64 and emits mean, rstd, scale, shift.
// ------------------------------------------------------------------------------------------------------------------
#define STAT_SPLIT_PIX 1024   // pixels per stage-1 block (256^2 x 64ch x 8 images -> 4096 blocks)

__device__ __forceinline__ void chan_merge(float& n_a, float& mean_a, float& m2_a, float n_b, float mean_b, float m2_b) {
    if (n_b == 0.f) return;
    if (n_a == 0.f) { n_a = n_b; mean_a = mean_b; m2_a = m2_b; return; }
    float n = n_a + n_b, d = mean_b - mean_a;
    mean_a += d * (n_b / n);
    m2_a += m2_b + d * d * (n_a * n_b / n);
    n_a = n;
}

__global__ __launch_bounds__(256) void instnorm_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                               int S) {
    __shared__ float sh[3][16][64];
    const int n = blockIdx.z, cb = blockIdx.y, s = blockIdx.x;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = cb * 64 + cq * 4;
    const int p0 = s * STAT_SPLIT_PIX, p1 = min(HW, p0 + STAT_SPLIT_PIX);
    float cnt = 0.f, ref[4] = {0, 0, 0, 0}, sd[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    if (c < C) {
        const float* base = x + (size_t)n * HW * C + c;
        const bool vec = (C & 3) == 0;
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

__global__ void instnorm_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         int ab_stride, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                         float* __restrict__ scale, float* __restrict__ shift, int N, int C, int S) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    int n = idx / C, c = idx % C;
    double na = 0, ma = 0, qa = 0;
    for (int s = 0; s < S; ++s) {
        const float* q = part + (((size_t)n * S + s) * C + c) * 3;
        double nb = q[0], mb = q[1], qb = q[2];
        if (nb == 0) continue;
        if (na == 0) { na = nb; ma = mb; qa = qb; continue; }
        double nn = na + nb, d = mb - ma;
        ma += d * (nb / nn); qa += qb + d * d * (na * nb / nn); na = nn;
    }
    float var = (float)(qa / na);
    float m = (float)ma, r = 1.0f / sqrtf(var + eps);
    mean[idx] = m; rstd[idx] = r;
    if (scale) {
        float g = gamma ? gamma[(size_t)n * ab_stride + c] : 1.f, b = beta ? beta[(size_t)n * ab_stride + c] : 0.f;
        float sc = r * g;
        scale[idx] = sc; shift[idx] = b - m * sc;
    }
}

extern "C" long long lp_instnorm_workspace_bytes(int N, int HW, int C) {
    long long S = (HW + STAT_SPLIT_PIX - 1) / STAT_SPLIT_PIX;
    return (long long)N * S * C * 3 * 4;
}

extern "C" int lp_instnorm_stats(const float* x, const float* gamma, const float* beta, int ab_stride, float eps, float* mean,
                                 float* rstd, float* scale, float* shift, float* workspace, int N, int HW, int C, void* stream) {
    if (!x || !mean || !rstd || !workspace) return lp_set_error(LP_ERR_ARG, "lp_instnorm_stats: null pointer");
    int S = (HW + STAT_SPLIT_PIX - 1) / STAT_SPLIT_PIX;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_partial_kernel, dim3(S, (C + 63) / 64, N), dim3(256), 0, st, x, workspace, HW, C, S);
    int rc = lp_check_launch("instnorm_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(cdiv((long long)N * C, 256)), dim3(256), 0, st, workspace, gamma, beta, ab_stride,
                       eps, mean, rstd, scale, shift, N, C, S);
    return lp_check_launch("instnorm_finalize");
}

// ------------------------------------------------------------------------------------------------------------------
// backward of relu(AdaIN(x)) [+ x2 nearest upsample]
//   pass 1: g = sum2x2?(dA) * mask  -> written into dx (as temporary), partial sums S1 = sum g, S2 = sum g*xhat
//   pass 2: coefficients; pass 3: dx = ca*g + cb*x + cc (+ add)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adain_bwd_partial_kernel(const float* __restrict__ dA, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                float* __restrict__ g_out, float* __restrict__ part, int H, int W, int C,
                                                                int ups, int S) {
    __shared__ float sh[2][16][64];
    const int n = blockIdx.z, cb = blockIdx.y, s = blockIdx.x;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = cb * 64 + cq * 4;
    const int HW = H * W;
    const int p0 = s * STAT_SPLIT_PIX, p1 = min(HW, p0 + STAT_SPLIT_PIX);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (c < C) {                                       // C % 4 == 0 required (checked on host)
        float4 mu = *(const float4*)(mean + (size_t)n * C + c), rs = *(const float4*)(rstd + (size_t)n * C + c);
        float4 sc = *(const float4*)(scale + (size_t)n * C + c), sf = *(const float4*)(shift + (size_t)n * C + c);
        for (int pix = p0 + pl; pix < p1; pix += 16) {
            float4 xv = *(const float4*)(x + ((size_t)n * HW + pix) * C + c);
            float4 g;
            if (ups) {
                int yy = pix / W, xx = pix % W;
                const float* b = dA + (((size_t)n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
                float4 a0 = *(const float4*)b, a1 = *(const float4*)(b + C), a2 = *(const float4*)(b + (size_t)2 * W * C),
                       a3 = *(const float4*)(b + (size_t)2 * W * C + C);
                g.x = (a0.x + a1.x) + (a2.x + a3.x); g.y = (a0.y + a1.y) + (a2.y + a3.y);
                g.z = (a0.z + a1.z) + (a2.z + a3.z); g.w = (a0.w + a1.w) + (a2.w + a3.w);
            } else g = *(const float4*)(dA + ((size_t)n * HW + pix) * C + c);
            g.x = fmaf(xv.x, sc.x, sf.x) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sc.y, sf.y) > 0.f ? g.y : 0.f;
            g.z = fmaf(xv.z, sc.z, sf.z) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sc.w, sf.w) > 0.f ? g.w : 0.f;
            *(float4*)(g_out + ((size_t)n * HW + pix) * C + c) = g;
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

__global__ void adain_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, int ab_stride,
                                          const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, float* __restrict__ coef, int N, int C, int S, float inv_hw) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    int n = idx / C, c = idx % C;
    double a1 = 0, a2 = 0;
    for (int s = 0; s < S; ++s) { const float* q = part + (((size_t)n * S + s) * C + c) * 2; a1 += q[0]; a2 += q[1]; }
    float S1 = (float)a1, S2 = (float)a2;
    if (dgamma) dgamma[(size_t)n * ab_stride + c] = S2;
    if (dbeta) dbeta[(size_t)n * ab_stride + c] = S1;
    float g = gamma ? gamma[(size_t)n * ab_stride + c] : 1.f;
    float r = rstd[idx], m = mean[idx];
    float ca = g * r;
    float cb = -ca * r * S2 * inv_hw;
    float cc = -ca * S1 * inv_hw - cb * m;
    coef[(size_t)idx * 3 + 0] = ca; coef[(size_t)idx * 3 + 1] = cb; coef[(size_t)idx * 3 + 2] = cc;
}

__global__ void adain_bwd_apply_kernel(float* __restrict__ dx /* holds g */, const float* __restrict__ x, const float* __restrict__ add,
                                       const float* __restrict__ coef, long long total4, int HW, int C) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C4 = C >> 2;
    for (; i < total4; i += stride) {
        int c = (int)(i % C4) * 4;
        int n = (int)(i / ((long long)C4 * HW));
        const float* cf = coef + ((size_t)n * C + c) * 3;
        float4 g = ((const float4*)dx)[i], xv = ((const float4*)x)[i], o;
        o.x = fmaf(cf[0], g.x, fmaf(cf[1], xv.x, cf[2]));
        o.y = fmaf(cf[3], g.y, fmaf(cf[4], xv.y, cf[5]));
        o.z = fmaf(cf[6], g.z, fmaf(cf[7], xv.z, cf[8]));
        o.w = fmaf(cf[9], g.w, fmaf(cf[10], xv.w, cf[11]));
        if (add) { float4 a = ((const float4*)add)[i]; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
        ((float4*)dx)[i] = o;
    }
}

extern "C" long long lp_adain_bwd_workspace_bytes(int N, int HW, int C) {
    long long S = (HW + STAT_SPLIT_PIX - 1) / STAT_SPLIT_PIX;
    return (long long)N * S * C * 2 * 4 + (long long)N * C * 3 * 4;
}

extern "C" int lp_adain_relu_bwd(const float* dA, const float* x, const float* add, const float* gamma, int ab_stride, const float* mean,
                                 const float* rstd, const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta,
                                 float* workspace, int N, int H, int W, int C, int upsample, void* stream) {
    if (!dA || !x || !mean || !rstd || !scale || !shift || !dx || !workspace) return lp_set_error(LP_ERR_ARG, "lp_adain_relu_bwd: null pointer");
    if (C & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_adain_relu_bwd: C must be a multiple of 4");
    const int HW = H * W;
    int S = (HW + STAT_SPLIT_PIX - 1) / STAT_SPLIT_PIX;
    float* part = workspace;
    float* coef = workspace + (size_t)N * S * C * 2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adain_bwd_partial_kernel, dim3(S, (C + 63) / 64, N), dim3(256), 0, st, dA, x, mean, rstd, scale, shift, dx, part,
                       H, W, C, upsample, S);
    int rc = lp_check_launch("adain_bwd_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(adain_bwd_finalize_kernel, dim3(cdiv((long long)N * C, 256)), dim3(256), 0, st, part, gamma, ab_stride, mean, rstd,
                       dgamma, dbeta, coef, N, C, S, 1.0f / (float)HW);
    rc = lp_check_launch("adain_bwd_finalize");
    if (rc) return rc;
    long long total4 = (long long)N * HW * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adain_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, dx, x, add, coef, total4, HW, C);
    return lp_check_launch("adain_bwd_apply");
}

// ------------------------------------------------------------------------------------------------------------------
// 2x2 block sum (adjoint of nearest x2 upsampling)
// ------------------------------------------------------------------------------------------------------------------
__global__ void sum2x2_kernel(const float* __restrict__ in, float* __restrict__ out, long long total4, int H, int W, int C) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C4 = C >> 2;
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
        ((float4*)out)[i] = o;
    }
}

extern "C" int lp_sum2x2(const float* in, float* out, int N, int H, int W, int C, void* stream) {
    if (!in || !out) return lp_set_error(LP_ERR_ARG, "lp_sum2x2: null pointer");
    if (C & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_sum2x2: C must be a multiple of 4");
    long long total4 = (long long)N * H * W * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum2x2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, total4, H, W, C);
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
                                float* __restrict__ dz, int N, int HW) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * HW) return;
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
}

extern "C" int lp_head_fwd(const float* z, float* t, float* fake_rgbs, float* fake_segm, int N, int H, int W, void* stream) {
    if (!z || !fake_rgbs || !fake_segm) return lp_set_error(LP_ERR_ARG, "lp_head_fwd: null pointer");
    long long total = (long long)N * H * W;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, z, t, fake_rgbs, fake_segm, N, H * W);
    return lp_check_launch("head_fwd");
}

extern "C" int lp_head_bwd(const float* t, const float* d_rgbs, const float* d_segm, float* dz, int N, int H, int W, void* stream) {
    if (!t || !d_rgbs || !dz) return lp_set_error(LP_ERR_ARG, "lp_head_bwd: null pointer");
    long long total = (long long)N * H * W;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, t, d_rgbs, d_segm, dz, N, H * W);
    return lp_check_launch("head_bwd");
}
