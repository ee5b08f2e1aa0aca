// Forward pass of the MobileNetV2 pose encoder (reference: embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26-28,56-58,
// torchvision's mobilenet_v2) on gfx950 -- the layers that are NOT dense contractions.  The 1x1 (pointwise) convs run on
// lp_conv16_fwd; everything here is bandwidth-bound fp32, NHWC, 16 bytes per lane:
//   stem_conv_s2   3x3 stride-2 conv of the NCHW RGB frame, 3 -> 32 channels (27 MACs per output: no matrix work)
//   dwconv3x3      depthwise 3x3, stride 1 | 2, with the producer's BatchNorm + ReLU6 applied while its input is loaded
//                  (clamp(x*scale[c]+shift[c], 0, 6); the conv's zero padding is applied after that activation)
//   affine_res     x = y*scale[c] + shift[c] (+ residual): the linear BatchNorm that ends an inverted-residual block, also emitting
//                  the 16-bit operand planes of x for the next block's 1x1 expand conv
//   affine_relu6_mean   global average pool of relu6(BatchNorm(y)) -> [N][C] (input of the classifier)
//   bn_stats       train-mode BatchNorm: batch statistics over the N*H*W positions of a channel -> (scale, shift), running_mean /
//                  running_var momentum update, in two launches
// BatchNorm appears as a per-channel (scale, shift): from the running statistics in eval mode, from lp_bn_stats in train mode.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"
// No packed fp32 arithmetic in this file: on gfx950 the forms the compiler picks for "vector times broadcast scalar" with the scalar in the
// HIGH dword of a register pair (v_pk_fma_f32 ... op_sel:[0,1,0], v_pk_mul_f32 / v_pk_add_f32 op_sel:[0,1]) returned a wrong LOW half in lanes
// 48..63 whenever the LDS-DMA convolution kernels ran beside them on another stream (scripts/pk_forms_probe.py,
// profiles/r05_pk_fp32_opsel_hazard.txt; alone they are exact).  tests/test_isa_lint.py keeps those forms out of the whole library.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#endif

// x [N][3][H][W] fp32 (NCHW, as the dataloader hands frames over), w [Cout][3][3][3] (nn.Conv2d layout) -> y [N][H/2][W/2][Cout]
__global__ __launch_bounds__(256) void stem_conv_s2_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                           int N, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float ws[];          // [27][Cout]
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) { const int co = i % Cout, k = i / Cout; ws[i] = w[co * 27 + k]; }
    __syncthreads();
    const int Ho = H >> 1, Wo = W >> 1, C4 = Cout >> 2;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long pix = i / C4;
        const int xo = (int)(pix % Wo); pix /= Wo;
        const int yo = (int)(pix % Ho); const int n = (int)(pix / Ho);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = 2 * yo + ky - 1, ix = 2 * xo + kx - 1;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                        const float v = x[(((size_t)n * 3 + ci) * H + iy) * W + ix];
                        const float4 wv = *(const float4*)(ws + ((ci * 3 + ky) * 3 + kx) * Cout + c4 * 4);
                        acc.x = fmaf(v, wv.x, acc.x); acc.y = fmaf(v, wv.y, acc.y); acc.z = fmaf(v, wv.z, acc.z); acc.w = fmaf(v, wv.w, acc.w);
                    }
                }
        *(float4*)(y + (((size_t)n * Ho + yo) * Wo + xo) * Cout + c4 * 4) = acc;
    }
}

extern "C" int lp_stem_conv_s2(const float* x, const float* w, float* y, int N, int H, int W, int Cout, void* stream) {
    if (!x || !w || !y) return lp_set_error(LP_ERR_ARG, "lp_stem_conv_s2: null pointer");
    if ((Cout & 3) || (H & 1) || (W & 1) || Cout > 256) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_stem_conv_s2: needs even H, W and Cout % 4 == 0, <= 256");
    const long long total = (long long)N * (H / 2) * (W / 2) * (Cout / 4);
    long long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(stem_conv_s2_kernel, dim3((unsigned)blocks), dim3(256), (size_t)27 * Cout * sizeof(float), (hipStream_t)stream, x, w, y, N, H, W, Cout);
    return lp_check_launch("stem_conv_s2");
}

// depthwise 3x3, pad 1: y[n,yo,xo,c] = sum_{ky,kx} act(x)[n, yo*s+ky-1, xo*s+kx-1, c] * w[c][ky][kx];  act = clamp(x*sc[c]+sh[c], 0, 6) | identity
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, float* __restrict__ y, int N, int H, int W, int C,
                                                        int stride) {
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, C4 = C >> 2;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        long long pix = i / C4;
        const int xo = (int)(pix % Wo); pix /= Wo;
        const int yo = (int)(pix % Ho); const int n = (int)(pix / Ho);
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sc) { s = *(const float4*)(sc + c); t = *(const float4*)(sh + c); }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = yo * stride + ky - 1, ix = xo * stride + kx - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    float4 v = *(const float4*)(x + (((size_t)n * H + iy) * W + ix) * C + c);
                    if (sc) {
                        v.x = fminf(fmaxf(fmaf(v.x, s.x, t.x), 0.f), 6.f); v.y = fminf(fmaxf(fmaf(v.y, s.y, t.y), 0.f), 6.f);
                        v.z = fminf(fmaxf(fmaf(v.z, s.z, t.z), 0.f), 6.f); v.w = fminf(fmaxf(fmaf(v.w, s.w, t.w), 0.f), 6.f);
                    }
                    // w: [C][3][3] (nn.Conv2d depthwise layout [C][1][3][3])
                    const int k = ky * 3 + kx;
                    acc.x = fmaf(v.x, w[(c + 0) * 9 + k], acc.x); acc.y = fmaf(v.y, w[(c + 1) * 9 + k], acc.y);
                    acc.z = fmaf(v.z, w[(c + 2) * 9 + k], acc.z); acc.w = fmaf(v.w, w[(c + 3) * 9 + k], acc.w);
                }
            }
        *(float4*)(y + (((size_t)n * Ho + yo) * Wo + xo) * C + c) = acc;
    }
}

extern "C" int lp_dwconv3x3_fwd(const float* x, const float* w, const float* in_scale, const float* in_shift, float* y,
                                int N, int H, int W, int C, int stride, void* stream) {
    if (!x || !w || !y) return lp_set_error(LP_ERR_ARG, "lp_dwconv3x3_fwd: null pointer");
    if ((C & 3) || (stride != 1 && stride != 2) || (!in_scale != !in_shift)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_dwconv3x3_fwd: needs C % 4 == 0, stride 1|2");
    const long long total = (long long)N * ((H + stride - 1) / stride) * ((W + stride - 1) / stride) * (C / 4);
    long long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, in_scale, in_shift, y, N, H, W, C, stride);
    return lp_check_launch("dwconv3x3_fwd");
}

// x = y*scale[c] + shift[c] (+ res); optionally also the 16-bit operand planes [P][C8] of x (C % 8 == 0 required for them)
template <int PREC>
__global__ __launch_bounds__(256) void affine_res_kernel(const float* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                                         const float* __restrict__ res, float* __restrict__ x, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo, long long items, int C) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    const int G = C >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % G) * 8;
        const float* src = y + i * 8;
        float v[8];
        const float4 p0 = *(const float4*)src, p1 = *(const float4*)(src + 4);
        v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[c + j], sh[c + j]);
        if (res) {
            const float4 r0 = *(const float4*)(res + i * 8), r1 = *(const float4*)(res + i * 8 + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
        *(float4*)(x + i * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(x + i * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        if (hi) {
            s16x8_t h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint16_t hb = lp_f32_to_op16<F16>(v[j]);
                h[j] = (short)hb;
                if (SPLIT) l[j] = (short)lp_f32_to_op16<false>(v[j] - lp_op16_to_f32<false>(hb));
            }
            *(s16x8_t*)(hi + i * 8) = h;
            if (SPLIT) *(s16x8_t*)(lo + i * 8) = l;
        }
    }
}

extern "C" int lp_affine_res(const float* y, const float* scale, const float* shift, const float* res, float* x, uint16_t* hi, uint16_t* lo,
                             long long P, int C, int prec, void* stream) {
    if (!y || !scale || !shift || !x) return lp_set_error(LP_ERR_ARG, "lp_affine_res: null pointer");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_affine_res: C must be a multiple of 8");
    if (hi && prec == LP_PREC_BF16X3 && !lo) return lp_set_error(LP_ERR_ARG, "lp_affine_res: bf16x3 planes need lo");
    const long long items = P * (C >> 3);
    if (items == 0) return LP_OK;
    long long blocks = (items + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipStream_t st = (hipStream_t)stream;
#define LP_AR(Q) hipLaunchKernelGGL(affine_res_kernel<Q>, dim3((unsigned)blocks), dim3(256), 0, st, y, scale, shift, res, x, hi, lo, items, C)
    if (prec == LP_PREC_BF16) LP_AR(LP_PREC_BF16);
    else if (prec == LP_PREC_BF16X3) LP_AR(LP_PREC_BF16X3);
    else if (prec == LP_PREC_F16) LP_AR(LP_PREC_F16);
    else return lp_set_error(LP_ERR_ARG, "lp_affine_res: unknown precision mode");
#undef LP_AR
    return lp_check_launch("affine_res");
}

// out[n][c] = mean over HW of clamp(y[n][p][c]*scale[c] + shift[c], 0, 6);  one block per (n, 64-channel group)
__global__ __launch_bounds__(256) void affine_relu6_mean_kernel(const float* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                                                float* __restrict__ out, int HW, int C) {
    __shared__ float red[4][64];
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
    float a = 0.f;
    if (c < C) {
        const float s = sc[c], t = sh[c];
        for (int p = pl; p < HW; p += 4) a += fminf(fmaxf(fmaf(y[((size_t)n * HW + p) * C + c], s, t), 0.f), 6.f);
    }
    red[pl][threadIdx.x & 63] = a;
    __syncthreads();
    if (pl == 0 && c < C) out[(size_t)n * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)HW;
}

extern "C" int lp_affine_relu6_mean(const float* y, const float* scale, const float* shift, float* out, int N, int HW, int C, void* stream) {
    if (!y || !scale || !shift || !out) return lp_set_error(LP_ERR_ARG, "lp_affine_relu6_mean: null pointer");
    hipLaunchKernelGGL(affine_relu6_mean_kernel, dim3((C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, y, scale, shift, out, HW, C);
    return lp_check_launch("affine_relu6_mean");
}

// ---- train-mode BatchNorm: batch statistics of y [P][C] -> (scale, shift), running-statistics update ------------------------------
// Pass 1: the tensor is a stream of float4 channel groups; the grid's thread count is trimmed to a multiple of the C/4 groups per
// position, so a thread meets the same 4 channels on every step and keeps (count, mean, M2) for them in registers (shifted by the
// first value it sees: no cancellation at large |mean|/std).  Pass 2: one block per channel group merges the per-thread partials
// (Chan et al. pairwise update), writes scale = gamma*rstd, shift = beta - mean*scale and the momentum update of running_mean /
// running_var (unbiased variance, as nn.BatchNorm2d does).

__device__ __forceinline__ void bn_merge(float& n_a, float& mean_a, float& m2_a, float n_b, float mean_b, float m2_b) {
    if (n_b == 0.f) return;
    if (n_a == 0.f) { n_a = n_b; mean_a = mean_b; m2_a = m2_b; return; }
    const float n = n_a + n_b, d = mean_b - mean_a;
    mean_a += d * (n_b / n);
    m2_a += m2_b + d * d * (n_a * n_b / n);
    n_a = n;
}

// launch shape: G4 <= 256: every block uses BT = 256 - 256 % G4 threads, so thread tid owns group tid % G4 and the block folds its
// BT / G4 threads per group in LDS -> one partial per (block, group).  G4 > 256 (C = 1280: a few hundred positions): one partial per
// thread of a grid trimmed to a multiple of G4.  Either way the partials of group g sit at the indices = g (mod G4) below `Tu`.
struct BnShape { int blocks, BT, Tu, T; };
static BnShape bn_shape(long long P, int C) {
    const long long G4 = C >> 2, items = P * G4;
    BnShape s;
    if (G4 <= 256) {
        s.BT = 256 - 256 % (int)G4;
        long long nb = (items + 8ll * s.BT - 1) / (8ll * s.BT);       // >= 8 float4 per thread where the tensor is large enough
        s.blocks = (int)(nb > 512 ? 512 : (nb < 1 ? 1 : nb));
        s.Tu = s.blocks * (int)G4; s.T = s.Tu;
    } else {
        long long T = (items + 7) / 8;
        if (T > 32768) T = 32768;
        if (T < G4) T = G4;
        T = (T + 255) / 256 * 256;
        s.blocks = (int)(T / 256); s.BT = 256; s.T = (int)T; s.Tu = (int)(T - T % G4);
    }
    return s;
}

__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long long items, int G4, int BT,
                                                         int Tu, int T) {
    __shared__ float sh[9][256];
    const bool fold = G4 <= 256;
    const long long stride = fold ? (long long)gridDim.x * BT : Tu;
    const long long t0 = fold ? (long long)blockIdx.x * BT + threadIdx.x : (long long)blockIdx.x * 256 + threadIdx.x;
    const bool active = fold ? (int)threadIdx.x < BT : t0 < Tu;
    float cnt = 0.f, ref[4] = {0, 0, 0, 0}, sd[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    if (active) {
        long long i = t0;
        for (; i + 3 * stride < items; i += 4 * stride) {                  // 4 loads in flight
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *(const float4*)(x + (i + u * stride) * 4);
            if (cnt == 0.f) { ref[0] = q[0].x; ref[1] = q[0].y; ref[2] = q[0].z; ref[3] = q[0].w; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float d;
                d = q[u].x - ref[0]; sd[0] += d; sq[0] = fmaf(d, d, sq[0]);
                d = q[u].y - ref[1]; sd[1] += d; sq[1] = fmaf(d, d, sq[1]);
                d = q[u].z - ref[2]; sd[2] += d; sq[2] = fmaf(d, d, sq[2]);
                d = q[u].w - ref[3]; sd[3] += d; sq[3] = fmaf(d, d, sq[3]);
            }
            cnt += 4.f;
        }
        for (; i < items; i += stride) {
            const float4 q = *(const float4*)(x + i * 4);
            if (cnt == 0.f) { ref[0] = q.x; ref[1] = q.y; ref[2] = q.z; ref[3] = q.w; }
            float d;
            d = q.x - ref[0]; sd[0] += d; sq[0] = fmaf(d, d, sq[0]);
            d = q.y - ref[1]; sd[1] += d; sq[1] = fmaf(d, d, sq[1]);
            d = q.z - ref[2]; sd[2] += d; sq[2] = fmaf(d, d, sq[2]);
            d = q.w - ref[3]; sd[3] += d; sq[3] = fmaf(d, d, sq[3]);
            cnt += 1.f;
        }
    }
    float mean[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mean[j] = 0.f; m2[j] = 0.f;
        if (cnt > 0.f) { mean[j] = ref[j] + sd[j] / cnt; m2[j] = fmaxf(sq[j] - sd[j] * sd[j] / cnt, 0.f); }
    }
    if (!fold) {                                                            // SoA [9][T], one partial per thread
        part[t0] = cnt;
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[(size_t)(1 + j) * T + t0] = mean[j]; part[(size_t)(5 + j) * T + t0] = m2[j]; }
        return;
    }
    sh[0][threadIdx.x] = cnt;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[1 + j][threadIdx.x] = mean[j]; sh[5 + j][threadIdx.x] = m2[j]; }
    __syncthreads();
    if ((int)threadIdx.x < G4) {
        for (int k = threadIdx.x + G4; k < BT; k += G4) {
            const float nb = sh[0][k];
            float na = cnt;
#pragma unroll
            for (int j = 0; j < 4; ++j) { na = cnt; bn_merge(na, mean[j], m2[j], nb, sh[1 + j][k], sh[5 + j][k]); }
            cnt = na;
        }
        const size_t o = (size_t)blockIdx.x * G4 + threadIdx.x;
        part[o] = cnt;
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[(size_t)(1 + j) * T + o] = mean[j]; part[(size_t)(5 + j) * T + o] = m2[j]; }
    }
}

// merge of two partials of one channel GROUP: the four channels share their count, so the two quotients are taken once
struct BnAcc { float n, mean[4], m2[4]; };
__device__ __forceinline__ void bn_merge4(BnAcc& a, float nb, const float* mb, const float* qb) {
    if (nb == 0.f) return;
    if (a.n == 0.f) { a.n = nb; for (int j = 0; j < 4; ++j) { a.mean[j] = mb[j]; a.m2[j] = qb[j]; } return; }
    const float n = a.n + nb, inv = 1.f / n, fb = nb * inv, fab = a.n * fb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = mb[j] - a.mean[j];
        a.mean[j] = fmaf(d, fb, a.mean[j]);
        a.m2[j] += qb[j] + d * d * fab;
    }
    a.n = n;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ rm, float* __restrict__ rv, float* __restrict__ scale,
                                                          float* __restrict__ shift, int G4, int Tu, int T, float eps, float momentum) {
    __shared__ float sh[4][9];
    const int g = blockIdx.x;
    BnAcc a; a.n = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a.mean[j] = 0.f; a.m2[j] = 0.f; }
    for (int t = g + threadIdx.x * G4; t < Tu; t += 256 * G4) {
        float mb[4], qb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { mb[j] = part[(size_t)(1 + j) * T + t]; qb[j] = part[(size_t)(5 + j) * T + t]; }
        bn_merge4(a, part[t], mb, qb);
    }
    for (int o = 32; o > 0; o >>= 1) {                                            // wave reduction by shuffles
        const float nb = __shfl_down(a.n, o, 64);
        float mb[4], qb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { mb[j] = __shfl_down(a.mean[j], o, 64); qb[j] = __shfl_down(a.m2[j], o, 64); }
        bn_merge4(a, nb, mb, qb);
    }
    if ((threadIdx.x & 63) == 0) {
        float* d = sh[threadIdx.x >> 6];
        d[0] = a.n;
#pragma unroll
        for (int j = 0; j < 4; ++j) { d[1 + j] = a.mean[j]; d[5 + j] = a.m2[j]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) bn_merge4(a, sh[w][0], &sh[w][1], &sh[w][5]);
        const float unbias = a.n > 1.f ? a.n / (a.n - 1.f) : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = g * 4 + j;
            const float var = a.m2[j] / a.n;
            const float sc = gamma[c] / sqrtf(var + eps);
            scale[c] = sc; shift[c] = beta[c] - a.mean[j] * sc;
            if (rm) {
                rm[c] = (1.f - momentum) * rm[c] + momentum * a.mean[j];
                rv[c] = (1.f - momentum) * rv[c] + momentum * var * unbias;
            }
        }
    }
}

extern "C" long long lp_bn_stats_workspace_bytes(long long P, int C) { return 9ll * bn_shape(P, C).T * (long long)sizeof(float); }

extern "C" int lp_bn_stats(const float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, float* scale,
                           float* shift, float* workspace, long long P, int C, float eps, float momentum, void* stream) {
    if (!y || !gamma || !beta || !scale || !shift || !workspace) return lp_set_error(LP_ERR_ARG, "lp_bn_stats: null pointer");
    if ((C & 3) || P < 1 || (!running_mean != !running_var)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_stats: needs C % 4 == 0, P >= 1");
    const int G4 = C >> 2;
    const BnShape sp = bn_shape(P, C);
    const int T = sp.T, Tu = sp.Tu;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(sp.blocks), dim3(256), 0, st, y, workspace, P * G4, G4, sp.BT, Tu, T);
    int rc = lp_check_launch("bn_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(G4), dim3(256), 0, st, workspace, gamma, beta, running_mean, running_var, scale, shift, G4, Tu, T,
                       eps, momentum);
    return lp_check_launch("bn_finalize");
}

// ---- pointwise (1x1) conv of the encoder on the VALU, fp32 ---------------------------------------------------------------------
// y[p][n] = sum_k a[p][k] * w[n][k],   a = in_relu6 ? relu6(x*in_scale[k]+in_shift[k]) : x*in_scale[k]+in_shift[k] (+ in_res[p][k])
// The encoder is ~0.3 GFLOP per frame spread over 35 such layers with 64 .. 131072 positions: launch-, latency- and bandwidth-bound,
// not MFMA work.  A 64 x 64 output tile per workgroup (4 x 4 per thread), k in chunks of 16 through LDS; the producer's BatchNorm
// (+ ReLU6, + the block's residual) is applied while the A tile is loaded, and the tile column 0 workgroups can write that
// activated input back (x_out: the block input a later residual add needs).  stats: per-(tile row, channel) count / mean / M2 of
// the raw outputs, in the partial layout of bn_finalize_kernel -- train-mode BatchNorm then costs one small finalize launch.
#define PW_BM 64
#define PW_BN 64
#define PW_BK 16
struct PwParams {
    const float* x; const float* w; float* y;
    const float* in_scale; const float* in_shift; const float* in_res; float* x_out;
    float* part;
    int P, K, N, in_relu6;
};

__global__ __launch_bounds__(256) void pwconv_kernel(PwParams p) {
    __shared__ __attribute__((aligned(16))) float sm[PW_BM * (PW_BN + 4)];        // As | Bs during the k loop, the output tile afterwards
    float* As = sm;                                                               // [PW_BK][PW_BM + 4]
    float* Bs = sm + PW_BK * (PW_BM + 4);                                         // [PW_BK][PW_BN + 4]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int p0 = blockIdx.x * PW_BM, n0 = blockIdx.y * PW_BN;
    const int lr = tid >> 2, lk = (tid & 3) * 4;                                   // this thread's (row, k quad) of both tile loads
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const bool arow = (p0 + lr) < p.P, brow = (n0 + lr) < p.N;
    // the global loads of chunk c+1 are issued before the FMAs of chunk c (registers), so their latency hides behind the arithmetic
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    auto gload = [&](int k0) {
        const int k = k0 + lk;
        a = make_float4(0.f, 0.f, 0.f, 0.f); b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (arow && k < p.K) a = *(const float4*)(p.x + (size_t)(p0 + lr) * p.K + k);
        if (brow && k < p.K) b = *(const float4*)(p.w + (size_t)(n0 + lr) * p.K + k);
    };
    gload(0);
    for (int k0 = 0; k0 < p.K; k0 += PW_BK) {
        const int k = k0 + lk;
        if (arow && k < p.K) {
            const size_t off = (size_t)(p0 + lr) * p.K + k;
            if (p.in_scale) {
                const float4 s = *(const float4*)(p.in_scale + k), t = *(const float4*)(p.in_shift + k);
                a.x = fmaf(a.x, s.x, t.x); a.y = fmaf(a.y, s.y, t.y); a.z = fmaf(a.z, s.z, t.z); a.w = fmaf(a.w, s.w, t.w);
            }
            if (p.in_relu6) { a.x = fminf(fmaxf(a.x, 0.f), 6.f); a.y = fminf(fmaxf(a.y, 0.f), 6.f); a.z = fminf(fmaxf(a.z, 0.f), 6.f); a.w = fminf(fmaxf(a.w, 0.f), 6.f); }
            if (p.in_res) { const float4 r = *(const float4*)(p.in_res + off); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
            if (p.x_out && blockIdx.y == 0) *(float4*)(p.x_out + off) = a;
        }
        __syncthreads();                                                           // previous chunk fully consumed
        As[(lk + 0) * (PW_BM + 4) + lr] = a.x; As[(lk + 1) * (PW_BM + 4) + lr] = a.y; As[(lk + 2) * (PW_BM + 4) + lr] = a.z; As[(lk + 3) * (PW_BM + 4) + lr] = a.w;
        Bs[(lk + 0) * (PW_BN + 4) + lr] = b.x; Bs[(lk + 1) * (PW_BN + 4) + lr] = b.y; Bs[(lk + 2) * (PW_BN + 4) + lr] = b.z; Bs[(lk + 3) * (PW_BN + 4) + lr] = b.w;
        __syncthreads();
        if (k0 + PW_BK < p.K) gload(k0 + PW_BK);
#pragma unroll
        for (int kk = 0; kk < PW_BK; ++kk) {
            const float4 av = *(const float4*)(As + kk * (PW_BM + 4) + ty * 4), bv = *(const float4*)(Bs + kk * (PW_BN + 4) + tx * 4);
            const float am[4] = {av.x, av.y, av.z, av.w}, bn[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(am[i], bn[j], acc[i][j]);
        }
    }
    __syncthreads();
    float* Ct = sm;                                                                // [PW_BM][PW_BN + 4]
#pragma unroll
    for (int i = 0; i < 4; ++i) *(float4*)(Ct + (ty * 4 + i) * (PW_BN + 4) + tx * 4) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {                                               // coalesced rows: 16 float4 per position
        const int idx = tid + it * 256, row = idx >> 4, c4 = (idx & 15) * 4;
        if (p0 + row < p.P && n0 + c4 < p.N) *(float4*)(p.y + (size_t)(p0 + row) * p.N + n0 + c4) = *(const float4*)(Ct + row * (PW_BN + 4) + c4);
    }
    if (p.part) {
        // channel c of the tile: 4 threads (q) take 16 positions each (mean, then M2 around it), thread q == 0 merges them
        __shared__ float st[3][4][64];
        const int c = tid & 63, q = tid >> 6;
        float cnt = 0.f, s = 0.f;
        for (int i = 0; i < 16; ++i) { const int row = q * 16 + i; if (p0 + row < p.P) { s += Ct[row * (PW_BN + 4) + c]; cnt += 1.f; } }
        const float mean = cnt > 0.f ? s / cnt : 0.f;
        float m2 = 0.f;
        for (int i = 0; i < 16; ++i) { const int row = q * 16 + i; if (p0 + row < p.P) { const float d = Ct[row * (PW_BN + 4) + c] - mean; m2 = fmaf(d, d, m2); } }
        st[0][q][c] = cnt; st[1][q][c] = mean; st[2][q][c] = m2;
        __syncthreads();
        if (q == 0 && n0 + c < p.N) {
            float na = st[0][0][c], ma = st[1][0][c], qa = st[2][0][c];
            for (int r = 1; r < 4; ++r) bn_merge(na, ma, qa, st[0][r][c], st[1][r][c], st[2][r][c]);
            const int G4 = p.N >> 2, T = gridDim.x * G4;
            const size_t o = (size_t)blockIdx.x * G4 + ((n0 + c) >> 2);
            const int j = c & 3;
            if (j == 0) p.part[o] = na;
            p.part[(size_t)(1 + j) * T + o] = ma;
            p.part[(size_t)(5 + j) * T + o] = qa;
        }
    }
}

// Few positions (<= PW_ROWS_MAXP: the 16x16 / 8x8 maps of the late blocks): a 64 x 64 tile grid would be a handful of workgroups, each
// walking the whole contraction alone.  Here a workgroup takes 8 positions (their activated inputs staged in LDS) and each of its waves
// one output channel at a time, the lanes splitting the contraction in float4 steps and reducing by shuffles -- the weight rows stream
// from L2 once per 8 positions, and the grid is (N / 4 channels) x (P / 8 position tiles) workgroups.
#define PW_RT 8
#define PW_ROWS_MAXP 2048
#define PW_ROWS_MAXK 2048
__global__ __launch_bounds__(256) void pwconv_rows_kernel(PwParams p) {
    extern __shared__ __attribute__((aligned(16))) float xs[];                     // [PW_RT][K]
    const int r0 = blockIdx.y * PW_RT, rt = min(PW_RT, p.P - r0);
    const int K4 = p.K >> 2;
    for (int i = threadIdx.x; i < rt * K4; i += 256) {
        const int r = i / K4, k = (i - r * K4) * 4;
        const size_t off = (size_t)(r0 + r) * p.K + k;
        float4 a = *(const float4*)(p.x + off);
        if (p.in_scale) {
            const float4 s = *(const float4*)(p.in_scale + k), t = *(const float4*)(p.in_shift + k);
            a.x = fmaf(a.x, s.x, t.x); a.y = fmaf(a.y, s.y, t.y); a.z = fmaf(a.z, s.z, t.z); a.w = fmaf(a.w, s.w, t.w);
        }
        if (p.in_relu6) { a.x = fminf(fmaxf(a.x, 0.f), 6.f); a.y = fminf(fmaxf(a.y, 0.f), 6.f); a.z = fminf(fmaxf(a.z, 0.f), 6.f); a.w = fminf(fmaxf(a.w, 0.f), 6.f); }
        if (p.in_res) { const float4 r4 = *(const float4*)(p.in_res + off); a.x += r4.x; a.y += r4.y; a.z += r4.z; a.w += r4.w; }
        if (p.x_out && blockIdx.x == 0) *(float4*)(p.x_out + off) = a;
        *(float4*)(xs + r * p.K + k) = a;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = blockIdx.x * 4 + wave; n < p.N; n += gridDim.x * 4) {
        float acc[PW_RT];
#pragma unroll
        for (int b = 0; b < PW_RT; ++b) acc[b] = 0.f;
        const float* wr = p.w + (size_t)n * p.K;
        for (int k = lane * 4; k < p.K; k += 256) {
            const float4 wv = *(const float4*)(wr + k);
#pragma unroll
            for (int b = 0; b < PW_RT; ++b) {
                if (b < rt) {
                    const float4 xv = *(const float4*)(xs + b * p.K + k);
                    acc[b] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[b]))));
                }
            }
        }
#pragma unroll
        for (int b = 0; b < PW_RT; ++b)
            for (int o = 32; o > 0; o >>= 1) acc[b] += __shfl_down(acc[b], o, 64);
        if (lane == 0) {
            float s = 0.f;
            for (int b = 0; b < rt; ++b) { p.y[(size_t)(r0 + b) * p.N + n] = acc[b]; s += acc[b]; }
            if (p.part) {
                const float mean = s / (float)rt;
                float m2 = 0.f;
                for (int b = 0; b < rt; ++b) { const float d = acc[b] - mean; m2 = fmaf(d, d, m2); }
                const int G4 = p.N >> 2, T = gridDim.y * G4, j = n & 3;
                const size_t o = (size_t)blockIdx.y * G4 + (n >> 2);
                if (j == 0) p.part[o] = (float)rt;
                p.part[(size_t)(1 + j) * T + o] = mean;
                p.part[(size_t)(5 + j) * T + o] = m2;
            }
        }
    }
}

static bool pw_use_rows(long long P, int K, int N) {
    return P <= PW_ROWS_MAXP && K <= PW_ROWS_MAXK && ((P + PW_BM - 1) / PW_BM) * ((N + PW_BN - 1) / PW_BN) < 96;
}

extern "C" int lp_pwconv_stat_rows(long long P, int K, int N) {
    return pw_use_rows(P, K, N) ? (int)((P + PW_RT - 1) / PW_RT) : (int)((P + PW_BM - 1) / PW_BM);
}

extern "C" int lp_pwconv_fwd(const float* x, const float* w, float* y, const float* in_scale, const float* in_shift, int in_relu6,
                             const float* in_res, float* x_out, float* stats_part, long long P, int K, int N, void* stream) {
    if (!x || !w || !y) return lp_set_error(LP_ERR_ARG, "lp_pwconv_fwd: null pointer");
    if ((K & 3) || (N & 3) || P < 1 || (!in_scale != !in_shift)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_pwconv_fwd: needs K % 4 == 0, N % 4 == 0");
    PwParams p;
    p.x = x; p.w = w; p.y = y; p.in_scale = in_scale; p.in_shift = in_shift; p.in_res = in_res; p.x_out = x_out; p.part = stats_part;
    p.P = (int)P; p.K = K; p.N = N; p.in_relu6 = in_relu6;
    if (pw_use_rows(P, K, N)) {
        const int tiles = (int)((P + PW_RT - 1) / PW_RT);
        int bx = (N + 3) / 4; if (bx > 512) bx = 512;      // (fewer, fatter workgroups were measured slower: a wave's channel loop is latency bound)
        dim3 grid((unsigned)bx, (unsigned)tiles);
        hipLaunchKernelGGL(pwconv_rows_kernel, grid, dim3(256), (size_t)PW_RT * K * sizeof(float), (hipStream_t)stream, p);
        return lp_check_launch("pwconv_rows");
    }
    dim3 grid((unsigned)((P + PW_BM - 1) / PW_BM), (unsigned)((N + PW_BN - 1) / PW_BN));
    hipLaunchKernelGGL(pwconv_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return lp_check_launch("pwconv_fwd");
}

// ---- depthwise 3x3 with the BatchNorm statistics of its output folded in (train mode) -----------------------------------------------
// Same arithmetic as dwconv3x3_kernel; the item loop has the shape of bn_partial_kernel (a thread keeps one group of 4 channels), so
// the count / mean / M2 partials of the outputs come out of the same pass.
__global__ __launch_bounds__(256) void dwconv3x3_stats_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ sc,
                                                              const float* __restrict__ sh, float* __restrict__ y, float* __restrict__ part,
                                                              int N, int H, int W, int C, int stride, int BT, int T) {
    __shared__ float shm[9][256];
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, G4 = C >> 2;
    const long long items = (long long)N * Ho * Wo * G4, step = (long long)gridDim.x * BT;
    float cnt = 0.f, ref[4] = {0, 0, 0, 0}, sd[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    if ((int)threadIdx.x < BT) {
        const int g = threadIdx.x % G4, c = g * 4;
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sc) { s = *(const float4*)(sc + c); t = *(const float4*)(sh + c); }
        float wk[9][4];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) wk[k][j] = w[(c + j) * 9 + k];
        for (long long i = (long long)blockIdx.x * BT + threadIdx.x; i < items; i += step) {
            long long pix = i / G4;
            const int xo = (int)(pix % Wo); pix /= Wo;
            const int yo = (int)(pix % Ho); const int n = (int)(pix / Ho);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = yo * stride + ky - 1, ix = xo * stride + kx - 1;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                        float4 v = *(const float4*)(x + (((size_t)n * H + iy) * W + ix) * C + c);
                        if (sc) {
                            v.x = fminf(fmaxf(fmaf(v.x, s.x, t.x), 0.f), 6.f); v.y = fminf(fmaxf(fmaf(v.y, s.y, t.y), 0.f), 6.f);
                            v.z = fminf(fmaxf(fmaf(v.z, s.z, t.z), 0.f), 6.f); v.w = fminf(fmaxf(fmaf(v.w, s.w, t.w), 0.f), 6.f);
                        }
                        const int k = ky * 3 + kx;
                        acc.x = fmaf(v.x, wk[k][0], acc.x); acc.y = fmaf(v.y, wk[k][1], acc.y);
                        acc.z = fmaf(v.z, wk[k][2], acc.z); acc.w = fmaf(v.w, wk[k][3], acc.w);
                    }
                }
            *(float4*)(y + (((size_t)n * Ho + yo) * Wo + xo) * C + c) = acc;
            if (cnt == 0.f) { ref[0] = acc.x; ref[1] = acc.y; ref[2] = acc.z; ref[3] = acc.w; }
            float d;
            d = acc.x - ref[0]; sd[0] += d; sq[0] = fmaf(d, d, sq[0]);
            d = acc.y - ref[1]; sd[1] += d; sq[1] = fmaf(d, d, sq[1]);
            d = acc.z - ref[2]; sd[2] += d; sq[2] = fmaf(d, d, sq[2]);
            d = acc.w - ref[3]; sd[3] += d; sq[3] = fmaf(d, d, sq[3]);
            cnt += 1.f;
        }
    }
    float mean[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mean[j] = 0.f; m2[j] = 0.f;
        if (cnt > 0.f) { mean[j] = ref[j] + sd[j] / cnt; m2[j] = fmaxf(sq[j] - sd[j] * sd[j] / cnt, 0.f); }
    }
    shm[0][threadIdx.x] = cnt;
#pragma unroll
    for (int j = 0; j < 4; ++j) { shm[1 + j][threadIdx.x] = mean[j]; shm[5 + j][threadIdx.x] = m2[j]; }
    __syncthreads();
    if ((int)threadIdx.x < G4) {
        for (int k = threadIdx.x + G4; k < BT; k += G4) {
            const float nb = shm[0][k];
            float na = cnt;
#pragma unroll
            for (int j = 0; j < 4; ++j) { na = cnt; bn_merge(na, mean[j], m2[j], nb, shm[1 + j][k], shm[5 + j][k]); }
            cnt = na;
        }
        const size_t o = (size_t)blockIdx.x * G4 + threadIdx.x;
        part[o] = cnt;
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[(size_t)(1 + j) * T + o] = mean[j]; part[(size_t)(5 + j) * T + o] = m2[j]; }
    }
}

static int dw_stat_blocks(long long items, int G4) {
    const int BT = 256 - 256 % G4;
    long long nb = (items + 4ll * BT - 1) / (4ll * BT);
    return (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}

extern "C" int lp_dwconv_stat_rows(int N, int H, int W, int C, int stride) {
    if ((C & 3) || (C >> 2) > 256) return 0;
    return dw_stat_blocks((long long)N * ((H + stride - 1) / stride) * ((W + stride - 1) / stride) * (C >> 2), C >> 2);
}

extern "C" int lp_dwconv3x3_stats_fwd(const float* x, const float* w, const float* in_scale, const float* in_shift, float* y, float* stats_part,
                                      int N, int H, int W, int C, int stride, void* stream) {
    if (!x || !w || !y || !stats_part) return lp_set_error(LP_ERR_ARG, "lp_dwconv3x3_stats_fwd: null pointer");
    if ((C & 3) || (C >> 2) > 256 || (stride != 1 && stride != 2) || (!in_scale != !in_shift))
        return lp_set_error(LP_ERR_UNSUPPORTED, "lp_dwconv3x3_stats_fwd: needs C % 4 == 0, C <= 1024, stride 1|2");
    const int G4 = C >> 2, BT = 256 - 256 % G4;
    const int blocks = lp_dwconv_stat_rows(N, H, W, C, stride);
    hipLaunchKernelGGL(dwconv3x3_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, in_scale, in_shift, y, stats_part, N, H, W, C,
                       stride, BT, blocks * G4);
    return lp_check_launch("dwconv3x3_stats_fwd");
}

// BatchNorm (scale, shift) + running-statistics update from `rows` partials per channel group (lp_pwconv_fwd / lp_dwconv3x3_stats_fwd)
extern "C" int lp_bn_finalize(const float* stats_part, int rows, const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float* scale, float* shift, int C, float eps, float momentum, void* stream) {
    if (!stats_part || !gamma || !beta || !scale || !shift) return lp_set_error(LP_ERR_ARG, "lp_bn_finalize: null pointer");
    if ((C & 3) || rows < 1 || (!running_mean != !running_var)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_finalize: needs C % 4 == 0, rows >= 1");
    const int G4 = C >> 2, T = rows * G4;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(G4), dim3(256), 0, (hipStream_t)stream, stats_part, gamma, beta, running_mean, running_var, scale,
                       shift, G4, T, T, eps, momentum);
    return lp_check_launch("bn_finalize");
}

// ---- depthwise 3x3 backward (meta-training trains the pose encoder: runners/holycow.py:34-41) -----------------------------------------
// Data gradient w.r.t. the ACTIVATED input a = act(x) (the BatchNorm + ReLU6 backward that follows is lp_norm_act_bwd on the raw x):
//   da[n,iy,ix,c] = sum_{ky,kx} dy[n,yo,xo,c] * w[c][ky][kx]   over the outputs with yo*stride + ky - 1 = iy, xo*stride + kx - 1 = ix
// a gather (every element written once, no atomics), 4 channels per thread.
__global__ __launch_bounds__(256) void dwconv3x3_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ da,
                                                              int N, int H, int W, int C, int stride) {
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        long long pix = i / C4;
        const int ix = (int)(pix % W); pix /= W;
        const int iy = (int)(pix % H); const int n = (int)(pix / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (stride == 2 && (ty & 1))) continue;
            const int yo = ty / stride;
            if (yo >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (stride == 2 && (tx & 1))) continue;
                const int xo = tx / stride;
                if (xo >= Wo) continue;
                const float4 d = *(const float4*)(dy + (((size_t)n * Ho + yo) * Wo + xo) * C + c);
                const int k = ky * 3 + kx;
                acc.x = fmaf(d.x, w[(c + 0) * 9 + k], acc.x); acc.y = fmaf(d.y, w[(c + 1) * 9 + k], acc.y);
                acc.z = fmaf(d.z, w[(c + 2) * 9 + k], acc.z); acc.w = fmaf(d.w, w[(c + 3) * 9 + k], acc.w);
            }
        }
        *(float4*)(da + (size_t)i * 4) = acc;
    }
}

extern "C" int lp_dwconv3x3_dgrad(const float* dy, const float* w, float* da, int N, int H, int W, int C, int stride, void* stream) {
    if (!dy || !w || !da) return lp_set_error(LP_ERR_ARG, "lp_dwconv3x3_dgrad: null pointer");
    if ((C & 3) || (stride != 1 && stride != 2)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_dwconv3x3_dgrad: needs C % 4 == 0, stride 1|2");
    const long long total = (long long)N * H * W * (C / 4);
    long long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dwconv3x3_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, w, da, N, H, W, C, stride);
    return lp_check_launch("dwconv3x3_dgrad");
}

// Weight gradient: dw[c][k] = sum_{n,yo,xo} dy[n,yo,xo,c] * act(x)[n, yo*s+ky-1, xo*s+kx-1, c],  act = clamp(x*sc[c]+sh[c], 0, 6) | identity
// (recomputed from the raw input while it is loaded, as in the forward).  Pass 1: the grid's thread count per block is trimmed to a
// multiple of the C/4 channel quads (C/4 <= 256), so a thread meets the same 4 channels on every step and keeps 9 x 4 partial sums in
// registers; the block folds its threads per quad through LDS -> part[block][9][C].  Pass 2: fixed-order sum over the blocks.
// Round 6: a work item is a SEGMENT of DWW_SEG consecutive output pixels of one row (not one pixel): the 3 x 3 window of activated inputs slides
// along the row in registers, so a stride-1 layer loads 3 new input values per output pixel instead of 9 (the re-reads went through L1 / L2 and
// bounded the large layers: 160 B of cache traffic per channel quad and pixel for 32 B of tensor), and the activation is applied once per
// loaded value.  Stride 2: the window advances by two columns, 6 loads per pixel.
#define DWW_MAXB 512
#define DWW_SEG 16
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                                              const float* __restrict__ dy, float* __restrict__ part, int N, int H, int W, int C,
                                                              int stride, int BT) {
    __shared__ float red[36][256];
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, C4 = C >> 2;
    const int nseg = (Wo + DWW_SEG - 1) / DWW_SEG;
    const long long total = (long long)N * Ho * nseg * C4;
    float acc[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
    const bool active = (int)threadIdx.x < BT;
    if (active) {
        const int c = ((int)threadIdx.x % C4) * 4;                      // BT % C4 == 0 and the grid stride is a multiple of C4: fixed quad
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool aff = sc != nullptr;
        if (aff) { s = *(const float4*)(sc + c); t = *(const float4*)(sh + c); }
        auto load = [&](const float* row, int ix) -> float4 {           // act(x) at column ix of an in-image row (row == nullptr: padding row)
            if (row == nullptr || ix < 0 || ix >= W) return make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v = *(const float4*)(row + (size_t)ix * C);
            if (aff) {
                v.x = fminf(fmaxf(fmaf(v.x, s.x, t.x), 0.f), 6.f); v.y = fminf(fmaxf(fmaf(v.y, s.y, t.y), 0.f), 6.f);
                v.z = fminf(fmaxf(fmaf(v.z, s.z, t.z), 0.f), 6.f); v.w = fminf(fmaxf(fmaf(v.w, s.w, t.w), 0.f), 6.f);
            }
            return v;
        };
        for (long long i = (long long)blockIdx.x * BT + threadIdx.x; i < total; i += (long long)gridDim.x * BT) {
            long long r = i / C4;
            const int sg = (int)(r % nseg); r /= nseg;
            const int yo = (int)(r % Ho); const int n = (int)(r / Ho);
            const int x0 = sg * DWW_SEG, x1 = min(x0 + DWW_SEG, Wo);
            const float* rows[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = yo * stride + ky - 1;
                rows[ky] = (iy >= 0 && iy < H) ? x + ((size_t)n * H + iy) * W * C + c : nullptr;
            }
            const float* drow = dy + (((size_t)n * Ho + yo) * Wo) * C + c;
            float4 win[3][3];                                           // win[ky][kx] = act(x)[yo*s + ky - 1][xo*s + kx - 1]
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                win[ky][1] = load(rows[ky], x0 * stride - 1);           // (shifted into place by the first step of the loop)
                win[ky][2] = load(rows[ky], x0 * stride);
            }
            for (int xo = x0; xo < x1; ++xo) {
                const float4 d = *(const float4*)(drow + (size_t)xo * C);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if (stride == 1) {
                        win[ky][0] = win[ky][1]; win[ky][1] = win[ky][2];
                        win[ky][2] = load(rows[ky], xo + 1);
                    } else {
                        win[ky][0] = (xo == x0) ? win[ky][1] : win[ky][2];      // column 2*xo - 1 = the previous pixel's column 2*(xo-1) + 1
                        win[ky][1] = (xo == x0) ? win[ky][2] : load(rows[ky], xo * 2);
                        win[ky][2] = load(rows[ky], xo * 2 + 1);
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int k = ky * 3 + kx;
                        const float4 v = win[ky][kx];
                        acc[k][0] = fmaf(d.x, v.x, acc[k][0]); acc[k][1] = fmaf(d.y, v.y, acc[k][1]);
                        acc[k][2] = fmaf(d.z, v.z, acc[k][2]); acc[k][3] = fmaf(d.w, v.w, acc[k][3]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[k * 4 + j][threadIdx.x] = acc[k][j];
    __syncthreads();
    // fold the BT / C4 threads of every quad: thread q < C4 sums entries q, q + C4, ... for its 36 values
    if ((int)threadIdx.x < C4) {
        for (int e = 0; e < 36; ++e) {
            float a = 0.f;
            for (int r = threadIdx.x; r < BT; r += C4) a += red[e][r];
            const int k = e >> 2, j = e & 3;
            part[((size_t)blockIdx.x * 9 + k) * C + threadIdx.x * 4 + j] = a;
        }
    }
}

// Pass 2: dw[c][k] = sum over the blocks' partials in a FIXED order.  32 (k, c) columns x 8 row lanes per workgroup: lane r sums blocks r, r + 8, ...
// on four independent accumulators, the eight lanes are folded in order through LDS (round 6: one thread per column walked all <= 512 partials
// through two accumulators -- 4 .. 34 workgroups of dependent L2 loads, 30 us per layer).
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int blocks, int C) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, lr = threadIdx.x >> 5;
    const int n = 9 * C;
    const int idx = blockIdx.x * 32 + col;                              // (k, c), c fastest: part[b][k][c] = part[b * 9C + idx]
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (idx < n) {
        const float* p = part + idx;
        int b = lr;
        for (; b + 24 < blocks; b += 32) {
            a0 += p[(size_t)b * n]; a1 += p[(size_t)(b + 8) * n]; a2 += p[(size_t)(b + 16) * n]; a3 += p[(size_t)(b + 24) * n];
        }
        for (; b < blocks; b += 8) a0 += p[(size_t)b * n];
    }
    red[lr][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (lr == 0 && idx < n) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) sum += red[r][col];
        dw[(size_t)(idx % C) * 9 + idx / C] = sum;
    }
}

static int dww_blocks(long long total, int BT) {
    long long nb = (total + BT - 1) / BT;                                // (items are DWW_SEG-pixel row segments: one per thread where there are enough)
    return (int)(nb > DWW_MAXB ? DWW_MAXB : (nb < 1 ? 1 : nb));
}

extern "C" long long lp_dwconv3x3_wgrad_workspace_bytes(int C) { return (long long)DWW_MAXB * 9 * C * (long long)sizeof(float); }

extern "C" int lp_dwconv3x3_wgrad(const float* x, const float* in_scale, const float* in_shift, const float* dy, float* dw, float* workspace,
                                  int N, int H, int W, int C, int stride, void* stream) {
    if (!x || !dy || !dw || !workspace) return lp_set_error(LP_ERR_ARG, "lp_dwconv3x3_wgrad: null pointer");
    if ((C & 3) || C > 1024 || (stride != 1 && stride != 2) || (!in_scale != !in_shift))
        return lp_set_error(LP_ERR_UNSUPPORTED, "lp_dwconv3x3_wgrad: needs C % 4 == 0, C <= 1024, stride 1|2");
    const int C4 = C >> 2, BT = 256 - 256 % C4;
    const int Wo_ = (W + stride - 1) / stride;
    const long long total = (long long)N * ((H + stride - 1) / stride) * ((Wo_ + DWW_SEG - 1) / DWW_SEG) * C4;
    const int blocks = dww_blocks(total, BT);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3(blocks), dim3(256), 0, st, x, in_scale, in_shift, dy, workspace, N, H, W, C, stride, BT);
    int rc = lp_check_launch("dwconv3x3_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(dwconv3x3_wgrad_reduce_kernel, dim3((9 * C + 31) / 32), dim3(256), 0, st, workspace, dw, blocks, C);
    return lp_check_launch("dwconv3x3_wgrad_reduce");
}

#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif
