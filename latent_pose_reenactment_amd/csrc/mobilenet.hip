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

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ rm, float* __restrict__ rv, float* __restrict__ scale,
                                                          float* __restrict__ shift, int G4, int Tu, int T, float eps, float momentum) {
    __shared__ float sh[9][256];
    const int g = blockIdx.x;
    float n = 0.f, mean[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0};
    for (int t = g + threadIdx.x * G4; t < Tu; t += 256 * G4) {
        const float nb = part[t];
        float na = n;
#pragma unroll
        for (int j = 0; j < 4; ++j) { na = n; bn_merge(na, mean[j], m2[j], nb, part[(size_t)(1 + j) * T + t], part[(size_t)(5 + j) * T + t]); }
        n = na;
    }
    sh[0][threadIdx.x] = n;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[1 + j][threadIdx.x] = mean[j]; sh[5 + j][threadIdx.x] = m2[j]; }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float nb = sh[0][threadIdx.x + o];
            float na = sh[0][threadIdx.x];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                na = sh[0][threadIdx.x];
                bn_merge(na, sh[1 + j][threadIdx.x], sh[5 + j][threadIdx.x], nb, sh[1 + j][threadIdx.x + o], sh[5 + j][threadIdx.x + o]);
            }
            sh[0][threadIdx.x] = na;
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const int j = threadIdx.x, c = g * 4 + j;
        const float cntf = sh[0][0], m = sh[1 + j][0], var = sh[5 + j][0] / cntf;
        const float sc = gamma[c] / sqrtf(var + eps);
        scale[c] = sc; shift[c] = beta[c] - m * sc;
        if (rm) {
            const float unbias = cntf > 1.f ? cntf / (cntf - 1.f) : 1.f;
            rm[c] = (1.f - momentum) * rm[c] + momentum * m;
            rv[c] = (1.f - momentum) * rv[c] + momentum * var * unbias;
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
