// Forward pass of the MobileNetV2 pose encoder (reference: embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26-28,56-58,
// torchvision's mobilenet_v2) on gfx950 -- the layers that are NOT dense contractions.  The 1x1 (pointwise) convs run on
// lp_conv16_fwd; everything here is bandwidth-bound fp32, NHWC, 16 bytes per lane:
//   stem_conv_s2   3x3 stride-2 conv of the NCHW RGB frame, 3 -> 32 channels (27 MACs per output: no matrix work)
//   dwconv3x3      depthwise 3x3, stride 1 | 2, with the producer's BatchNorm + ReLU6 applied while its input is loaded
//                  (clamp(x*scale[c]+shift[c], 0, 6); the conv's zero padding is applied after that activation)
//   affine_res     x = y*scale[c] + shift[c] (+ residual): the linear BatchNorm that ends an inverted-residual block, also emitting
//                  the 16-bit operand planes of x for the next block's 1x1 expand conv
//   affine_relu6_mean   global average pool of relu6(BatchNorm(y)) -> [N][C] (input of the classifier)
//   bn_running_update   running_mean / running_var momentum update from batch statistics (train-mode BatchNorm)
// BatchNorm appears as a per-channel (scale, shift): from the running statistics in eval mode, from lp_instnorm_stats over the whole
// batch (N*H*W positions per channel) in train mode.
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

// train-mode BatchNorm bookkeeping: running_mean <- (1-m) running_mean + m mean;  running_var <- (1-m) running_var + m var * n/(n-1),
// var = 1/rstd^2 - eps (lp_instnorm_stats hands out the biased batch variance as rstd)
__global__ void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ rm, float* __restrict__ rv,
                                         int C, float momentum, float eps, float unbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float r = rstd[c];
    const float var = 1.f / (r * r) - eps;
    rm[c] = (1.f - momentum) * rm[c] + momentum * mean[c];
    rv[c] = (1.f - momentum) * rv[c] + momentum * var * unbias;
}

extern "C" int lp_bn_running_update(const float* mean, const float* rstd, float* running_mean, float* running_var, int C, float momentum,
                                    float eps, long long count, void* stream) {
    if (!mean || !rstd || !running_mean || !running_var) return lp_set_error(LP_ERR_ARG, "lp_bn_running_update: null pointer");
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, rstd, running_mean, running_var, C,
                       momentum, eps, unbias);
    return lp_check_launch("bn_running_update");
}
