// Layers around the contractions of the ResNeXt-50 32x4d identity encoder (embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:
// 26-28,37-54 = torchvision resnext50_32x4d(num_classes=512)) for gfx950 -- all bandwidth-bound, one pass each:
//   * lp_im2col_planes     7x7 / stride 2 stem as a 1x1 contraction: the 147 taps of every output pixel as one 16-bit operand row
//   * lp_bn_relu_maxpool   BatchNorm affine + ReLU + MaxPool2d(3, 2, 1) forward (fp32 + operand planes + 1-byte argmax), backward
//   * lp_bn_add_act        BatchNorm affine of the block output + identity / BatchNorm'ed downsample branch + ReLU (+ operand planes)
//   * lp_subsample2 / lp_zero_stuff2 / lp_add_strided2   stride-2 plumbing (pick / adjoint of pick) on fp32 tensors and operand planes
//   * lp_spatial_mean      AdaptiveAvgPool2d(1) forward / backward
//   * lp_pack_grouped      weight image of the block-diagonal grouped 3x3 conv (lp_gconv16_fwd)
// The contractions themselves (1x1 convs, the grouped 3x3, the stem on its im2col rows, the classifier) run on conv_dma.hip /
// conv_wgrad.hip; the BatchNorm statistics and BatchNorm backward on the instance-norm kernels of elementwise.hip (a BatchNorm over
// [P][C] is an instance norm with one "image" of P pixels).
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

static inline unsigned grid_for(long long items, int cap = 16384) {
    long long b = (items + 255) / 256;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

template <bool F16, bool SPLIT>
__device__ __forceinline__ void store_op8(const float (&v)[8], uint16_t* hi, uint16_t* lo, size_t off8) {
    s16x8_t h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint16_t hb = lp_f32_to_op16<F16>(v[j]);
        h[j] = (short)hb;
        if (SPLIT) l[j] = (short)lp_f32_to_op16<false>(v[j] - lp_op16_to_f32<false>(hb));
    }
    *(s16x8_t*)(hi + off8) = h;
    if (SPLIT) *(s16x8_t*)(lo + off8) = l;
}

// 8 consecutive channels of a tensor that lives either as fp32 or -- the fp16 mode's 16-bit-resident conv outputs ("y16": the unscaled
// fp16 plane the conv epilogue wrote instead of fp32 y) -- as one fp16 operand plane
template <bool H16> __device__ __forceinline__ void load8(const void* base, size_t off, float (&v)[8]) {
    if (H16) {
        const s16x8_t q = *(const s16x8_t*)((const uint16_t*)base + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lp_op16_to_f32<true>((uint16_t)q[j]);
    } else {
        const float4 a = *(const float4*)((const float*)base + off), b = *(const float4*)((const float*)base + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
// ReLU pattern of 8 channels from a 16-bit plane: > 0  <=>  sign clear and magnitude non-zero
__device__ __forceinline__ void mask8_16(const uint16_t* plane, size_t off, float (&g)[8]) {
    const s16x8_t q = *(const s16x8_t*)(plane + off);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = ((unsigned)((uint16_t)q[j]) - 1u) < 0x7fffu ? g[j] : 0.f;
}

// ---- im2col rows as operand planes -----------------------------------------------------------------------------------------------
// x [N][C][H][W] fp32 (NCHW, what the dataloader delivers) -> rows [N*Ho*Wo][K8], K = C*KS*KS, k = (c*KS + ky)*KS + kx (the order of
// nn.Conv2d's weight.view(Cout, -1)), zero padding, pad columns zero.  One thread per (pixel, 8 consecutive k): the row of a pixel is
// K8*2 contiguous bytes, so a wave writes whole rows (coalesced); the gathers hit L1/L2 (every input value is read KS*KS/stride^2 times).
template <int PREC>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                     long long items, int C, int H, int W, int Ho, int Wo, int KS, int stride, int pad,
                                                     int K, int G) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    const int KK = KS * KS;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        const long long pix = i / G;
        const int ox = (int)(pix % Wo);
        const long long t = pix / Wo;
        const int oy = (int)(t % Ho), n = (int)(t / Ho);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = g * 8 + j;
            float q = 0.f;
            if (k < K) {
                const int c = k / KK, r = k - c * KK;
                const int ky = r / KS, kx = r - ky * KS;
                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) q = x[(((size_t)n * C + c) * H + iy) * W + ix];
            }
            v[j] = q;
        }
        store_op8<F16, SPLIT>(v, hi, lo, (size_t)i * 8);
    }
}

extern "C" int lp_im2col_planes(const float* x, uint16_t* hi, uint16_t* lo, int N, int C, int H, int W, int ksize, int stride, int pad,
                                int prec, void* stream) {
    if (!x || !hi) return lp_set_error(LP_ERR_ARG, "lp_im2col_planes: null pointer");
    if (prec == LP_PREC_BF16X3 && !lo) return lp_set_error(LP_ERR_ARG, "lp_im2col_planes: bf16x3 needs the lo plane");
    if (ksize < 1 || stride < 1 || pad < 0) return lp_set_error(LP_ERR_ARG, "lp_im2col_planes: bad geometry");
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const int K = C * ksize * ksize, G = (K + 7) / 8;
    const long long items = (long long)N * Ho * Wo * G;
    if (items <= 0) return LP_OK;
    hipStream_t st = (hipStream_t)stream;
#define LP_I2C(P) hipLaunchKernelGGL(im2col_kernel<P>, dim3(grid_for(items, 65536)), dim3(256), 0, st, x, hi, lo, items, C, H, W, Ho, Wo, ksize, stride, pad, K, G)
    if (prec == LP_PREC_BF16) LP_I2C(LP_PREC_BF16);
    else if (prec == LP_PREC_BF16X3) LP_I2C(LP_PREC_BF16X3);
    else if (prec == LP_PREC_F16) LP_I2C(LP_PREC_F16);
    else return lp_set_error(LP_ERR_ARG, "lp_im2col_planes: unknown precision mode");
#undef LP_I2C
    return lp_check_launch("im2col_planes");
}

// ---- BatchNorm affine + ReLU + MaxPool2d(kernel 3, stride 2, padding 1) -----------------------------------------------------------
// y [N][H][W][C] raw conv output -> out [N][Ho][Wo][C] = max over the window of relu(y*scale[c]+shift[c]) (padding = -inf, i.e. ignored),
// the operand planes of out, and idx = position 0..8 (ky*3+kx) of the first maximum in row-major scan order (torch's tie rule).
template <int PREC>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                                              float* __restrict__ out, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                              unsigned char* __restrict__ idx, long long items, int H, int W, int Ho, int Wo, int C) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    const int C4 = C >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int ox = (int)(pix % Wo);
        const long long t = pix / Wo;
        const int oy = (int)(t % Ho), n = (int)(t / Ho);
        const float4 s = *(const float4*)(sc + c), b = *(const float4*)(sh + c);
        float m[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        int am[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const float4 q = *(const float4*)(y + (((size_t)n * H + iy) * W + ix) * C + c);
                    const float a0 = fmaxf(fmaf(q.x, s.x, b.x), 0.f), a1 = fmaxf(fmaf(q.y, s.y, b.y), 0.f);
                    const float a2 = fmaxf(fmaf(q.z, s.z, b.z), 0.f), a3 = fmaxf(fmaf(q.w, s.w, b.w), 0.f);
                    const int k = ky * 3 + kx;
                    if (a0 > m[0]) { m[0] = a0; am[0] = k; }
                    if (a1 > m[1]) { m[1] = a1; am[1] = k; }
                    if (a2 > m[2]) { m[2] = a2; am[2] = k; }
                    if (a3 > m[3]) { m[3] = a3; am[3] = k; }
                }
            }
        }
        *(float4*)(out + (size_t)i * 4) = make_float4(m[0], m[1], m[2], m[3]);
        if (idx) *(uchar4*)(idx + (size_t)i * 4) = make_uchar4((unsigned char)am[0], (unsigned char)am[1], (unsigned char)am[2], (unsigned char)am[3]);
        if (hi) {
            ushort4 h, l;
            uint16_t* hp = (uint16_t*)&h; uint16_t* lp = (uint16_t*)&l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hp[j] = lp_f32_to_op16<F16>(m[j]);
                if (SPLIT) lp[j] = lp_f32_to_op16<false>(m[j] - lp_op16_to_f32<false>(hp[j]));
            }
            *(ushort4*)(hi + (size_t)i * 4) = h;
            if (SPLIT) *(ushort4*)(lo + (size_t)i * 4) = l;
        }
    }
}

extern "C" int lp_bn_relu_maxpool_fwd(const float* y, const float* scale, const float* shift, float* out, uint16_t* hi, uint16_t* lo,
                                      unsigned char* idx, int N, int H, int W, int C, int prec, void* stream) {
    if (!y || !scale || !shift || !out) return lp_set_error(LP_ERR_ARG, "lp_bn_relu_maxpool_fwd: null pointer");
    if ((C & 7) || H < 2 || W < 2) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_relu_maxpool_fwd: C % 8 == 0, H, W >= 2");
    if (hi && prec == LP_PREC_BF16X3 && !lo) return lp_set_error(LP_ERR_ARG, "lp_bn_relu_maxpool_fwd: bf16x3 planes need lo");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long items = (long long)N * Ho * Wo * (C >> 2);
    hipStream_t st = (hipStream_t)stream;
#define LP_MP(P) hipLaunchKernelGGL(bn_relu_maxpool_kernel<P>, dim3(grid_for(items, 65536)), dim3(256), 0, st, y, scale, shift, out, hi, lo, idx, items, H, W, Ho, Wo, C)
    if (prec == LP_PREC_BF16) LP_MP(LP_PREC_BF16);
    else if (prec == LP_PREC_BF16X3) LP_MP(LP_PREC_BF16X3);
    else if (prec == LP_PREC_F16) LP_MP(LP_PREC_F16);
    else return lp_set_error(LP_ERR_ARG, "lp_bn_relu_maxpool_fwd: unknown precision mode");
#undef LP_MP
    return lp_check_launch("bn_relu_maxpool");
}

// dA [N][H][W][C] (gradient w.r.t. relu(bn(y)), the maxpool input) = sum over the <= 4 windows containing the pixel of d_out where the
// window's recorded argmax is this pixel.  A gather: no atomics, every element written exactly once.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx, float* __restrict__ dA,
                                                          long long items, int H, int W, int Ho, int Wo, int C) {
    const int C4 = C >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int ix = (int)(pix % W);
        const long long t = pix / W;
        const int iy = (int)(t % H), n = (int)(t / H);
        // windows oy with 2*oy-1 <= iy <= 2*oy+1: iy even -> oy = iy/2 (ky 1); iy odd -> oy = (iy-1)/2 (ky 2) and (iy+1)/2 (ky 0)
        const int oy0 = iy >> 1, ky0 = (iy & 1) ? 2 : 1, ny = (iy & 1) ? 2 : 1;
        const int ox0 = ix >> 1, kx0 = (ix & 1) ? 2 : 1, nx = (ix & 1) ? 2 : 1;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < ny; ++p) {
            const int oy = oy0 + p, ky = p ? 0 : ky0;
            if (oy >= Ho) continue;
            for (int q = 0; q < nx; ++q) {
                const int ox = ox0 + q, kx = q ? 0 : kx0;
                if (ox >= Wo) continue;
                const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + c;
                const uchar4 id = *(const uchar4*)(idx + o);
                const float4 d = *(const float4*)(dout + o);
                const int k = ky * 3 + kx;
                a.x += (id.x == k) ? d.x : 0.f; a.y += (id.y == k) ? d.y : 0.f;
                a.z += (id.z == k) ? d.z : 0.f; a.w += (id.w == k) ? d.w : 0.f;
            }
        }
        *(float4*)(dA + (size_t)i * 4) = a;
    }
}

extern "C" int lp_maxpool_bwd(const float* dout, const unsigned char* idx, float* dA, int N, int H, int W, int C, void* stream) {
    if (!dout || !idx || !dA) return lp_set_error(LP_ERR_ARG, "lp_maxpool_bwd: null pointer");
    if (C & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_maxpool_bwd: C % 4 == 0");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long items = (long long)N * H * W * (C >> 2);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(items, 65536)), dim3(256), 0, (hipStream_t)stream, dout, idx, dA, items, H, W, Ho, Wo, C);
    return lp_check_launch("maxpool_bwd");
}

// ---- block output: out = act( y*scale[c]+shift[c] + r ),  r = res | res*rscale[c]+rshift[c] | 0;  act = ReLU | identity -------------
// RES16 (round 6, bf16 / bf16x3): r is read from the OPERAND PLANES of the block input (rhi [+ rlo]: hi + lo carries 16 significant bits -- the
// same values the block's first conv multiplies) and ``out`` may be NULL: an identity bottleneck of a bf16x3 network then neither reads nor
// writes an fp32 copy of a block output (16 -> 12 B per element here)
template <int PREC, bool Y16, bool RES16 = false>
__global__ __launch_bounds__(256) void bn_add_act_kernel(const void* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                                         const float* __restrict__ res, const float* __restrict__ rsc, const float* __restrict__ rsh,
                                                         float* __restrict__ out, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                         long long items, int C, int relu,
                                                         const uint16_t* __restrict__ rhi = nullptr, const uint16_t* __restrict__ rlo = nullptr) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    const int G = C >> 3;
    const float floor_v = relu ? 0.f : -3.0e38f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % G) * 8;
        float v[8];
        load8<Y16>(y, (size_t)i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[c + j], sh[c + j]);
        if (RES16) {
            const s16x8_t qh = *(const s16x8_t*)(rhi + i * 8);
            s16x8_t ql = qh;
            if (SPLIT) ql = *(const s16x8_t*)(rlo + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float r = lp_op16_to_f32<false>((uint16_t)qh[j]);
                if (SPLIT) r += lp_op16_to_f32<false>((uint16_t)ql[j]);
                v[j] += r;
            }
        } else if (res) {
            const float4 r0 = *(const float4*)(res + i * 8), r1 = *(const float4*)(res + i * 8 + 4);
            float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            if (rsc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = fmaf(r[j], rsc[c + j], rsh[c + j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], floor_v);
        if (out) {
            *(float4*)(out + i * 8) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(out + i * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (hi) store_op8<F16, SPLIT>(v, hi, lo, (size_t)i * 8);
    }
}

static int bn_add_act_impl(const float* y, const uint16_t* y16, const float* scale, const float* shift, const float* res, const float* res_scale,
                           const float* res_shift, float* out, uint16_t* hi, uint16_t* lo, long long P, int C, int relu, int prec, hipStream_t st) {
    if ((!y && !y16) || !scale || !shift || (!out && !hi)) return lp_set_error(LP_ERR_ARG, "lp_bn_add_act: null pointer");      // (out NULL: planes only, ABI 12)
    if (!res_scale != !res_shift || (res_scale && !res)) return lp_set_error(LP_ERR_ARG, "lp_bn_add_act: res_scale/res_shift go together and need res");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_add_act: C must be a multiple of 8");
    if (hi && prec == LP_PREC_BF16X3 && !lo) return lp_set_error(LP_ERR_ARG, "lp_bn_add_act: bf16x3 planes need lo");
    if (y16 && prec != LP_PREC_F16) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_add_act16: 16-bit y exists in the fp16 mode only");
    const long long items = P * (C >> 3);
    if (items == 0) return LP_OK;
#define LP_BA(Q, H, SRC) hipLaunchKernelGGL((bn_add_act_kernel<Q, H>), dim3(grid_for(items)), dim3(256), 0, st, (const void*)(SRC), scale, shift, res, res_scale, res_shift, out, hi, lo, items, C, relu)
    if (y16) LP_BA(LP_PREC_F16, true, y16);
    else if (prec == LP_PREC_BF16) LP_BA(LP_PREC_BF16, false, y);
    else if (prec == LP_PREC_BF16X3) LP_BA(LP_PREC_BF16X3, false, y);
    else if (prec == LP_PREC_F16) LP_BA(LP_PREC_F16, false, y);
    else return lp_set_error(LP_ERR_ARG, "lp_bn_add_act: unknown precision mode");
#undef LP_BA
    return lp_check_launch("bn_add_act");
}

extern "C" int lp_bn_add_act(const float* y, const float* scale, const float* shift, const float* res, const float* res_scale,
                             const float* res_shift, float* out, uint16_t* hi, uint16_t* lo, long long P, int C, int relu, int prec,
                             void* stream) {
    if (!y) return lp_set_error(LP_ERR_ARG, "lp_bn_add_act: null pointer");
    return bn_add_act_impl(y, nullptr, scale, shift, res, res_scale, res_shift, out, hi, lo, P, C, relu, prec, (hipStream_t)stream);
}

// identity bottleneck of a bf16 / bf16x3 network: the residual from the operand planes of the block input, fp32 ``out`` optional (ABI 12)
extern "C" int lp_bn_add_act_planes(const float* y, const float* scale, const float* shift, const uint16_t* res_hi, const uint16_t* res_lo,
                                    float* out, uint16_t* hi, uint16_t* lo, long long P, int C, int relu, int prec, void* stream) {
    if (!y || !scale || !shift || !res_hi || !hi) return lp_set_error(LP_ERR_ARG, "lp_bn_add_act_planes: null pointer");
    if (prec != LP_PREC_BF16 && prec != LP_PREC_BF16X3)
        return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_add_act_planes: bf16 / bf16x3 planes only (an fp16 plane has 11 bits: not a residual stream)");
    if (prec == LP_PREC_BF16X3 && (!res_lo || !lo)) return lp_set_error(LP_ERR_ARG, "lp_bn_add_act_planes: bf16x3 planes need lo");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_add_act_planes: C must be a multiple of 8");
    const long long items = P * (C >> 3);
    if (items == 0) return LP_OK;
    const float* nof = nullptr;
    if (prec == LP_PREC_BF16X3)
        hipLaunchKernelGGL((bn_add_act_kernel<LP_PREC_BF16X3, false, true>), dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const void*)y, scale, shift,
                           nof, nof, nof, out, hi, lo, items, C, relu, res_hi, res_lo);
    else
        hipLaunchKernelGGL((bn_add_act_kernel<LP_PREC_BF16, false, true>), dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const void*)y, scale, shift,
                           nof, nof, nof, out, hi, lo, items, C, relu, res_hi, res_lo);
    return lp_check_launch("bn_add_act_planes");
}

// y as the fp16 plane a conv epilogue wrote instead of fp32 (fp16 mode)
extern "C" int lp_bn_add_act16(const uint16_t* y16, const float* scale, const float* shift, const float* res, const float* res_scale,
                               const float* res_shift, float* out, uint16_t* hi, long long P, int C, int relu, void* stream) {
    if (!y16) return lp_set_error(LP_ERR_ARG, "lp_bn_add_act16: null pointer");
    return bn_add_act_impl(nullptr, y16, scale, shift, res, res_scale, res_shift, out, hi, nullptr, P, C, relu, LP_PREC_F16, (hipStream_t)stream);
}

// ---- BatchNorm affine (+ ReLU) of a 16-bit-resident conv output, to the operand planes of the next conv (fp16 mode) ----------------
// a[p][c] = fp16( relu?( y16[p][c] * scale[c] + shift[c] ) ): lp_act_pack's prologues 4 / 5 reading 2 B instead of 4 B per element
__global__ __launch_bounds__(256) void bn_act16_kernel(const uint16_t* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                                       uint16_t* __restrict__ out, long long items, int C, int relu) {
    const int G = C >> 3;
    const float floor_v = relu ? 0.f : -3.0e38f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % G) * 8;
        float v[8];
        load8<true>(y, (size_t)i * 8, v);
        const float4 s0 = *(const float4*)(sc + c), s1 = *(const float4*)(sc + c + 4), h0 = *(const float4*)(sh + c), h1 = *(const float4*)(sh + c + 4);
        const float a8[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, b8[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], a8[j], b8[j]), floor_v);
        store_op8<true, false>(v, out, nullptr, (size_t)i * 8);
    }
}

extern "C" int lp_bn_act16(const uint16_t* y16, const float* scale, const float* shift, uint16_t* out_hi, long long P, int C, int relu,
                           void* stream) {
    if (!y16 || !scale || !shift || !out_hi) return lp_set_error(LP_ERR_ARG, "lp_bn_act16: null pointer");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_act16: C must be a multiple of 8");
    const long long items = P * (C >> 3);
    if (items == 0) return LP_OK;
    hipLaunchKernelGGL(bn_act16_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, y16, scale, shift, out_hi, items, C, relu);
    return lp_check_launch("bn_act16");
}

// ---- AdaIN (+ ReLU) of a 16-bit-resident conv output with PER-IMAGE (n, c) affines (the generator's instance norms: blocks.py:18-26,70-73) --
// a[n][p][c] = fp16( relu?( y16[n][p][c] * scale[n][c] + shift[n][c] ) ): lp_act_pack's prologue 1 reading 2 B instead of 4 B per element
__global__ __launch_bounds__(256) void adain_act16_kernel(const uint16_t* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                                          uint16_t* __restrict__ out, long long items, long long items_per_image, int C, int relu) {
    const int G = C >> 3;
    const float floor_v = relu ? 0.f : -3.0e38f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % G) * 8;
        const size_t a = (size_t)(i / items_per_image) * C + c;
        float v[8];
        load8<true>(y, (size_t)i * 8, v);
        const float4 s0 = *(const float4*)(sc + a), s1 = *(const float4*)(sc + a + 4), h0 = *(const float4*)(sh + a), h1 = *(const float4*)(sh + a + 4);
        const float a8[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, b8[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], a8[j], b8[j]), floor_v);
        store_op8<true, false>(v, out, nullptr, (size_t)i * 8);
    }
}

extern "C" int lp_adain_act16(const uint16_t* y16, const float* scale, const float* shift, uint16_t* out_hi, int N, long long HW, int C, int relu,
                              void* stream) {
    if (!y16 || !scale || !shift || !out_hi) return lp_set_error(LP_ERR_ARG, "lp_adain_act16: null pointer");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_adain_act16: C must be a multiple of 8");
    const long long per = HW * (C >> 3), items = per * N;
    if (items == 0) return LP_OK;
    hipLaunchKernelGGL(adain_act16_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, y16, scale, shift, out_hi, items, per, C, relu);
    return lp_check_launch("adain_act16");
}

// ---- stride-2 plumbing on [N][H][W][row of `units` 16-byte pieces] tensors (fp32 NHWC: units = C/4; operand planes: units = C8/8) ----
// subsample: out[n,i,j] = in[n,2i,2j]  (what a stride-2 1x1 conv reads; the stride-2 3x3 conv output picked from the stride-1 result)
__global__ __launch_bounds__(256) void subsample2_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long long items, int Ho, int Wo,
                                                         int H, int W, int units) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int u = (int)(i % units);
        const long long pix = i / units;
        const int ox = (int)(pix % Wo);
        const long long t = pix / Wo;
        const int oy = (int)(t % Ho), n = (int)(t / Ho);
        out[i] = in[(((size_t)n * H + 2 * oy) * W + 2 * ox) * units + u];
    }
}
// zero stuffing (adjoint of subsample): out[n,y,x] = (y, x both even) ? in[n,y/2,x/2] : 0 -- out is [N][H][W], in [N][ceil(H/2)][ceil(W/2)]
__global__ __launch_bounds__(256) void zero_stuff2_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long long items, int Hi, int Wi,
                                                          int H, int W, int units) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int u = (int)(i % units);
        const long long pix = i / units;
        const int x = (int)(pix % W);
        const long long t = pix / W;
        const int y = (int)(t % H), n = (int)(t / H);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (!((x | y) & 1)) v = in[(((size_t)n * Hi + (y >> 1)) * Wi + (x >> 1)) * units + u];
        out[i] = v;
    }
}
// d[n,2i,2j,:] += s[n,i,j,:]  (fp32; adjoint of the pick inside a sum: the stride-2 downsample branch's data gradient joins the main one)
__global__ __launch_bounds__(256) void add_strided2_kernel(float4* __restrict__ d, const float4* __restrict__ s, long long items, int Hs, int Ws,
                                                           int H, int W, int C4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int u = (int)(i % C4);
        const long long pix = i / C4;
        const int ox = (int)(pix % Ws);
        const long long t = pix / Ws;
        const int oy = (int)(t % Hs), n = (int)(t / Hs);
        const size_t o = (((size_t)n * H + 2 * oy) * W + 2 * ox) * C4 + u;
        float4 a = d[o];
        const float4 b = s[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        d[o] = a;
    }
}

extern "C" int lp_subsample2(const void* in, void* out, int N, int H, int W, int row_bytes, void* stream) {
    if (!in || !out) return lp_set_error(LP_ERR_ARG, "lp_subsample2: null pointer");
    if (row_bytes & 15) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_subsample2: rows must be multiples of 16 bytes");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, units = row_bytes >> 4;
    const long long items = (long long)N * Ho * Wo * units;
    if (items == 0) return LP_OK;
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, items, Ho, Wo, H, W, units);
    return lp_check_launch("subsample2");
}
extern "C" int lp_zero_stuff2(const void* in, void* out, int N, int H, int W, int row_bytes, void* stream) {
    if (!in || !out) return lp_set_error(LP_ERR_ARG, "lp_zero_stuff2: null pointer");
    if (row_bytes & 15) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_zero_stuff2: rows must be multiples of 16 bytes");
    const int Hi = (H + 1) / 2, Wi = (W + 1) / 2, units = row_bytes >> 4;
    const long long items = (long long)N * H * W * units;
    if (items == 0) return LP_OK;
    hipLaunchKernelGGL(zero_stuff2_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, items, Hi, Wi, H, W, units);
    return lp_check_launch("zero_stuff2");
}
extern "C" int lp_add_strided2(float* d, const float* s, int N, int H, int W, int C, void* stream) {
    if (!d || !s) return lp_set_error(LP_ERR_ARG, "lp_add_strided2: null pointer");
    if (C & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_add_strided2: C % 4 == 0");
    const int Hs = (H + 1) / 2, Ws = (W + 1) / 2;
    const long long items = (long long)N * Hs * Ws * (C >> 2);
    if (items == 0) return LP_OK;
    hipLaunchKernelGGL(add_strided2_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (float4*)d, (const float4*)s, items, Hs, Ws, H, W, C >> 2);
    return lp_check_launch("add_strided2");
}

// ---- AdaptiveAvgPool2d(1): out[n][c] = mean over HW of x[n][p][c];  backward: dx[n][p][c] = g[n][c] / HW -----------------------------
__global__ __launch_bounds__(256) void spatial_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int C) {
    __shared__ float red[4][64];
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
    float a = 0.f;
    if (c < C) for (int p = pl; p < HW; p += 4) a += x[((size_t)n * HW + p) * C + c];
    red[pl][threadIdx.x & 63] = a;
    __syncthreads();
    if (pl == 0 && c < C) out[(size_t)n * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)HW;
}
__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ dx, long long items, int HW, int C4, float inv) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int u = (int)(i % C4);
        const long long n = i / ((long long)C4 * HW);
        const float4 v = g[n * C4 + u];
        dx[i] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
}
extern "C" int lp_spatial_mean_fwd(const float* x, float* out, int N, int HW, int C, void* stream) {
    if (!x || !out) return lp_set_error(LP_ERR_ARG, "lp_spatial_mean_fwd: null pointer");
    hipLaunchKernelGGL(spatial_mean_kernel, dim3((C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, x, out, HW, C);
    return lp_check_launch("spatial_mean");
}
extern "C" int lp_spatial_mean_bwd(const float* g, float* dx, int N, int HW, int C, void* stream) {
    if (!g || !dx) return lp_set_error(LP_ERR_ARG, "lp_spatial_mean_bwd: null pointer");
    if (C & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_spatial_mean_bwd: C % 4 == 0");
    const long long items = (long long)N * HW * (C >> 2);
    if (items == 0) return LP_OK;
    hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (float4*)dx, items, HW, C >> 2,
                       1.0f / (float)HW);
    return lp_check_launch("spatial_mean_bwd");
}

// ---- weight image of the block-diagonal grouped 3x3 conv -------------------------------------------------------------------------
// w [C][cg][9] (nn.Conv2d(C, C, 3, groups = C/cg) layout).  mode 0 (forward):  out[t][co][cl]   = w[co][ci % cg][t]   for ci = 64*(co/64) + cl
// in co's group, else 0;  mode 1 (data gradient): out[8-t][ci][cl] = w[co][ci % cg][t] for co = 64*(ci/64) + cl in ci's group, else 0.
// Rows padded to CP (zero).
__global__ __launch_bounds__(256) void pack_grouped_kernel(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                           int C, int cg, int CP, int mode, int f16, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cl = (int)(i & 63);
        const long long r = i >> 6;
        const int row = (int)(r % CP), t = (int)(r / CP);
        float v = 0.f;
        if (row < C) {
            const int other = (row & ~63) + cl;              // the channel of the OTHER side (input for mode 0, output for mode 1)
            if (other < C && other / cg == row / cg) {
                const int co = mode ? other : row, ci = mode ? row : other;
                const int ts = mode ? 8 - t : t;
                v = w[((size_t)co * cg + (ci % cg)) * 9 + ts];
            }
        }
        uint16_t hb;
        if (f16) hb = lp_f32_to_op16<true>(v); else hb = lp_f32_to_op16<false>(v);
        hi[i] = hb;
        if (lo) lo[i] = lp_f32_to_op16<false>(v - lp_op16_to_f32<false>(hb));
    }
}
extern "C" int lp_pack_grouped(const float* w, uint16_t* hi, uint16_t* lo, int C, int group_size, int CP, int mode, int f16, void* stream) {
    if (!w || !hi) return lp_set_error(LP_ERR_ARG, "lp_pack_grouped: null pointer");
    if ((C & 63) || group_size < 1 || 64 % group_size || CP < C) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_pack_grouped: C % 64 == 0, group size dividing 64");
    const long long total = 9ll * CP * 64;
    hipLaunchKernelGGL(pack_grouped_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, hi, f16 ? nullptr : lo, C, group_size, CP, mode, f16, total);
    return lp_check_launch("pack_grouped");
}

// ---- BatchNorm (+ activation) backward straight to operand planes ------------------------------------------------------------------
// The embedder's conv -> BatchNorm -> ReLU chains only ever consume the BatchNorm input gradient dy as the 16-bit operand of the weight /
// data gradient contractions, so it is never written in fp32:
//   pass 1 (partial):  per (pixel split, channel)  S1 = sum g, S2 = sum g*xhat, max|g|, max|xhat|     g = dA * activation mask (recomputed, not stored)
//   pass 2 (finalize): dgamma = S2, dbeta = S1, coefficients of dy = ca*g + cb*x + cc, and a per-channel BOUND of |dy|
//                      (|ca| * (max|g| + |S1|/P + max|xhat| * |S2|/P)) from which the fp16 gradient scale is taken (a power of two putting the
//                      bound into [2^12, 2^13): the true amax is at most the bound, so the scaled planes cannot overflow)
//   pass 3 (apply):    recomputes g from dA and the mask, writes dy * s as operand planes (hi [, lo]); block 0 publishes {s, 1/s}
// 8 B read in pass 1 and 8 B read + 2 B written in pass 3 per element (fp16 mode), against 30 B for partial + apply + lp_act_pack on fp32 dy.
// mask modes as lp_norm_act_bwd: 0 own activation 0 < x*scale+shift < act_hi, 1 none, 2 mask_src > 0; 3: a 16-bit operand plane > 0.
// X16: x is the fp16 plane of a 16-bit-resident conv output (6 B read per element and pass instead of 8 B).
template <bool X16>
__global__ __launch_bounds__(256) void bn_bwd16_partial_kernel(const float* __restrict__ dA, const void* __restrict__ x, const void* __restrict__ mask_src,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               float* __restrict__ part, long long P, int C, int mask_mode, float act_hi, int PB) {
    __shared__ float sh[4][16][64];
    const int cb = blockIdx.y, s = blockIdx.x;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = cb * 64 + cq * 4;
    const long long p0 = (long long)s * PB, p1 = min(P, p0 + PB);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, gm[4] = {0, 0, 0, 0}, xm[4] = {0, 0, 0, 0};
    if (c < C) {
        const float4 mu = *(const float4*)(mean + c), rs = *(const float4*)(rstd + c);
        const float4 sc = *(const float4*)(scale + c), sf = *(const float4*)(shift + c);
        const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, r4[4] = {rs.x, rs.y, rs.z, rs.w}, a4[4] = {sc.x, sc.y, sc.z, sc.w}, b4[4] = {sf.x, sf.y, sf.z, sf.w};
#pragma unroll 4
        for (long long pix = p0 + pl; pix < p1; pix += 16) {
            const float4 gv = *(const float4*)(dA + pix * C + c);
            float xs[4], g[4] = {gv.x, gv.y, gv.z, gv.w};
            if (X16) {
                const s16x4_t q = *(const s16x4_t*)((const uint16_t*)x + pix * C + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) xs[j] = lp_op16_to_f32<true>((uint16_t)q[j]);
            } else {
                const float4 xv = *(const float4*)((const float*)x + pix * C + c);
                xs[0] = xv.x; xs[1] = xv.y; xs[2] = xv.z; xs[3] = xv.w;
            }
            if (mask_mode == 2) {
                const float4 mk = *(const float4*)((const float*)mask_src + pix * C + c);
                const float k4[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = k4[j] > 0.f ? g[j] : 0.f;
            } else if (mask_mode == 3) {
                const s16x4_t mk = *(const s16x4_t*)((const uint16_t*)mask_src + pix * C + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = ((unsigned)((uint16_t)mk[j]) - 1u) < 0x7fffu ? g[j] : 0.f;
            } else if (mask_mode == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float a = fmaf(xs[j], a4[j], b4[j]); g[j] = (a > 0.f && a < act_hi) ? g[j] : 0.f; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xs[j] - m4[j]) * r4[j];
                s1[j] += g[j]; s2[j] = fmaf(g[j], xh, s2[j]);
                gm[j] = fmaxf(gm[j], fabsf(g[j])); xm[j] = fmaxf(xm[j], fabsf(xh));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[0][pl][cq * 4 + j] = s1[j]; sh[1][pl][cq * 4 + j] = s2[j]; sh[2][pl][cq * 4 + j] = gm[j]; sh[3][pl][cq * 4 + j] = xm[j]; }
    __syncthreads();
    {
        const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
        float a = 0.f;
        if (which < 2) { for (int k = 0; k < 16; ++k) a += sh[which][k][ch]; }
        else { for (int k = 0; k < 16; ++k) a = fmaxf(a, sh[which][k][ch]); }
        const int cg = cb * 64 + ch;
        if (cg < C) part[((size_t)s * C + cg) * 4 + which] = a;
    }
}

// one wave per channel: merges the S partials; coef[c] = {ca, cb, cc}, bound[c] >= max|dy| of the channel
__global__ __launch_bounds__(256) void bn_bwd16_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ coef, float* __restrict__ bound, int C, int S, float inv_p, int frozen) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double a1 = 0, a2 = 0;
    float gm = 0.f, xm = 0.f;
    for (int s = lane; s < S; s += 64) {
        const float* q = part + ((size_t)s * C + c) * 4;
        a1 += q[0]; a2 += q[1]; gm = fmaxf(gm, q[2]); xm = fmaxf(xm, q[3]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64);
        gm = fmaxf(gm, __shfl_down(gm, o, 64)); xm = fmaxf(xm, __shfl_down(xm, o, 64));
    }
    if (lane != 0) return;
    const float S1 = (float)a1, S2 = (float)a2;
    dgamma[c] = S2; dbeta[c] = S1;
    const float r = rstd[c], m = mean[c], ca = gamma[c] * r;
    const float cb = frozen ? 0.f : -ca * r * S2 * inv_p;
    const float cc = frozen ? 0.f : -ca * S1 * inv_p - cb * m;
    coef[c * 3 + 0] = ca; coef[c * 3 + 1] = cb; coef[c * 3 + 2] = cc;
    bound[c] = frozen ? fabsf(ca) * gm : fabsf(ca) * (gm + fabsf(S1) * inv_p + xm * fabsf(S2) * inv_p);
}

template <int PREC, bool X16>
__global__ __launch_bounds__(256) void bn_bwd16_apply_kernel(const float* __restrict__ dA, const void* __restrict__ x, const void* __restrict__ mask_src,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ coef,
                                                             const float* __restrict__ bound, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                             float* __restrict__ out_scale, long long items, int C, int mask_mode, float act_hi,
                                                             float* __restrict__ g_out) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    float sc_out = 1.f;
    if (F16) {       // every block derives the same power-of-two scale from the per-channel bounds (C floats, L2 resident)
        __shared__ float shm[4];
        float m = 0.f;
        for (int j = threadIdx.x; j < C; j += 256) m = fmaxf(m, bound[j]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
        if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
        float inv = 1.f;
        if (m > 0.f && m < 3.0e38f) {
            int e;
            (void)frexpf(m, &e);
            int k = 13 - e;
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            sc_out = ldexpf(1.f, k); inv = ldexpf(1.f, -k);
        }
        if (out_scale && blockIdx.x == 0 && threadIdx.x == 0) { out_scale[0] = sc_out; out_scale[1] = inv; }
    } else if (out_scale && blockIdx.x == 0 && threadIdx.x == 0) { out_scale[0] = 1.f; out_scale[1] = 1.f; }
    const int G = C >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % G) * 8;
        const float4 g0 = *(const float4*)(dA + i * 8), g1 = *(const float4*)(dA + i * 8 + 4);
        float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, xs[8];
        load8<X16>(x, (size_t)i * 8, xs);
        if (mask_mode == 2) {
            float k8[8];
            load8<false>(mask_src, (size_t)i * 8, k8);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = k8[j] > 0.f ? g[j] : 0.f;
        } else if (mask_mode == 3) {
            mask8_16((const uint16_t*)mask_src, (size_t)i * 8, g);
        } else if (mask_mode == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = fmaf(xs[j], scale[c + j], shift[c + j]); g[j] = (a > 0.f && a < act_hi) ? g[j] : 0.f; }
        }
        if (g_out) {          // the masked incoming gradient itself: what the identity branch of a residual block receives
            *(float4*)(g_out + i * 8) = make_float4(g[0], g[1], g[2], g[3]);
            *(float4*)(g_out + i * 8 + 4) = make_float4(g[4], g[5], g[6], g[7]);
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* cf = coef + (c + j) * 3;
            v[j] = fmaf(cf[0], g[j], fmaf(cf[1], xs[j], cf[2])) * sc_out;
        }
        store_op8<F16, SPLIT>(v, hi, lo, (size_t)i * 8);
    }
}

// pass 3 for C/8 <= 256 channel groups (every layer of the two encoders): a thread keeps ONE group of 8 channels -- its 40 coefficients
// (ca, cb, cc, scale, shift) are loaded once as float4s -- and walks the pixels of its block (256 / (C/8) pixels per iteration, the
// lanes of a pixel contiguous: coalesced 32 B per lane).  The one-item-per-thread form above re-reads 40 scalars per 8 elements.
template <int PREC, bool X16>
__global__ __launch_bounds__(256) void bn_bwd16_apply_rows_kernel(const float* __restrict__ dA, const void* __restrict__ x, const void* __restrict__ mask_src,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ coef,
                                                                  const float* __restrict__ bound, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                                  float* __restrict__ out_scale, long long P, int C, int rows, int PB, int mask_mode, float act_hi,
                                                                  float* __restrict__ g_out) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    float sc_out = 1.f;
    if (F16) {
        __shared__ float shm[4];
        float m = 0.f;
        for (int j = threadIdx.x; j < C; j += 256) m = fmaxf(m, bound[j]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
        if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
        float inv = 1.f;
        if (m > 0.f && m < 3.0e38f) {
            int e;
            (void)frexpf(m, &e);
            int k = 13 - e;
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            sc_out = ldexpf(1.f, k); inv = ldexpf(1.f, -k);
        }
        if (out_scale && blockIdx.x == 0 && threadIdx.x == 0) { out_scale[0] = sc_out; out_scale[1] = inv; }
    } else if (out_scale && blockIdx.x == 0 && threadIdx.x == 0) { out_scale[0] = 1.f; out_scale[1] = 1.f; }
    const int G = C >> 3;
    const int cg = threadIdx.x % G, r = threadIdx.x / G;
    if (r >= rows) return;
    const int c = cg * 8;
    float cf[24], a8[8], b8[8];
#pragma unroll
    for (int j = 0; j < 6; ++j) { const float4 q = *(const float4*)(coef + (size_t)c * 3 + j * 4); cf[j * 4] = q.x; cf[j * 4 + 1] = q.y; cf[j * 4 + 2] = q.z; cf[j * 4 + 3] = q.w; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 q = *(const float4*)(scale + c + j * 4), t = *(const float4*)(shift + c + j * 4);
        a8[j * 4] = q.x; a8[j * 4 + 1] = q.y; a8[j * 4 + 2] = q.z; a8[j * 4 + 3] = q.w;
        b8[j * 4] = t.x; b8[j * 4 + 1] = t.y; b8[j * 4 + 2] = t.z; b8[j * 4 + 3] = t.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { cf[j * 3] *= sc_out; cf[j * 3 + 1] *= sc_out; cf[j * 3 + 2] *= sc_out; }      // power of two: exact
    const long long p0 = (long long)blockIdx.x * PB, p1 = min(P, p0 + PB);
#pragma unroll 2
    for (long long pix = p0 + r; pix < p1; pix += rows) {
        const size_t off = (size_t)pix * C + c;
        const float4 g0 = *(const float4*)(dA + off), g1 = *(const float4*)(dA + off + 4);
        float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, xs[8];
        load8<X16>(x, off, xs);
        if (mask_mode == 2) {
            float k8[8];
            load8<false>(mask_src, off, k8);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = k8[j] > 0.f ? g[j] : 0.f;
        } else if (mask_mode == 3) {
            mask8_16((const uint16_t*)mask_src, off, g);
        } else if (mask_mode == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = fmaf(xs[j], a8[j], b8[j]); g[j] = (a > 0.f && a < act_hi) ? g[j] : 0.f; }
        }
        if (g_out) {
            *(float4*)(g_out + off) = make_float4(g[0], g[1], g[2], g[3]);
            *(float4*)(g_out + off + 4) = make_float4(g[4], g[5], g[6], g[7]);
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(cf[j * 3], g[j], fmaf(cf[j * 3 + 1], xs[j], cf[j * 3 + 2]));
        store_op8<F16, SPLIT>(v, hi, lo, off);
    }
}

extern "C" long long lp_bn_bwd16_workspace_bytes(long long P, int C) {
    const int PB = lp_stat_split_pix(1, P, C);
    const long long S = (P + PB - 1) / PB;
    return (S * C * 4 + (long long)C * 4) * (long long)sizeof(float);
}

static int bn_bwd16_impl(const float* dA, const float* x, const uint16_t* x16, const void* mask_src, const float* gamma, const float* mean,
                         const float* rstd, const float* scale, const float* shift, uint16_t* out_hi, uint16_t* out_lo, float* out_scale,
                         float* dgamma, float* dbeta, float* workspace, long long P, int C, int mask_mode, float act_hi, int frozen_stats,
                         int prec, float* g_out, hipStream_t st) {
    if (!dA || (!x && !x16) || !gamma || !mean || !rstd || !scale || !shift || !out_hi || !out_scale || !dgamma || !dbeta || !workspace)
        return lp_set_error(LP_ERR_ARG, "lp_bn_bwd16: null pointer");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_bwd16: C must be a multiple of 8");
    if (mask_mode < 0 || mask_mode > 3 || (mask_mode >= 2 && !mask_src)) return lp_set_error(LP_ERR_ARG, "lp_bn_bwd16: bad mask mode");
    if (prec == LP_PREC_BF16X3 && !out_lo) return lp_set_error(LP_ERR_ARG, "lp_bn_bwd16: bf16x3 needs the lo plane");
    if (x16 && prec != LP_PREC_F16) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_bn_bwd16_h: 16-bit x exists in the fp16 mode only");
    if (P < 1) return LP_OK;
    const int PB = lp_stat_split_pix(1, P, C);
    const int S = (int)((P + PB - 1) / PB);
    float* part = workspace;
    float* coef = workspace + (size_t)S * C * 4;
    float* bound = coef + (size_t)C * 3;
    const float hi_ = act_hi > 0.f ? act_hi : 3.0e38f;
    const void* xv = x16 ? (const void*)x16 : (const void*)x;
    if (x16) hipLaunchKernelGGL(bn_bwd16_partial_kernel<true>, dim3(S, (C + 63) / 64), dim3(256), 0, st, dA, xv, mask_src, mean, rstd, scale, shift, part, P, C, mask_mode, hi_, PB);
    else hipLaunchKernelGGL(bn_bwd16_partial_kernel<false>, dim3(S, (C + 63) / 64), dim3(256), 0, st, dA, xv, mask_src, mean, rstd, scale, shift, part, P, C, mask_mode, hi_, PB);
    int rc = lp_check_launch("bn_bwd16_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd16_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, gamma, mean, rstd, dgamma, dbeta, coef, bound, C, S,
                       1.0f / (float)P, frozen_stats);
    rc = lp_check_launch("bn_bwd16_finalize");
    if (rc) return rc;
    const long long items = P * (C >> 3);
    const int G = C >> 3;
    static const bool one_item = getenv("LP_BNB_ITEMS") != nullptr;          // A/B knob: the one-item-per-thread form
    if (G <= 256 && !one_item) {
        const int rows = 256 / G;
        long long ppb = (P + 4095) / 4096;                                   // <= 4096 workgroups, >= 4 pixel rows per thread where P allows
        if (ppb < 4ll * rows) ppb = 4ll * rows;
        ppb = (ppb + rows - 1) / rows * rows;
        const unsigned grid = (unsigned)((P + ppb - 1) / ppb);
#define LP_BR(Q, H) hipLaunchKernelGGL((bn_bwd16_apply_rows_kernel<Q, H>), dim3(grid), dim3(256), 0, st, dA, xv, mask_src, scale, shift, coef, bound, out_hi, out_lo, out_scale, P, C, rows, (int)ppb, mask_mode, hi_, g_out)
        if (x16) LP_BR(LP_PREC_F16, true);
        else if (prec == LP_PREC_BF16) LP_BR(LP_PREC_BF16, false);
        else if (prec == LP_PREC_BF16X3) LP_BR(LP_PREC_BF16X3, false);
        else if (prec == LP_PREC_F16) LP_BR(LP_PREC_F16, false);
        else return lp_set_error(LP_ERR_ARG, "lp_bn_bwd16: unknown precision mode");
#undef LP_BR
        return lp_check_launch("bn_bwd16_apply");
    }
#define LP_BB(Q, H) hipLaunchKernelGGL((bn_bwd16_apply_kernel<Q, H>), dim3(grid_for(items, 8192)), dim3(256), 0, st, dA, xv, mask_src, scale, shift, coef, bound, out_hi, out_lo, out_scale, items, C, mask_mode, hi_, g_out)
    if (x16) LP_BB(LP_PREC_F16, true);
    else if (prec == LP_PREC_BF16) LP_BB(LP_PREC_BF16, false);
    else if (prec == LP_PREC_BF16X3) LP_BB(LP_PREC_BF16X3, false);
    else if (prec == LP_PREC_F16) LP_BB(LP_PREC_F16, false);
    else return lp_set_error(LP_ERR_ARG, "lp_bn_bwd16: unknown precision mode");
#undef LP_BB
    return lp_check_launch("bn_bwd16_apply");
}

extern "C" int lp_bn_bwd16(const float* dA, const float* x, const float* mask_src, const float* gamma, const float* mean, const float* rstd,
                           const float* scale, const float* shift, uint16_t* out_hi, uint16_t* out_lo, float* out_scale, float* dgamma,
                           float* dbeta, float* workspace, long long P, int C, int mask_mode, float act_hi, int frozen_stats, int prec,
                           float* g_out, void* stream) {
    if (!x) return lp_set_error(LP_ERR_ARG, "lp_bn_bwd16: null pointer");
    if (mask_mode == 3) return lp_set_error(LP_ERR_ARG, "lp_bn_bwd16: mask mode 3 (16-bit plane) is lp_bn_bwd16_h's");
    return bn_bwd16_impl(dA, x, nullptr, mask_src, gamma, mean, rstd, scale, shift, out_hi, out_lo, out_scale, dgamma, dbeta, workspace, P, C,
                         mask_mode, act_hi, frozen_stats, prec, g_out, (hipStream_t)stream);
}

// x as fp32 (x16 = NULL) or as the fp16 plane a conv epilogue wrote instead of fp32 y (x = NULL, fp16 mode); mask modes 0 / 1 / 2 as
// lp_bn_bwd16 (2: mask_src fp32 > 0) and 3: mask_src = a 16-bit OPERAND PLANE [P][C] whose elements are > 0 (the block output's planes:
// 2 B instead of 4 B per element for the ReLU pattern behind the residual add)
extern "C" int lp_bn_bwd16_h(const float* dA, const float* x, const uint16_t* x16, const void* mask_src, const float* gamma, const float* mean,
                             const float* rstd, const float* scale, const float* shift, uint16_t* out_hi, uint16_t* out_lo, float* out_scale,
                             float* dgamma, float* dbeta, float* workspace, long long P, int C, int mask_mode, float act_hi, int frozen_stats,
                             int prec, float* g_out, void* stream) {
    return bn_bwd16_impl(dA, x, x16, mask_src, gamma, mean, rstd, scale, shift, out_hi, out_lo, out_scale, dgamma, dbeta, workspace, P, C,
                         mask_mode, act_hi, frozen_stats, prec, g_out, (hipStream_t)stream);
}
