// Small-batch linear layers and the fixed-bbox crop of the identity criterion for gfx950 -- both bandwidth-bound, fp32.
//
// lp_linear_fwd / lp_linear_bwd: y = alpha * x W^T + bias for B <= 64 rows (the generator's AdaIN-parameter projector,
//   generators/vector_pose_unsupervised_segmentation_noBottleneck.py:96-101: 768 -> 768 -> 13056, and the critic's 512 -> 1 head,
//   discriminators/no_landmarks.py:88,105).  With B = 1..8 these are weight streams (42 MB fp32 for the projector), not GEMMs: one
//   wave owns a weight row, reads it once with 16-byte loads and multiplies it with the B input rows held in LDS; 1/sigma of the
//   spectral norm (alpha, device scalar) and the bias are applied in the same pass.  The backward pass streams W once more: it
//   writes the raw weight gradient dW[n,:] = sum_b g[b,n] x[b,:] row by row while accumulating dx[b,:] += g[b,n] W[n,:] in
//   registers (block partials + a small reduction), and emits the bias gradient.
//
// lp_grid_crop_fwd / lp_grid_crop_bwd: criterions/idt_embed.py:58-83 crop_and_resize -- per-sample bbox [t,b,l,r] -> affine grid
//   (align_corners=False) -> bilinear sampling with reflection padding, and its adjoint (fp32 atomics; the crop is a 1.8x zoom, so a
//   source pixel collects from ~3 output pixels).  Replaces affine_grid + grid_sample + their backward (4 ATen launches per image).
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

#define LIN_BT 8          // batch rows per pass (accumulators per lane)
#define LIN_MAXK 1024     // 8 batch rows x 1024 inputs: the backward pass keeps its dx accumulators (8 x 4 float4 per lane) in registers

// y[b][n] = alpha * sum_k x[b][k] w[n][k] + bias[n];  grid.x over groups of 4 rows (one per wave), grid.y over batch tiles of LIN_BT
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ alpha, float* __restrict__ y, int B, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [bt][K]
    const int b0 = blockIdx.y * LIN_BT, bt = min(LIN_BT, B - b0);
    for (int i = threadIdx.x; i < bt * K; i += 256) xs[i] = x[(size_t)b0 * K + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float al = alpha ? alpha[0] : 1.f;
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        float acc[LIN_BT];
#pragma unroll
        for (int b = 0; b < LIN_BT; ++b) acc[b] = 0.f;
        const float* wr = w + (size_t)n * K;
        for (int k = lane * 4; k < K; k += 256) {
            const float4 wv = *(const float4*)(wr + k);
#pragma unroll
            for (int b = 0; b < LIN_BT; ++b) {
                if (b < bt) {
                    const float4 xv = *(const float4*)(xs + b * K + k);
                    acc[b] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[b]))));
                }
            }
        }
#pragma unroll
        for (int b = 0; b < LIN_BT; ++b)
            for (int o = 32; o > 0; o >>= 1) acc[b] += __shfl_down(acc[b], o, 64);
        if (lane == 0) {
            const float bs = bias ? bias[n] : 0.f;
            for (int b = 0; b < bt; ++b) y[(size_t)(b0 + b) * N + n] = fmaf(acc[b], al, bs);
        }
    }
}

// one pass over the rows of W: dw[n][:] = sum_b g[b][n] x[b][:];  dx partial[block][b][:] += g[b][n] * w[n][:];  db[n] = sum_b g[b][n]
// (grid.x row groups; B <= LIN_BT per launch: the host loops over batch tiles, accumulating dw / db on the later ones)
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ g,
                                                         float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dxpart,
                                                         int B, int N, int K, int accumulate, int want_dx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];          // xs [B][K], then red [4][B][K] reused for the dx reduction
    float* xs = sm;
    for (int i = threadIdx.x; i < B * K; i += 256) xs[i] = x[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int KT = LIN_MAXK / 256;                                  // float4 columns per lane at the maximum K
    float4 dxa[LIN_BT][KT];
#pragma unroll
    for (int b = 0; b < LIN_BT; ++b)
#pragma unroll
        for (int j = 0; j < KT; ++j) dxa[b][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        float gv[LIN_BT];
        float gsum = 0.f;
#pragma unroll
        for (int b = 0; b < LIN_BT; ++b) { gv[b] = b < B ? g[(size_t)b * N + n] : 0.f; gsum += gv[b]; }
        if (db && lane == 0) db[n] = accumulate ? db[n] + gsum : gsum;
        const float* wr = w + (size_t)n * K;
        float* dwr = dw ? dw + (size_t)n * K : nullptr;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const int k = lane * 4 + j * 256;
            if (k < K) {
                if (want_dx) {
                    const float4 wv = *(const float4*)(wr + k);
#pragma unroll
                    for (int b = 0; b < LIN_BT; ++b) {
                        dxa[b][j].x = fmaf(gv[b], wv.x, dxa[b][j].x); dxa[b][j].y = fmaf(gv[b], wv.y, dxa[b][j].y);
                        dxa[b][j].z = fmaf(gv[b], wv.z, dxa[b][j].z); dxa[b][j].w = fmaf(gv[b], wv.w, dxa[b][j].w);
                    }
                }
                if (dwr) {
                    float4 o = accumulate ? *(const float4*)(dwr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int b = 0; b < LIN_BT; ++b) {
                        if (b < B) {
                            const float4 xv = *(const float4*)(xs + b * K + k);
                            o.x = fmaf(gv[b], xv.x, o.x); o.y = fmaf(gv[b], xv.y, o.y); o.z = fmaf(gv[b], xv.z, o.z); o.w = fmaf(gv[b], xv.w, o.w);
                        }
                    }
                    *(float4*)(dwr + k) = o;
                }
            }
        }
    }
    if (!want_dx) return;
    // block partial of dx: the 4 waves combine through LDS (xs is dead), then one row [B][K] per block goes to the workspace
    __syncthreads();
    float* red = sm;                                                    // [4][B][K]
#pragma unroll
    for (int b = 0; b < LIN_BT; ++b)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const int k = lane * 4 + j * 256;
            if (b < B && k < K) *(float4*)(red + ((size_t)wave * B + b) * K + k) = dxa[b][j];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < B * K; i += 256)
        dxpart[(size_t)blockIdx.x * B * K + i] = (red[i] + red[(size_t)B * K + i]) + (red[(size_t)2 * B * K + i] + red[(size_t)3 * B * K + i]);
}

__global__ __launch_bounds__(256) void linear_dx_reduce_kernel(const float* __restrict__ part, const float* __restrict__ alpha, float* __restrict__ dx,
                                                               int nblocks, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = 0;
    for (; r + 3 < nblocks; r += 4) {
        a0 += part[(size_t)r * total + i]; a1 += part[(size_t)(r + 1) * total + i];
        a2 += part[(size_t)(r + 2) * total + i]; a3 += part[(size_t)(r + 3) * total + i];
    }
    for (; r < nblocks; ++r) a0 += part[(size_t)r * total + i];
    dx[i] = ((a0 + a1) + (a2 + a3)) * (alpha ? alpha[0] : 1.f);
}

static int linear_blocks(int N) { int b = (N + 3) / 4; return b > 512 ? 512 : (b < 1 ? 1 : b); }

extern "C" int lp_linear_fwd(const float* x, const float* w, const float* bias, const float* alpha, float* y, int B, int N, int K, void* stream) {
    if (!x || !w || !y) return lp_set_error(LP_ERR_ARG, "lp_linear_fwd: null pointer");
    if (B < 1 || B > 64 || (K & 3) || K > 2 * LIN_MAXK) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_linear_fwd: needs 1 <= B <= 64, K % 4 == 0, K <= 2048");
    dim3 grid(linear_blocks(N), (B + LIN_BT - 1) / LIN_BT);
    hipLaunchKernelGGL(linear_fwd_kernel, grid, dim3(256), (size_t)LIN_BT * K * sizeof(float), (hipStream_t)stream, x, w, bias, alpha, y, B, N, K);
    return lp_check_launch("linear_fwd");
}

extern "C" long long lp_linear_bwd_workspace_bytes(int B, int N, int K) {
    (void)B;
    return (long long)linear_blocks(N) * LIN_BT * K * sizeof(float);
}

extern "C" int lp_linear_bwd(const float* x, const float* w, const float* g, const float* alpha, float* dx, float* dw, float* db,
                             float* workspace, int B, int N, int K, void* stream) {
    if (!x || !w || !g) return lp_set_error(LP_ERR_ARG, "lp_linear_bwd: null pointer");
    if (dx && !workspace) return lp_set_error(LP_ERR_ARG, "lp_linear_bwd: dx needs the workspace");
    if (B < 1 || B > 64 || (K & 3) || K > LIN_MAXK) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_linear_bwd: needs 1 <= B <= 64, K % 4 == 0, K <= 1024");
    hipStream_t st = (hipStream_t)stream;
    const int nb = linear_blocks(N);
    static thread_local int attr_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipGetDevice failed");
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void*)linear_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_dev = dev;
    }
    for (int b0 = 0; b0 < B; b0 += LIN_BT) {
        const int bt = (B - b0) < LIN_BT ? (B - b0) : LIN_BT;
        const size_t lds = (size_t)4 * bt * K * sizeof(float);
        if (lds > 160 * 1024) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_linear_bwd: K too large for the dx reduction in LDS");
        hipLaunchKernelGGL(linear_bwd_kernel, dim3(nb), dim3(256), lds, st, x + (size_t)b0 * K, w, g + (size_t)b0 * N, dw, db, workspace, bt, N, K,
                           b0 > 0 ? 1 : 0, dx ? 1 : 0);
        if (dx) hipLaunchKernelGGL(linear_dx_reduce_kernel, dim3((bt * K + 255) / 256), dim3(256), 0, st, workspace, alpha, dx + (size_t)b0 * K, nb, bt * K);
    }
    return lp_check_launch("linear_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// crop_and_resize (criterions/idt_embed.py:58-83): theta = [[(r-l)/W, 0, (l+r)/W - 1], [0, (b-t)/H, (t+b)/H - 1]];
// affine_grid(align_corners=False) + grid_sample(bilinear, reflection, align_corners=False)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float reflect_coord(float in, int size) {
    // torch grid_sample, align_corners=False: reflect about -0.5 and size-0.5, then clip to [0, size-1]
    const float mn = -0.5f, span = (float)size;
    in = fabsf(in - mn);
    const float extra = fmodf(in, span);
    const int flips = (int)floorf(in / span);
    float out = (flips & 1) ? (span - extra + mn) : (extra + mn);
    return fminf(fmaxf(out, 0.f), (float)(size - 1));
}

template <bool BWD>
__global__ __launch_bounds__(256) void grid_crop_kernel(const float* __restrict__ img, const float* __restrict__ boxes, float* __restrict__ out,
                                                        const float* __restrict__ dout, float* __restrict__ dimg,
                                                        int N, int C, int H, int W, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % Wo), i = (int)((idx / Wo) % Ho), n = (int)(idx / ((long long)Wo * Ho));
    const float t = boxes[n * 4 + 0], b = boxes[n * 4 + 1], l = boxes[n * 4 + 2], r = boxes[n * 4 + 3];
    const float xn = (2.f * j + 1.f) / Wo - 1.f, yn = (2.f * i + 1.f) / Ho - 1.f;              // affine_grid base coordinates
    const float gx = (r - l) / W * xn + ((l + r) / W - 1.f), gy = (b - t) / H * yn + ((t + b) / H - 1.f);
    const float ix = reflect_coord(((gx + 1.f) * W - 1.f) * 0.5f, W), iy = reflect_coord(((gy + 1.f) * H - 1.f) * 0.5f, H);
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
    const float fx = ix - x0, fy = iy - y0;
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    const bool x1ok = x0 + 1 < W, y1ok = y0 + 1 < H;
    for (int c = 0; c < C; ++c) {
        const size_t base = ((size_t)n * C + c) * H * W;
        const size_t o = (((size_t)n * C + c) * Ho + i) * Wo + j;
        if (!BWD) {
            float v = w00 * img[base + (size_t)y0 * W + x0];
            if (x1ok) v += w01 * img[base + (size_t)y0 * W + x0 + 1];
            if (y1ok) v += w10 * img[base + (size_t)(y0 + 1) * W + x0];
            if (x1ok && y1ok) v += w11 * img[base + (size_t)(y0 + 1) * W + x0 + 1];
            out[o] = v;
        } else {
            const float gq = dout[o];
            unsafeAtomicAdd(dimg + base + (size_t)y0 * W + x0, w00 * gq);
            if (x1ok) unsafeAtomicAdd(dimg + base + (size_t)y0 * W + x0 + 1, w01 * gq);
            if (y1ok) unsafeAtomicAdd(dimg + base + (size_t)(y0 + 1) * W + x0, w10 * gq);
            if (x1ok && y1ok) unsafeAtomicAdd(dimg + base + (size_t)(y0 + 1) * W + x0 + 1, w11 * gq);
        }
    }
}

extern "C" int lp_grid_crop_fwd(const float* images, const float* boxes, float* out, int N, int C, int H, int W, int Ho, int Wo, void* stream) {
    if (!images || !boxes || !out) return lp_set_error(LP_ERR_ARG, "lp_grid_crop_fwd: null pointer");
    const long long total = (long long)N * Ho * Wo;
    hipLaunchKernelGGL(grid_crop_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images, boxes, out,
                       nullptr, nullptr, N, C, H, W, Ho, Wo);
    return lp_check_launch("grid_crop_fwd");
}

extern "C" int lp_grid_crop_bwd(const float* dout, const float* boxes, float* dimages, int N, int C, int H, int W, int Ho, int Wo, void* stream) {
    if (!dout || !boxes || !dimages) return lp_set_error(LP_ERR_ARG, "lp_grid_crop_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dimages, 0, (size_t)N * C * H * W * sizeof(float), st) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipMemsetAsync failed");
    const long long total = (long long)N * Ho * Wo;
    hipLaunchKernelGGL(grid_crop_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, nullptr, boxes, nullptr, dout, dimages,
                       N, C, H, W, Ho, Wo);
    return lp_check_launch("grid_crop_bwd");
}
