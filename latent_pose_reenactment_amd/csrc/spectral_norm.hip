// Batched spectral normalisation for gfx950.  The reference wraps every conv/linear in the legacy
// torch.nn.utils.spectral_norm hook: per layer and per forward  v <- normalize(W^T u), u <- normalize(W v) (train mode, in place,
// no grad), sigma = u^T W v, W_eff = W / sigma -- about a dozen tiny ATen launches per layer (~1 200 per training step).
// Here ONE launch handles all layers of a module: blockIdx.x = layer (descriptor table in device memory), 1024 threads stream
// the layer's matrix twice.  sigma == |W v| after the u update, so no third mat-vec is needed.  1/sigma is handed to the conv
// kernels as their epilogue scale `alpha`, so W/sigma is never materialised.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

struct SnDesc {
    const float* w; float* u; float* v;          // W [rows][cols] row-major; persistent buffers weight_u [rows], weight_v [cols]
    float* u_out; float* v_out; float* sig_out;  // copies of the (u, v) used by this forward (for backward) and {sigma, 1/sigma}
    float* part;                                 // scratch: [row_blocks][cols] partial W^T u, then [rows] W v behind it
    int rows, cols; float eps; int pad;
};
#define SN_RB 32          // rows per workgroup in the two mat-vec phases

__device__ __forceinline__ float block_sum_256(float x, float* red) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// phase 1: part[rb][c] = sum_{r in row block rb} W[r][c] * u[r]            grid (max_row_blocks, layers)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const SnDesc* __restrict__ table) {
    const SnDesc d = table[blockIdx.y];
    const int r0 = blockIdx.x * SN_RB;
    if (r0 >= d.rows) return;
    const int r1 = min(d.rows, r0 + SN_RB), C = d.cols;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a0 = 0.f, a1 = 0.f;
        int r = r0;
        for (; r + 2 <= r1; r += 2) { a0 = fmaf(d.w[(size_t)r * C + c], d.u[r], a0); a1 = fmaf(d.w[(size_t)(r + 1) * C + c], d.u[r + 1], a1); }
        if (r < r1) a0 = fmaf(d.w[(size_t)r * C + c], d.u[r], a0);
        d.part[(size_t)blockIdx.x * C + c] = a0 + a1;
    }
}

// phase 2: v = normalize(sum_rb part[rb])  (do_iter) | v unchanged; v_out = v                 grid (layers)
__global__ __launch_bounds__(256) void sn_v_kernel(const SnDesc* __restrict__ table, int do_iter) {
    __shared__ float red[4];
    const SnDesc d = table[blockIdx.x];
    const int C = d.cols, nrb = (d.rows + SN_RB - 1) / SN_RB;
    if (!do_iter) { for (int c = threadIdx.x; c < C; c += 256) d.v_out[c] = d.v[c]; return; }
    float sq = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int k = 0; k < nrb; ++k) t += d.part[(size_t)k * C + c];
        d.v_out[c] = t; sq += t * t;                         // v_out holds the unnormalised vector for a moment
    }
    const float inv = 1.f / fmaxf(sqrtf(block_sum_256(sq, red)), d.eps);
    for (int c = threadIdx.x; c < C; c += 256) { float t = d.v_out[c] * inv; d.v_out[c] = t; d.v[c] = t; }
}

// phase 3: s[r] = W[r] . v_out  for the rows of one block (one wave per row)                  grid (max_row_blocks, layers)
__global__ __launch_bounds__(256) void sn_wv_kernel(const SnDesc* __restrict__ table) {
    const SnDesc d = table[blockIdx.y];
    const int r0 = blockIdx.x * SN_RB;
    if (r0 >= d.rows) return;
    const int r1 = min(d.rows, r0 + SN_RB), C = d.cols, nrb = (d.rows + SN_RB - 1) / SN_RB;
    float* s = d.part + (size_t)nrb * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = r0 + wave; r < r1; r += 4) {
        const float* wr = d.w + (size_t)r * C;
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(wr[c], d.v_out[c], a);
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0) s[r] = a;
    }
}

// phase 4: u = normalize(s) (do_iter), sigma = u . s, outputs                                  grid (layers)
__global__ __launch_bounds__(256) void sn_u_kernel(const SnDesc* __restrict__ table, int do_iter) {
    __shared__ float red[4];
    const SnDesc d = table[blockIdx.x];
    const int R = d.rows, nrb = (d.rows + SN_RB - 1) / SN_RB;
    const float* s = d.part + (size_t)nrb * d.cols;
    float part = 0.f;
    if (do_iter) {
        for (int r = threadIdx.x; r < R; r += 256) part += s[r] * s[r];
        const float inv = 1.f / fmaxf(sqrtf(block_sum_256(part, red)), d.eps);
        part = 0.f;
        for (int r = threadIdx.x; r < R; r += 256) { float un = s[r] * inv; d.u[r] = un; d.u_out[r] = un; part += un * s[r]; }
    } else {
        for (int r = threadIdx.x; r < R; r += 256) { float un = d.u[r]; d.u_out[r] = un; part += un * s[r]; }
    }
    const float sigma = block_sum_256(part, red);
    if (threadIdx.x == 0) { d.sig_out[0] = sigma; d.sig_out[1] = 1.f / sigma; }
}

extern "C" int lp_sn_desc_bytes(void) { return (int)sizeof(SnDesc); }
extern "C" int lp_sn_row_block(void) { return SN_RB; }

extern "C" int lp_sn_power_iter(const void* table, int num_layers, int do_iter, int max_rows, int max_cols, void* stream) {
    if (!table || num_layers <= 0) return lp_set_error(LP_ERR_ARG, "lp_sn_power_iter: bad arguments");
    (void)max_cols;
    hipStream_t st = (hipStream_t)stream;
    const SnDesc* t = (const SnDesc*)table;
    const int nrb = (max_rows + SN_RB - 1) / SN_RB;
    if (do_iter) hipLaunchKernelGGL(sn_wtu_kernel, dim3(nrb, num_layers), dim3(256), 0, st, t);
    hipLaunchKernelGGL(sn_v_kernel, dim3(num_layers), dim3(256), 0, st, t, do_iter);
    hipLaunchKernelGGL(sn_wv_kernel, dim3(nrb, num_layers), dim3(256), 0, st, t);
    hipLaunchKernelGGL(sn_u_kernel, dim3(num_layers), dim3(256), 0, st, t, do_iter);
    return lp_check_launch("sn_power_iter");
}

// <g, w> with both operands contiguous (coalesced), one atomic per block
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dot,
                                                     long long total) {
    __shared__ float red[4];
    float a = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) a = fmaf(g[i], w[i], a);
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dot, a);
}

// dW_orig = alpha * G - (<G, W_orig> * alpha^2) * u v^T, in place on G  (legacy-hook autograd: u, v constants).
// <G, W_orig> is computed here first (sn_dot_kernel, scratch scalar `dot`).
__global__ void sn_grad_apply_kernel(float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v,
                                     const float* __restrict__ sig, const float* __restrict__ dot, int R, int C) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)R * C) return;
    const float alpha = sig[1];
    const float k = dot[0] * alpha * alpha;
    int r = (int)(i / C), c = (int)(i % C);
    g[i] = fmaf(alpha, g[i], -k * u[r] * v[c]);
}

extern "C" int lp_sn_grad_apply(float* g, const float* w_orig, const float* u, const float* v, const float* sig, float* dot, int rows,
                                int cols, void* stream) {
    if (!g || !w_orig || !u || !v || !sig || !dot) return lp_set_error(LP_ERR_ARG, "lp_sn_grad_apply: null pointer");
    long long total = (long long)rows * cols;
    if (hipMemsetAsync(dot, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipMemsetAsync failed");
    int db = (int)((total + 1023) / 1024); if (db > 512) db = 512; if (db < 1) db = 1;
    hipLaunchKernelGGL(sn_dot_kernel, dim3(db), dim3(256), 0, (hipStream_t)stream, g, w_orig, dot, total);
    hipLaunchKernelGGL(sn_grad_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, u, v, sig, dot,
                       rows, cols);
    return lp_check_launch("sn_grad_apply");
}
