// Batched spectral normalisation for gfx950.  The reference wraps every conv/linear in the legacy
// torch.nn.utils.spectral_norm hook: per layer and per forward  v <- normalize(W^T u), u <- normalize(W v) (train mode, in place,
// no grad), sigma = u^T W v, W_eff = W / sigma -- about a dozen tiny ATen launches per layer (~1 200 per training step).
// Here ONE launch handles all layers of a module: blockIdx.x = layer (descriptor table in device memory), 1024 threads stream
// the layer's matrix twice.  sigma == |W v| after the u update, so no third mat-vec is needed.  1/sigma is handed to the conv
// kernels as their epilogue scale `alpha`, so W/sigma is never materialised.
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
// HBM-bound: each lane owns 4 consecutive columns (16-B loads) and keeps 4 rows in flight.
__global__ __launch_bounds__(256) void sn_wtu_kernel(const SnDesc* __restrict__ table) {
    const SnDesc d = table[blockIdx.y];
    const int r0 = blockIdx.x * SN_RB;
    if (r0 >= d.rows) return;
    const int r1 = min(d.rows, r0 + SN_RB), C = d.cols;
    __shared__ float us[SN_RB];
    if (threadIdx.x < SN_RB) us[threadIdx.x] = (r0 + (int)threadIdx.x < r1) ? d.u[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    if ((C & 3) == 0 && (((size_t)d.w | (size_t)d.part) & 15) == 0) {
        const int C4 = C >> 2;
        const float4* w4 = reinterpret_cast<const float4*>(d.w);
        float4* out = reinterpret_cast<float4*>(d.part + (size_t)blockIdx.x * C);
        for (int c = threadIdx.x; c < C4; c += 256) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            int r = r0;
            for (; r + 4 <= r1; r += 4) {
                const float4 x0 = w4[(size_t)r * C4 + c], x1 = w4[(size_t)(r + 1) * C4 + c];
                const float4 x2 = w4[(size_t)(r + 2) * C4 + c], x3 = w4[(size_t)(r + 3) * C4 + c];
                const float u0 = us[r - r0], u1 = us[r - r0 + 1], u2 = us[r - r0 + 2], u3 = us[r - r0 + 3];
                a.x = fmaf(x0.x, u0, a.x); a.y = fmaf(x0.y, u0, a.y); a.z = fmaf(x0.z, u0, a.z); a.w = fmaf(x0.w, u0, a.w);
                b.x = fmaf(x1.x, u1, b.x); b.y = fmaf(x1.y, u1, b.y); b.z = fmaf(x1.z, u1, b.z); b.w = fmaf(x1.w, u1, b.w);
                a.x = fmaf(x2.x, u2, a.x); a.y = fmaf(x2.y, u2, a.y); a.z = fmaf(x2.z, u2, a.z); a.w = fmaf(x2.w, u2, a.w);
                b.x = fmaf(x3.x, u3, b.x); b.y = fmaf(x3.y, u3, b.y); b.z = fmaf(x3.z, u3, b.z); b.w = fmaf(x3.w, u3, b.w);
            }
            for (; r < r1; ++r) {
                const float4 x0 = w4[(size_t)r * C4 + c]; const float u0 = us[r - r0];
                a.x = fmaf(x0.x, u0, a.x); a.y = fmaf(x0.y, u0, a.y); a.z = fmaf(x0.z, u0, a.z); a.w = fmaf(x0.w, u0, a.w);
            }
            out[c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        float a0 = 0.f, a1 = 0.f;
        int r = r0;
        for (; r + 2 <= r1; r += 2) { a0 = fmaf(d.w[(size_t)r * C + c], us[r - r0], a0); a1 = fmaf(d.w[(size_t)(r + 1) * C + c], us[r - r0 + 1], a1); }
        if (r < r1) a0 = fmaf(d.w[(size_t)r * C + c], us[r - r0], a0);
        d.part[(size_t)blockIdx.x * C + c] = a0 + a1;
    }
}

// phase 2a: v_out = sum_rb part[rb] (unnormalised), one squared-norm partial per 64-column block, stored behind the layer's
// scratch (part + nrb*cols + rows).  A block = 64 columns x 4 slices of the row-block partials (the 13 056-row projector
// has 408 of them), 4 loads in flight per thread, combined through LDS.                                   grid (col blocks, layers)
#define SN_VCOLS 64
__global__ __launch_bounds__(256) void sn_vsum_kernel(const SnDesc* __restrict__ table) {
    __shared__ float comb[4][SN_VCOLS];
    __shared__ float red[4];
    const SnDesc d = table[blockIdx.y];
    const int C = d.cols, nrb = (d.rows + SN_RB - 1) / SN_RB;
    if (blockIdx.x * SN_VCOLS >= C) return;
    const int cl = threadIdx.x & (SN_VCOLS - 1), sl = threadIdx.x >> 6;
    const int c = blockIdx.x * SN_VCOLS + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        int k = sl;
        for (; k + 12 < nrb; k += 16) {
            a0 += d.part[(size_t)k * C + c]; a1 += d.part[(size_t)(k + 4) * C + c];
            a2 += d.part[(size_t)(k + 8) * C + c]; a3 += d.part[(size_t)(k + 12) * C + c];
        }
        for (; k < nrb; k += 4) a0 += d.part[(size_t)k * C + c];
    }
    comb[sl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    float t = 0.f;
    if (sl == 0 && c < C) {
        t = (comb[0][cl] + comb[1][cl]) + (comb[2][cl] + comb[3][cl]);
        d.v_out[c] = t;
    }
    const float s = block_sum_256(t * t, red);
    if (threadIdx.x == 0) d.part[(size_t)nrb * C + d.rows + blockIdx.x] = s;
}

// phase 2b: v = v_out = v_out / max(|v_out|, eps)  (do_iter) | v unchanged; v_out = v                 grid (layers)
__global__ __launch_bounds__(256) void sn_v_kernel(const SnDesc* __restrict__ table, int do_iter) {
    const SnDesc d = table[blockIdx.x];
    const int C = d.cols, nrb = (d.rows + SN_RB - 1) / SN_RB;
    if (!do_iter) { for (int c = threadIdx.x; c < C; c += 256) d.v_out[c] = d.v[c]; return; }
    float sq = 0.f;
    const int ncb = (C + SN_VCOLS - 1) / SN_VCOLS;
    for (int j = 0; j < ncb; ++j) sq += d.part[(size_t)nrb * C + d.rows + j];            // (same order in every thread: deterministic)
    const float inv = 1.f / fmaxf(sqrtf(sq), d.eps);
    for (int c = threadIdx.x; c < C; c += 256) { float t = d.v_out[c] * inv; d.v_out[c] = t; d.v[c] = t; }
}

// phase 3: s[r] = W[r] . v_out  for the rows of one block (one wave per row, 16-B loads, 4 in flight)   grid (max_row_blocks, layers)
__global__ __launch_bounds__(256) void sn_wv_kernel(const SnDesc* __restrict__ table) {
    const SnDesc d = table[blockIdx.y];
    const int r0 = blockIdx.x * SN_RB;
    if (r0 >= d.rows) return;
    const int r1 = min(d.rows, r0 + SN_RB), C = d.cols, nrb = (d.rows + SN_RB - 1) / SN_RB;
    float* s = d.part + (size_t)nrb * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool vec = (C & 3) == 0 && (((size_t)d.w | (size_t)d.v_out) & 15) == 0;
    const int C4 = C >> 2;
    const float4* v4 = reinterpret_cast<const float4*>(d.v_out);
    for (int r = r0 + wave; r < r1; r += 4) {
        const float* wr = d.w + (size_t)r * C;
        float a = 0.f;
        if (vec) {
            const float4* w4 = reinterpret_cast<const float4*>(wr);
            float a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int c = lane;
            for (; c + 192 < C4; c += 256) {
                const float4 x0 = w4[c], x1 = w4[c + 64], x2 = w4[c + 128], x3 = w4[c + 192];
                const float4 y0 = v4[c], y1 = v4[c + 64], y2 = v4[c + 128], y3 = v4[c + 192];
                a  = fmaf(x0.x, y0.x, fmaf(x0.y, y0.y, fmaf(x0.z, y0.z, fmaf(x0.w, y0.w, a))));
                a1 = fmaf(x1.x, y1.x, fmaf(x1.y, y1.y, fmaf(x1.z, y1.z, fmaf(x1.w, y1.w, a1))));
                a2 = fmaf(x2.x, y2.x, fmaf(x2.y, y2.y, fmaf(x2.z, y2.z, fmaf(x2.w, y2.w, a2))));
                a3 = fmaf(x3.x, y3.x, fmaf(x3.y, y3.y, fmaf(x3.z, y3.z, fmaf(x3.w, y3.w, a3))));
            }
            for (; c < C4; c += 64) {
                const float4 x0 = w4[c], y0 = v4[c];
                a = fmaf(x0.x, y0.x, fmaf(x0.y, y0.y, fmaf(x0.z, y0.z, fmaf(x0.w, y0.w, a))));
            }
            a = (a + a1) + (a2 + a3);
        } else {
            for (int c = lane; c < C; c += 64) a = fmaf(wr[c], d.v_out[c], a);
        }
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
    hipStream_t st = (hipStream_t)stream;
    const SnDesc* t = (const SnDesc*)table;
    const int nrb = (max_rows + SN_RB - 1) / SN_RB;
    if (do_iter) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(nrb, num_layers), dim3(256), 0, st, t);
        hipLaunchKernelGGL(sn_vsum_kernel, dim3((max_cols + SN_VCOLS - 1) / SN_VCOLS, num_layers), dim3(256), 0, st, t);
    }
    hipLaunchKernelGGL(sn_v_kernel, dim3(num_layers), dim3(256), 0, st, t, do_iter);
    hipLaunchKernelGGL(sn_wv_kernel, dim3(nrb, num_layers), dim3(256), 0, st, t);
    hipLaunchKernelGGL(sn_u_kernel, dim3(num_layers), dim3(256), 0, st, t, do_iter);
    return lp_check_launch("sn_power_iter");
}

// <g, w> with both operands contiguous (coalesced): one partial sum per block into dot[blockIdx.x] (<= SN_DOT_BLOCKS blocks;
// deterministic, and no memset of an accumulator is needed)
#define SN_DOT_BLOCKS 512
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dot,
                                                     long long total) {
    __shared__ float red[4];
    float a = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) a = fmaf(g[i], w[i], a);
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) dot[blockIdx.x] = a;
}

// dW_orig = alpha * G - (<G, W_orig> * alpha^2) * u v^T  (legacy-hook autograd: u, v constants), written in place
// on G, or -- `accum` given -- added to accum (the parameter's .grad: fused gradient accumulation, G untouched).
// <G, W_orig> is computed first (sn_dot_kernel, scratch scalar `dot`).
__global__ __launch_bounds__(256) void sn_grad_apply_kernel(float* __restrict__ g, const float* __restrict__ u,
                                                            const float* __restrict__ v, const float* __restrict__ sig,
                                                            const float* __restrict__ dot, int ndot, float* __restrict__ accum, int R,
                                                            int C) {
    __shared__ float red[4];
    float d = 0.f;
    for (int j = threadIdx.x; j < ndot; j += 256) d += dot[j];          // every block re-sums the (L2-resident) partials
    d = block_sum_256(d, red);
    const float alpha = sig[1];
    const float k = d * alpha * alpha;
    const long long total = (long long)R * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                        // 4 elements per thread, 256 apart: 1 KiB runs per block
        const long long i = (long long)blockIdx.x * 1024 + j * 256 + threadIdx.x;
        if (i < total) {
            const int r = (int)(i / C), c = (int)(i % C);
            const float val = fmaf(alpha, g[i], -k * u[r] * v[c]);
            if (accum) accum[i] += val; else g[i] = val;
        }
    }
}

extern "C" int lp_sn_grad_apply(float* g, const float* w_orig, const float* u, const float* v, const float* sig, float* dot, int ndot,
                                float* accum, int rows, int cols, void* stream) {
    if (!g || !w_orig || !u || !v || !sig || !dot) return lp_set_error(LP_ERR_ARG, "lp_sn_grad_apply: null pointer");
    long long total = (long long)rows * cols;
    int db = ndot;
    if (ndot <= 0) {         // <g, w_orig> not supplied by the producer of g (lp_conv16_wgrad's sn_dot): take it here
        db = (int)((total + 1023) / 1024); if (db > SN_DOT_BLOCKS) db = SN_DOT_BLOCKS; if (db < 1) db = 1;
        hipLaunchKernelGGL(sn_dot_kernel, dim3(db), dim3(256), 0, (hipStream_t)stream, g, w_orig, dot, total);
    }
    hipLaunchKernelGGL(sn_grad_apply_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, g, u, v, sig, dot,
                       db, accum, rows, cols);
    return lp_check_launch("sn_grad_apply");
}

// MANY lp_sn_grad_apply jobs in one launch (round 5 launch diet: 67 per-layer launches of a meta-training step -> 3).  Every job's <G, W_orig>
// partials must already exist (ndot > 0: written by the weight-gradient reduction that produced G) and every job ACCUMULATES into its own
// `accum` (distinct per job: two jobs on one target would race).  The descriptors travel BY VALUE in the kernel arguments (captured by value in
// a hipGraph: no device table, no host-to-device copy), SN_BATCH_MAX per launch; `host_descs` is read on the host at call time only.
struct SnApplyDesc { float* g; const float* u; const float* v; const float* sig; const float* dot; float* accum; int ndot, rows, cols, block0; };
#define SN_BATCH_MAX 40
struct SnApplyBatch { int n; int pad; SnApplyDesc d[SN_BATCH_MAX]; };

__global__ __launch_bounds__(256) void sn_grad_apply_batch_kernel(SnApplyBatch b) {
    __shared__ float red[4];
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.d[j + 1].block0) ++j;          // (block0 ascending; <= 40 entries, scalar loads from the kernarg segment)
    const SnApplyDesc q = b.d[j];
    float d = 0.f;
    for (int t = threadIdx.x; t < q.ndot; t += 256) d += q.dot[t];
    d = block_sum_256(d, red);
    const float alpha = q.sig[1];
    const float k = d * alpha * alpha;
    const long long total = (long long)q.rows * q.cols;
    const long long base = (long long)((int)blockIdx.x - q.block0) * 1024;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long i = base + e * 256 + threadIdx.x;
        if (i < total) {
            const int r = (int)(i / q.cols), c = (int)(i % q.cols);
            q.accum[i] += fmaf(alpha, q.g[i], -k * q.u[r] * q.v[c]);
        }
    }
}

extern "C" int lp_sn_apply_desc_bytes(void) { return (int)sizeof(SnApplyDesc); }

extern "C" int lp_sn_grad_apply_batch(const void* host_descs, int count, void* stream) {
    if (count < 0 || (count > 0 && !host_descs)) return lp_set_error(LP_ERR_ARG, "lp_sn_grad_apply_batch: bad arguments");
    const SnApplyDesc* src = (const SnApplyDesc*)host_descs;
    for (int i0 = 0; i0 < count; i0 += SN_BATCH_MAX) {
        SnApplyBatch b;
        b.n = count - i0 < SN_BATCH_MAX ? count - i0 : SN_BATCH_MAX; b.pad = 0;
        long long blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            b.d[i] = src[i0 + i];
            if (!b.d[i].g || !b.d[i].u || !b.d[i].v || !b.d[i].sig || !b.d[i].dot || !b.d[i].accum || b.d[i].ndot < 1)
                return lp_set_error(LP_ERR_ARG, "lp_sn_grad_apply_batch: every job needs g, u, v, sig, dot partials (ndot >= 1) and an accumulation target");
            b.d[i].block0 = (int)blocks;
            blocks += ((long long)b.d[i].rows * b.d[i].cols + 1023) / 1024;
        }
        if (blocks > 0x7fffffffll) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_sn_grad_apply_batch: too many elements in one batch");
        if (blocks > 0) hipLaunchKernelGGL(sn_grad_apply_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
    }
    return lp_check_launch("sn_grad_apply_batch");
}

// Gradient of the spectrally normalised label embedding (discriminators/no_landmarks.py:84-86,152; nn.SNEmbeddingFn): the dense part of
//   dW_orig = scatter(rows at label) - coef * u v^T        (coef = <G, W_orig> / sigma^2, device scalar)
// ADDED to grad [N][E]: one read-modify-write pass over the 98000 x 512 matrix for the rank-1 term (what torch.addmm_ did through
// rocBLAS), then the B gradient rows in ONE workgroup that walks the batch in order (duplicate labels accumulate deterministically).
__global__ __launch_bounds__(256) void sn_embed_rank1_kernel(float* __restrict__ grad, const float* __restrict__ u, const float* __restrict__ v,
                                                             const float* __restrict__ coef, long long total4, int E4) {
    const float c = -coef[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long row = i / E4;
        const int c4 = (int)(i - row * E4);
        const float s = c * u[row];
        const float4 vv = ((const float4*)v)[c4];
        float4 g = ((float4*)grad)[i];
        g.x = fmaf(s, vv.x, g.x); g.y = fmaf(s, vv.y, g.y); g.z = fmaf(s, vv.z, g.z); g.w = fmaf(s, vv.w, g.w);
        ((float4*)grad)[i] = g;
    }
}

__global__ __launch_bounds__(256) void sn_embed_rows_kernel(float* __restrict__ grad, const long long* __restrict__ label, const float* __restrict__ rows,
                                                            int N, int E, int B) {
    for (int b = 0; b < B; ++b) {
        const long long r = label[b];
        if (r < 0 || r >= N) continue;
        for (int j = threadIdx.x; j < E; j += 256) grad[r * E + j] += rows[(long long)b * E + j];      // (a column stays with its thread)
    }
}

extern "C" int lp_sn_embed_grad(float* grad, const float* u, const float* v, const float* coef, const long long* label, const float* rows,
                                int N, int E, int B, void* stream) {
    if (!grad || !u || !v || !coef || !label || !rows) return lp_set_error(LP_ERR_ARG, "lp_sn_embed_grad: null pointer");
    if (E & 3) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_sn_embed_grad: embedding width must be a multiple of 4");
    const long long total4 = (long long)N * (E / 4);
    long long blocks = (total4 + 1023) / 1024; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sn_embed_rank1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, u, v, coef, total4, E / 4);
    hipLaunchKernelGGL(sn_embed_rows_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, grad, label, rows, N, E, B);
    return lp_check_launch("sn_embed_grad");
}

#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif
