// Reflection padding of the 3x3 ResBlock convs (reference: generators/common/blocks.py:76-88 `padding(1)` = nn.ReflectionPad2d(1) in front of
// a conv with padding 0, selected by --gen_padding / --dis_padding 'reflection': noBottleneck.py:53-58, no_landmarks.py:45-50) as a BORDER
// CORRECTION of the zero-padded conv, by linearity:
//
//     conv(reflect_pad(x)) = conv_zero_pad(x) + B(x),      B(x)[p] = sum over the taps t that fall OUTSIDE the image at output pixel p of
//                                                                    W[:, :, t] . x[mirror(p + t)]        (mirror(-1) = 1, mirror(H) = H - 2)
//
// B touches the 2W + 2(H - 2) border pixels only (3 taps each, 5 at the corners): 0.3 ... 6 % of the conv's work, so the LDS-DMA / MFMA kernels
// keep their zero-padding loaders untouched and three small fp32 kernels add B, B^T (data gradient: a gather per pixel of the ring one pixel
// inside the border) and the border's share of the weight gradient.  Operands: x from the 16-bit operand planes the conv itself consumed
// (optionally at half resolution: the nearest-neighbour x2 upsample of an up block is applied by index), W_orig in fp32 with the spectral
// norm's 1/sigma as `alpha`, dy in fp32.  Deterministic (no atomics).  The option is not used by the shipped configs: these kernels are plain
// VALU code, sized for correctness first.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

#define RB_NB 8          // images per pass of the accumulators
#define RB_T 5           // at most 5 taps leave the image at one output pixel (a corner)
#define RB_E 8           // at most 7 (tap, output pixel) pairs read one ring pixel through the mirror (a ring corner)

template <int PREC> __device__ __forceinline__ float rb_plane(const uint16_t* __restrict__ hi, const uint16_t* __restrict__ lo, size_t i) {
    if (PREC == LP_PREC_F16) return lp_op16_to_f32<true>(hi[i]);
    float v = lp_op16_to_f32<false>(hi[i]);
    if (PREC == LP_PREC_BF16X3) v += lp_op16_to_f32<false>(lo[i]);
    return v;
}
__device__ __forceinline__ int rb_mirror(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
// b-th border pixel of an H x W map: top row, bottom row, then the left / right pixel of the rows in between
__device__ __forceinline__ void rb_border_pixel(int b, int H, int W, int& oy, int& ox) {
    if (b < W) { oy = 0; ox = b; }
    else if (b < 2 * W) { oy = H - 1; ox = b - W; }
    else { const int k = b - 2 * W; oy = 1 + (k >> 1); ox = (k & 1) ? W - 1 : 0; }
}
// r-th pixel of the ring one pixel inside the border (rows 1 and H-2 in full, columns 1 and W-2 of every other row): the mirror sources
__device__ __forceinline__ void rb_ring_pixel(int r, int H, int W, int& sy, int& sx) {
    if (r < W) { sy = 1; sx = r; return; }
    if (H > 3 && r < 2 * W) { sy = H - 2; sx = r - W; return; }
    const int k = r - (H > 3 ? 2 : 1) * W, j = k >> 1;      // rows other than 1 and H-2, in ascending order
    int row = 0, seen = 0;
    for (int y = 0; y < H; ++y) { if (y == 1 || y == H - 2) continue; if (seen == j) { row = y; break; } ++seen; }
    sy = row; sx = (k & 1) ? W - 2 : 1;
}

struct RbParams {
    const uint16_t* x_hi; const uint16_t* x_lo;      // operand planes [N][H >> up][W >> up][C8]
    const float* w;                                  // W_orig [Cout][Cin][3][3]
    const float* alpha;                              // device scalar 1/sigma | NULL
    const float* dy; float* out;                     // fwd: out = y [N][H][W][Cout] (+=); dgrad: dy [N][H][W][Cout], out = dx [N][H][W][Cin] (+=); wgrad: out = gw
    const uint16_t* mask_hi; int mask_c8;            // dgrad: planes of relu(x) -> the correction is multiplied by [x > 0] like the main launch's epilogue
    int N, H, W, Cin, Cout, C8, up;
};

// ---- forward: y[n][p][co] += alpha * sum_{t outside at p} sum_ci W[co][ci][t] * x[n][mirror(p + t)][ci]            grid (border pixels, Cout / 64)
template <int PREC>
__global__ __launch_bounds__(256) void reflect_fwd_kernel(RbParams p) {
    __shared__ float xs[RB_T][RB_NB][64];
    __shared__ float red[4][RB_NB][64];
    int oy, ox;
    rb_border_pixel(blockIdx.x, p.H, p.W, oy, ox);
    int tap[RB_T], sy[RB_T], sx[RB_T], nt = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) { tap[nt] = ky * 3 + kx; sy[nt] = rb_mirror(iy, p.H) >> p.up; sx[nt] = rb_mirror(ix, p.W) >> p.up; ++nt; }
        }
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6, co = blockIdx.y * 64 + cl;
    const int Hs = p.H >> p.up, Ws = p.W >> p.up;
    const float a = p.alpha ? p.alpha[0] : 1.f;
    for (int n0 = 0; n0 < p.N; n0 += RB_NB) {
        const int nn = min(RB_NB, p.N - n0);
        float acc[RB_NB];
#pragma unroll
        for (int i = 0; i < RB_NB; ++i) acc[i] = 0.f;
        for (int ci0 = 0; ci0 < p.Cin; ci0 += 64) {
            __syncthreads();
            for (int i = threadIdx.x; i < nt * RB_NB * 64; i += 256) {
                const int t = i / (RB_NB * 64), n = (i >> 6) % RB_NB, c = i & 63;
                float v = 0.f;
                if (n < nn && ci0 + c < p.Cin) v = rb_plane<PREC>(p.x_hi, p.x_lo, (((size_t)(n0 + n) * Hs + sy[t]) * Ws + sx[t]) * p.C8 + ci0 + c);
                xs[t][n][c] = v;
            }
            __syncthreads();
            if (co < p.Cout) {
                const int c1 = min(16, p.Cin - ci0 - sl * 16);
                for (int t = 0; t < nt; ++t)
                    for (int c = 0; c < c1; ++c) {
                        const float wv = p.w[((size_t)co * p.Cin + ci0 + sl * 16 + c) * 9 + tap[t]];
#pragma unroll
                        for (int n = 0; n < RB_NB; ++n) acc[n] = fmaf(wv, xs[t][n][sl * 16 + c], acc[n]);
                    }
            }
        }
#pragma unroll
        for (int n = 0; n < RB_NB; ++n) red[sl][n][cl] = acc[n];
        __syncthreads();
        if (sl == 0 && co < p.Cout)
            for (int n = 0; n < nn; ++n) {
                const float s = (red[0][n][cl] + red[1][n][cl]) + (red[2][n][cl] + red[3][n][cl]);
                p.out[(((size_t)(n0 + n) * p.H + oy) * p.W + ox) * p.Cout + co] += a * s;
            }
        __syncthreads();
    }
}

// ---- data gradient: dx[n][s][ci] += alpha * sum over the (tap t, output pixel q) pairs with mirror(q + t) = s, q + t outside, of
//                                     sum_co W[co][ci][t] * dy[n][q][co]                                          grid (ring pixels, Cin / 64)
__global__ __launch_bounds__(256) void reflect_dgrad_kernel(RbParams p) {
    __shared__ float ds[RB_E][RB_NB][64];
    __shared__ float red[4][RB_NB][64];
    int sy, sx;
    rb_ring_pixel(blockIdx.x, p.H, p.W, sy, sx);
    int tap[RB_E], qy[RB_E], qx[RB_E], ne = 0;
    // pre-images of the ring pixel under the mirror: itself and, for a coordinate of 1 (H-2), the outside coordinate -1 (H)
    for (int a = 0; a < 3; ++a) {
        const int iy = a == 0 ? sy : (a == 1 ? (sy == 1 ? -1 : -9) : (sy == p.H - 2 ? p.H : -9));
        if (iy == -9) continue;
        for (int b = 0; b < 3; ++b) {
            const int ix = b == 0 ? sx : (b == 1 ? (sx == 1 ? -1 : -9) : (sx == p.W - 2 ? p.W : -9));
            if (ix == -9 || (a == 0 && b == 0)) continue;          // (both inside: the zero-padded conv's own term)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int oy = iy - ky + 1, ox = ix - kx + 1;
                    if (oy >= 0 && oy < p.H && ox >= 0 && ox < p.W && ne < RB_E) { tap[ne] = ky * 3 + kx; qy[ne] = oy; qx[ne] = ox; ++ne; }
                }
        }
    }
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6, ci = blockIdx.y * 64 + cl;
    const float a = p.alpha ? p.alpha[0] : 1.f;
    for (int n0 = 0; n0 < p.N; n0 += RB_NB) {
        const int nn = min(RB_NB, p.N - n0);
        float acc[RB_NB];
#pragma unroll
        for (int i = 0; i < RB_NB; ++i) acc[i] = 0.f;
        for (int co0 = 0; co0 < p.Cout; co0 += 64) {
            __syncthreads();
            for (int i = threadIdx.x; i < ne * RB_NB * 64; i += 256) {
                const int e = i / (RB_NB * 64), n = (i >> 6) % RB_NB, c = i & 63;
                ds[e][n][c] = (n < nn && co0 + c < p.Cout) ? p.dy[(((size_t)(n0 + n) * p.H + qy[e]) * p.W + qx[e]) * p.Cout + co0 + c] : 0.f;
            }
            __syncthreads();
            if (ci < p.Cin) {
                const int c1 = min(16, p.Cout - co0 - sl * 16);
                for (int e = 0; e < ne; ++e)
                    for (int c = 0; c < c1; ++c) {
                        const float wv = p.w[((size_t)(co0 + sl * 16 + c) * p.Cin + ci) * 9 + tap[e]];
#pragma unroll
                        for (int n = 0; n < RB_NB; ++n) acc[n] = fmaf(wv, ds[e][n][sl * 16 + c], acc[n]);
                    }
            }
        }
#pragma unroll
        for (int n = 0; n < RB_NB; ++n) red[sl][n][cl] = acc[n];
        __syncthreads();
        if (sl == 0 && ci < p.Cin)
            for (int n = 0; n < nn; ++n) {
                const size_t pix = ((size_t)(n0 + n) * p.H + sy) * p.W + sx;
                float s = a * ((red[0][n][cl] + red[1][n][cl]) + (red[2][n][cl] + red[3][n][cl]));
                if (p.mask_hi && !((short)p.mask_hi[pix * p.mask_c8 + ci] > 0)) s = 0.f;       // planes of relu(x): positive <=> x > 0 (bf16 and fp16 alike)
                p.out[pix * p.Cin + ci] += s;
            }
        __syncthreads();
    }
}

// ---- weight gradient of the border terms: gw[co][ci][t] = sum_n sum_{q: q + t outside} dy[n][q][co] * x[n][mirror(q + t)][ci]   (centre tap: 0)
//                                                                                                   grid (9 taps, Cout / 16, Cin / 64)
template <int PREC>
__global__ __launch_bounds__(256) void reflect_wgrad_kernel(RbParams p) {
    __shared__ float dys[16][16];
    __shared__ float xs[16][64];
    const int t = blockIdx.x, ky = t / 3, kx = t % 3;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, ci = blockIdx.z * 64 + cl, co0 = blockIdx.y * 16;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int Hs = p.H >> p.up, Ws = p.W >> p.up;
    // output pixels where tap t leaves the image: the row oy_r (ky != 1), the column ox_c (kx != 1), their union for a corner tap
    const int oy_r = ky == 0 ? 0 : p.H - 1, ox_c = kx == 0 ? 0 : p.W - 1;
    const int nrow = ky != 1 ? p.W : 0, ncol = kx != 1 ? (ky != 1 ? p.H - 1 : p.H) : 0, Q = nrow + ncol;
    const int items = p.N * Q;
    for (int i0 = 0; i0 < items; i0 += 16) {
        __syncthreads();
        {   // stage 16 (image, pixel) items: dy[.][16 channels of this block], x[.][64 channels of this block]
            const int it = threadIdx.x >> 4, c = threadIdx.x & 15, i = i0 + it;
            float dv = 0.f;
            int n = 0, oy = 0, ox = 0;
            if (i < items) {
                n = i / Q; const int q = i % Q;
                if (q < nrow) { oy = oy_r; ox = q; }
                else { const int j = q - nrow; ox = ox_c; oy = (ky != 1) ? (oy_r == 0 ? j + 1 : j) : j; }
                if (co0 + c < p.Cout) dv = p.dy[(((size_t)n * p.H + oy) * p.W + ox) * p.Cout + co0 + c];
            }
            dys[it][c] = dv;
            for (int cc = c; cc < 64; cc += 16) {
                float xv = 0.f;
                if (i < items && blockIdx.z * 64 + cc < p.Cin) {
                    const int my = rb_mirror(oy + ky - 1, p.H) >> p.up, mx = rb_mirror(ox + kx - 1, p.W) >> p.up;
                    xv = rb_plane<PREC>(p.x_hi, p.x_lo, (((size_t)n * Hs + my) * Ws + mx) * p.C8 + blockIdx.z * 64 + cc);
                }
                xs[it][cc] = xv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const float xv = xs[it][cl];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(dys[it][g * 4 + j], xv, acc[j]);
        }
    }
    if (ci < p.Cin)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (co0 + g * 4 + j < p.Cout) p.out[((size_t)(co0 + g * 4 + j) * p.Cin + ci) * 9 + t] = acc[j];
}

static int rb_check(const char* who, int N, int H, int W, int Cin, int Cout, int upsample) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H < 4 || W < 4 || (upsample && ((H | W) & 1)))
        return lp_set_error(LP_ERR_ARG, who);      // (a 2-pixel map has no pixel to mirror onto: nn.ReflectionPad2d(1) needs >= 2, the ring bookkeeping here >= 4)
    return LP_OK;
}

extern "C" int lp_reflect_border_fwd(const uint16_t* x_hi, const uint16_t* x_lo, int prec, int N, int H, int W, int Cin, int C8, int upsample,
                                     const float* w, int Cout, const float* alpha, float* y, void* stream) {
    if (!x_hi || !w || !y || (prec == LP_PREC_BF16X3 && !x_lo)) return lp_set_error(LP_ERR_ARG, "lp_reflect_border_fwd: null argument");
    if (int e = rb_check("lp_reflect_border_fwd: bad geometry", N, H, W, Cin, Cout, upsample)) return e;
    RbParams p{}; p.x_hi = x_hi; p.x_lo = x_lo; p.w = w; p.alpha = alpha; p.out = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.C8 = C8; p.up = upsample ? 1 : 0;
    const dim3 grid(2 * W + 2 * (H - 2), (Cout + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    if (prec == LP_PREC_F16) hipLaunchKernelGGL(reflect_fwd_kernel<LP_PREC_F16>, grid, dim3(256), 0, st, p);
    else if (prec == LP_PREC_BF16X3) hipLaunchKernelGGL(reflect_fwd_kernel<LP_PREC_BF16X3>, grid, dim3(256), 0, st, p);
    else if (prec == LP_PREC_BF16) hipLaunchKernelGGL(reflect_fwd_kernel<LP_PREC_BF16>, grid, dim3(256), 0, st, p);
    else return lp_set_error(LP_ERR_ARG, "lp_reflect_border_fwd: unknown operand mode");
    return lp_check_launch("lp_reflect_border_fwd");
}

extern "C" int lp_reflect_border_dgrad(const float* dy, int N, int H, int W, int Cout, const float* w, int Cin, const float* alpha,
                                       const uint16_t* mask_hi, int mask_c8, float* dx, void* stream) {
    if (!dy || !w || !dx) return lp_set_error(LP_ERR_ARG, "lp_reflect_border_dgrad: null argument");
    if (int e = rb_check("lp_reflect_border_dgrad: bad geometry", N, H, W, Cin, Cout, 0)) return e;
    RbParams p{}; p.dy = dy; p.w = w; p.alpha = alpha; p.out = dx; p.mask_hi = mask_hi; p.mask_c8 = mask_c8;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    const int rows = H > 3 ? 2 : 1, ring = rows * W + 2 * (H - rows);
    hipLaunchKernelGGL(reflect_dgrad_kernel, dim3(ring, (Cin + 63) / 64), dim3(256), 0, (hipStream_t)stream, p);
    return lp_check_launch("lp_reflect_border_dgrad");
}

extern "C" int lp_reflect_border_wgrad(const uint16_t* x_hi, const uint16_t* x_lo, int prec, int N, int H, int W, int Cin, int C8, int upsample,
                                       const float* dy, int Cout, float* gw, void* stream) {
    if (!x_hi || !dy || !gw || (prec == LP_PREC_BF16X3 && !x_lo)) return lp_set_error(LP_ERR_ARG, "lp_reflect_border_wgrad: null argument");
    if (int e = rb_check("lp_reflect_border_wgrad: bad geometry", N, H, W, Cin, Cout, upsample)) return e;
    RbParams p{}; p.x_hi = x_hi; p.x_lo = x_lo; p.dy = dy; p.out = gw;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.C8 = C8; p.up = upsample ? 1 : 0;
    const dim3 grid(9, (Cout + 15) / 16, (Cin + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    if (prec == LP_PREC_F16) hipLaunchKernelGGL(reflect_wgrad_kernel<LP_PREC_F16>, grid, dim3(256), 0, st, p);
    else if (prec == LP_PREC_BF16X3) hipLaunchKernelGGL(reflect_wgrad_kernel<LP_PREC_BF16X3>, grid, dim3(256), 0, st, p);
    else if (prec == LP_PREC_BF16) hipLaunchKernelGGL(reflect_wgrad_kernel<LP_PREC_BF16>, grid, dim3(256), 0, st, p);
    else return lp_set_error(LP_ERR_ARG, "lp_reflect_border_wgrad: unknown operand mode");
    return lp_check_launch("lp_reflect_border_wgrad");
}
