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

template <int PREC> __device__ __forceinline__ float rb_plane(const uint16_t* __restrict__ hi, const uint16_t* __restrict__ lo, size_t i) {
    if (PREC == LP_PREC_F16) return lp_op16_to_f32<true>(hi[i]);
    float v = lp_op16_to_f32<false>(hi[i]);
    if (PREC == LP_PREC_BF16X3) v += lp_op16_to_f32<false>(lo[i]);
    return v;
}
__device__ __forceinline__ int rb_mirror(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
// b-th border pixel of an H x W map: top row, bottom row, then the left and the right column of the rows in between (consecutive pixels share
// their set of outside taps except at the corners)
__device__ __forceinline__ void rb_border_pixel(int b, int H, int W, int& oy, int& ox) {
    if (b < W) { oy = 0; ox = b; }
    else if (b < 2 * W) { oy = H - 1; ox = b - W; }
    else { const int k = b - 2 * W; if (k < H - 2) { oy = 1 + k; ox = 0; } else { oy = 1 + k - (H - 2); ox = W - 1; } }
}
// r-th pixel of the ring one pixel inside the border (rows 1 and H-2 in full, columns 1 and W-2 of every other row): the mirror sources
__device__ __forceinline__ void rb_ring_pixel(int r, int H, int W, int& sy, int& sx) {
    if (r < W) { sy = 1; sx = r; return; }
    if (r < 2 * W) { sy = H - 2; sx = r - W; return; }
    const int k = r - 2 * W, j = k >> 1;                    // j-th of the rows other than 1 and H-2 (H >= 4): 0, 2, 3, ..., H-3, H-1
    sy = j == 0 ? 0 : (j == H - 3 ? H - 1 : j + 1); sx = (k & 1) ? W - 2 : 1;
}

struct RbParams {
    const uint16_t* x_hi; const uint16_t* x_lo;      // operand planes [N][H >> up][W >> up][C8]
    const float* w;                                  // W_orig [Cout][Cin][3][3]
    const float* alpha;                              // device scalar 1/sigma | NULL
    const float* dy; float* out;                     // fwd: out = y [N][H][W][Cout] (+=); dgrad: dy [N][H][W][Cout], out = dx [N][H][W][Cin] (+=); wgrad: out = gw
    const uint16_t* mask_hi; int mask_c8;            // dgrad: planes of relu(x) -> the correction is multiplied by [x > 0] like the main launch's epilogue
    int N, H, W, Cin, Cout, C8, up;
};

// ---- forward: y[n][p][co] += alpha * sum_{t outside at p} sum_ci W[co][ci][t] * x[n][mirror(p + t)][ci]
// grid (groups of RB_PB consecutive border pixels, Cout / 64).  A block keeps RB_PB pixels x RB_NB images of accumulators per thread (thread =
// output channel, the 4 waves split each 16-channel chunk of Cin), so a weight element is read once per RB_PB * RB_NB products; the weight
// tile of a chunk -- 64 rows of 16 x 9 contiguous floats -- is staged through LDS with coalesced loads.  The eight non-centre taps are walked
// uniformly; a (tap, pixel) pair whose tap stays inside the image contributes a zero operand.
#define RB_PB 4
__device__ __forceinline__ int rb_tap(int t8) { return t8 < 4 ? t8 : t8 + 1; }      // the 8 taps around the centre
template <int PREC>
__global__ __launch_bounds__(256) void reflect_fwd_kernel(RbParams p) {
    __shared__ float wsh[64][16 * 9 + 1];
    __shared__ float xs[8][RB_PB][16 + 1][RB_NB];                      // (+1 row: the staging writes of a wave -- 4 pixels x 8 images at one channel -- hit 32 distinct banks)
    __shared__ int src[8][RB_PB];
    const int P = 2 * p.W + 2 * (p.H - 2), Hs = p.H >> p.up, Ws = p.W >> p.up;
    if (threadIdx.x < 8 * RB_PB) {
        const int t8 = threadIdx.x / RB_PB, j = threadIdx.x % RB_PB, b = blockIdx.x * RB_PB + j, t = rb_tap(t8);
        int v = -1;
        if (b < P) {
            int oy, ox;
            rb_border_pixel(b, p.H, p.W, oy, ox);
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) v = (rb_mirror(iy, p.H) >> p.up) * Ws + (rb_mirror(ix, p.W) >> p.up);
        }
        src[t8][j] = v;
    }
    __syncthreads();
    bool active[8];
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8) { bool any = false; for (int j = 0; j < RB_PB; ++j) any |= src[t8][j] >= 0; active[t8] = any; }
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6, co = blockIdx.y * 64 + cl;
    const float a = p.alpha ? p.alpha[0] : 1.f;
    float* red = &wsh[0][0];                                   // [4 slices][RB_PB * RB_NB][64] after the main loop (8192 <= 64 * 145 floats)
    for (int n0 = 0; n0 < p.N; n0 += RB_NB) {
        const int nn = min(RB_NB, p.N - n0);
        float acc[RB_PB][RB_NB];
#pragma unroll
        for (int j = 0; j < RB_PB; ++j)
#pragma unroll
            for (int n = 0; n < RB_NB; ++n) acc[j][n] = 0.f;
        // software pipeline over the 16-channel chunks of Cin: the global loads of chunk k+1 (weight tile: 9 x 16 B per thread; one operand row of
        // 16 channels per thread) are in flight while chunk k is multiplied out of LDS
        const int xn = threadIdx.x & (RB_NB - 1), xj = (threadIdx.x / RB_NB) % RB_PB, xt8 = threadIdx.x / (RB_NB * RB_PB);
        const int xsrc = src[xt8][xj];
        const bool xlive = active[xt8] && xsrc >= 0 && xn < nn;
        const size_t xrow = ((size_t)(n0 + xn) * Hs * Ws + (xlive ? xsrc : 0)) * p.C8;
        float4 wreg[9];
        float xv[16];
        auto fetch = [&](int ci0) {
            if ((p.Cin & 3) == 0 && ci0 + 16 <= p.Cin) {
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int i = threadIdx.x + u * 256, r = i / 36, k4 = i % 36, cg = blockIdx.y * 64 + r;
                    wreg[u] = cg < p.Cout ? *reinterpret_cast<const float4*>(p.w + ((size_t)cg * p.Cin + ci0) * 9 + k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (xlive && ci0 + 16 <= p.C8) {
                const s16x8_t h0 = *reinterpret_cast<const s16x8_t*>(p.x_hi + xrow + ci0), h1 = *reinterpret_cast<const s16x8_t*>(p.x_hi + xrow + ci0 + 8);
                s16x8_t l0 = h0, l1 = h1;
                if (PREC == LP_PREC_BF16X3) { l0 = *reinterpret_cast<const s16x8_t*>(p.x_lo + xrow + ci0); l1 = *reinterpret_cast<const s16x8_t*>(p.x_lo + xrow + ci0 + 8); }
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const uint16_t hb = (uint16_t)(c < 8 ? h0[c] : h1[c - 8]), lb = (uint16_t)(c < 8 ? l0[c] : l1[c - 8]);
                    float f = lp_op16_to_f32<PREC == LP_PREC_F16>(hb);
                    if (PREC == LP_PREC_BF16X3) f += lp_op16_to_f32<false>(lb);
                    xv[c] = ci0 + c < p.Cin ? f : 0.f;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) xv[c] = (xlive && ci0 + c < p.Cin) ? rb_plane<PREC>(p.x_hi, p.x_lo, xrow + ci0 + c) : 0.f;
            }
        };
        auto commit = [&](int ci0) {
            if ((p.Cin & 3) == 0 && ci0 + 16 <= p.Cin) {
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int i = threadIdx.x + u * 256, r = i / 36, k4 = i % 36;
                    wsh[r][k4 * 4] = wreg[u].x; wsh[r][k4 * 4 + 1] = wreg[u].y; wsh[r][k4 * 4 + 2] = wreg[u].z; wsh[r][k4 * 4 + 3] = wreg[u].w;
                }
            } else {          // ragged channel counts: element-wise, not prefetched
#pragma unroll 12
                for (int i = threadIdx.x; i < 64 * 144; i += 256) {
                    const int r = i / 144, k = i % 144, cg = blockIdx.y * 64 + r;
                    wsh[r][k] = (cg < p.Cout && ci0 + k / 9 < p.Cin) ? p.w[((size_t)cg * p.Cin + ci0) * 9 + k] : 0.f;
                }
            }
            if (active[xt8]) {
#pragma unroll
                for (int c = 0; c < 16; ++c) xs[xt8][xj][c][xn] = xv[c];
            }
        };
        fetch(0);
        for (int ci0 = 0; ci0 < p.Cin; ci0 += 16) {
            __syncthreads();
            commit(ci0);
            __syncthreads();
            if (ci0 + 16 < p.Cin) fetch(ci0 + 16);
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                if (!active[t8]) continue;
                const int t = rb_tap(t8);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float wv = wsh[cl][(sl * 4 + c) * 9 + t];
#pragma unroll
                    for (int j = 0; j < RB_PB; ++j) {
                        const float4 x0 = *reinterpret_cast<const float4*>(&xs[t8][j][sl * 4 + c][0]);
                        const float4 x1 = *reinterpret_cast<const float4*>(&xs[t8][j][sl * 4 + c][4]);
                        acc[j][0] = fmaf(wv, x0.x, acc[j][0]); acc[j][1] = fmaf(wv, x0.y, acc[j][1]); acc[j][2] = fmaf(wv, x0.z, acc[j][2]); acc[j][3] = fmaf(wv, x0.w, acc[j][3]);
                        acc[j][4] = fmaf(wv, x1.x, acc[j][4]); acc[j][5] = fmaf(wv, x1.y, acc[j][5]); acc[j][6] = fmaf(wv, x1.z, acc[j][6]); acc[j][7] = fmaf(wv, x1.w, acc[j][7]);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RB_PB; ++j)
#pragma unroll
            for (int n = 0; n < RB_NB; ++n) red[((size_t)sl * RB_PB * RB_NB + j * RB_NB + n) * 64 + cl] = acc[j][n];
        __syncthreads();
        {   // wave j adds pixel j of the group: the RB_NB old values are loaded together, then stored (one memory latency, not RB_NB)
            static_assert(RB_PB == 4, "one wave per pixel of the group");
            const int b = blockIdx.x * RB_PB + sl;
            if (b < P && co < p.Cout) {
                int oy, ox;
                rb_border_pixel(b, p.H, p.W, oy, ox);
                float* dst = p.out + ((size_t)n0 * p.H * p.W + (size_t)oy * p.W + ox) * p.Cout + co;
                const size_t img = (size_t)p.H * p.W * p.Cout;
                float old[RB_NB];
#pragma unroll
                for (int n = 0; n < RB_NB; ++n) old[n] = n < nn ? dst[n * img] : 0.f;
#pragma unroll
                for (int n = 0; n < RB_NB; ++n) {
                    const int e = (sl * RB_NB + n) * 64 + cl, q = RB_PB * RB_NB * 64;
                    if (n < nn) dst[n * img] = old[n] + a * ((red[e] + red[q + e]) + (red[2 * q + e] + red[3 * q + e]));
                }
            }
        }
        __syncthreads();
    }
}

// ---- data gradient: dx[n][s][ci] += alpha * sum_t sum_co W[co][ci][t] * ( sum over the output pixels q with mirror(q + t) = s, q + t outside, of dy[n][q][co] )
// grid (groups of RB_PB consecutive ring pixels, Cin / 64).  The same structure with the roles of the channels swapped: thread = input channel, the
// weight tile of a 16-output-channel chunk is 16 rows of 64 x 9 contiguous floats; per (tap, ring pixel) the up to three output pixels that reach
// the ring pixel through the mirror (row folded, column folded, both) are summed while dy is staged.  A gather: no atomics, deterministic.
__global__ __launch_bounds__(256) void reflect_dgrad_kernel(RbParams p) {
    __shared__ float wsh[16][64 * 9];
    __shared__ float ds[8][RB_PB][16 + 1][RB_NB];                      // (+1 row: conflict-free staging writes, as xs in the forward kernel)
    __shared__ int qs[8][RB_PB][3];
    const int R = 2 * p.W + 2 * (p.H - 2);
    if (threadIdx.x < 8 * RB_PB) {
        const int t8 = threadIdx.x / RB_PB, j = threadIdx.x % RB_PB, r = blockIdx.x * RB_PB + j, t = rb_tap(t8), ky = t / 3, kx = t % 3;
        int q[3] = {-1, -1, -1}, nq = 0;
        if (r < R) {
            int sy, sx;
            rb_ring_pixel(r, p.H, p.W, sy, sx);
            // pre-images of the ring pixel under the mirror: itself and, for a coordinate of 1 (H-2), the outside coordinate -1 (H)
            for (int aa = 0; aa < 3; ++aa) {
                const int iy = aa == 0 ? sy : (aa == 1 ? (sy == 1 ? -1 : -9) : (sy == p.H - 2 ? p.H : -9));
                if (iy == -9) continue;
                for (int bb = 0; bb < 3; ++bb) {
                    const int ix = bb == 0 ? sx : (bb == 1 ? (sx == 1 ? -1 : -9) : (sx == p.W - 2 ? p.W : -9));
                    if (ix == -9 || (aa == 0 && bb == 0)) continue;          // (both inside: the zero-padded conv's own term)
                    const int oy = iy - ky + 1, ox = ix - kx + 1;
                    if (oy >= 0 && oy < p.H && ox >= 0 && ox < p.W && nq < 3) q[nq++] = oy * p.W + ox;
                }
            }
        }
        qs[t8][j][0] = q[0]; qs[t8][j][1] = q[1]; qs[t8][j][2] = q[2];
    }
    __syncthreads();
    bool active[8];
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8) { bool any = false; for (int j = 0; j < RB_PB; ++j) any |= qs[t8][j][0] >= 0; active[t8] = any; }
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6, tile0 = blockIdx.y * 64, ci = tile0 + cl;
    const float a = p.alpha ? p.alpha[0] : 1.f;
    float* red = &wsh[0][0];                                   // [4 slices][RB_PB * RB_NB][64] after the main loop (8192 <= 16 * 576 floats)
    for (int n0 = 0; n0 < p.N; n0 += RB_NB) {
        const int nn = min(RB_NB, p.N - n0);
        float acc[RB_PB][RB_NB];
#pragma unroll
        for (int j = 0; j < RB_PB; ++j)
#pragma unroll
            for (int n = 0; n < RB_NB; ++n) acc[j][n] = 0.f;
        // software pipeline over the 16-channel chunks of Cout, as in the forward kernel
        const int xn = threadIdx.x & (RB_NB - 1), xj = (threadIdx.x / RB_NB) % RB_PB, xt8 = threadIdx.x / (RB_NB * RB_PB);
        const int q0 = qs[xt8][xj][0], q1 = qs[xt8][xj][1], q2 = qs[xt8][xj][2];
        const bool xlive = active[xt8] && xn < nn;
        float4 wreg[9];
        float dv[16];
        auto fetch = [&](int co0) {
            if ((p.Cin & 3) == 0 && tile0 + 64 <= p.Cin) {
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int i = threadIdx.x + u * 256, c = i / 144, k4 = i % 144;
                    wreg[u] = co0 + c < p.Cout ? *reinterpret_cast<const float4*>(p.w + ((size_t)(co0 + c) * p.Cin + tile0) * 9 + k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) dv[c] = 0.f;
            if (xlive) {
                const bool vec = (p.Cout & 3) == 0 && co0 + 16 <= p.Cout;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int q = k == 0 ? q0 : (k == 1 ? q1 : q2);
                    if (q < 0) continue;
                    const float* src_ = p.dy + ((size_t)(n0 + xn) * p.H * p.W + q) * p.Cout + co0;
                    if (vec) {
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const float4 f = *reinterpret_cast<const float4*>(src_ + c4 * 4);
                            dv[c4 * 4] += f.x; dv[c4 * 4 + 1] += f.y; dv[c4 * 4 + 2] += f.z; dv[c4 * 4 + 3] += f.w;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 16; ++c) if (co0 + c < p.Cout) dv[c] += src_[c];
                    }
                }
            }
        };
        auto commit = [&](int co0) {
            if ((p.Cin & 3) == 0 && tile0 + 64 <= p.Cin) {
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int i = threadIdx.x + u * 256, c = i / 144, k4 = i % 144;
                    *reinterpret_cast<float4*>(&wsh[c][k4 * 4]) = wreg[u];
                }
            } else {          // ragged channel counts: element-wise, not prefetched
#pragma unroll 12
                for (int i = threadIdx.x; i < 16 * 576; i += 256) {
                    const int c = i / 576, k = i % 576;
                    wsh[c][k] = (co0 + c < p.Cout && tile0 + k / 9 < p.Cin) ? p.w[((size_t)(co0 + c) * p.Cin + tile0) * 9 + k] : 0.f;
                }
            }
            if (active[xt8]) {
#pragma unroll
                for (int c = 0; c < 16; ++c) ds[xt8][xj][c][xn] = dv[c];
            }
        };
        fetch(0);
        for (int co0 = 0; co0 < p.Cout; co0 += 16) {
            __syncthreads();
            commit(co0);
            __syncthreads();
            if (co0 + 16 < p.Cout) fetch(co0 + 16);
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                if (!active[t8]) continue;
                const int t = rb_tap(t8);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float wv = wsh[sl * 4 + c][cl * 9 + t];
#pragma unroll
                    for (int j = 0; j < RB_PB; ++j) {
                        const float4 x0 = *reinterpret_cast<const float4*>(&ds[t8][j][sl * 4 + c][0]);
                        const float4 x1 = *reinterpret_cast<const float4*>(&ds[t8][j][sl * 4 + c][4]);
                        acc[j][0] = fmaf(wv, x0.x, acc[j][0]); acc[j][1] = fmaf(wv, x0.y, acc[j][1]); acc[j][2] = fmaf(wv, x0.z, acc[j][2]); acc[j][3] = fmaf(wv, x0.w, acc[j][3]);
                        acc[j][4] = fmaf(wv, x1.x, acc[j][4]); acc[j][5] = fmaf(wv, x1.y, acc[j][5]); acc[j][6] = fmaf(wv, x1.z, acc[j][6]); acc[j][7] = fmaf(wv, x1.w, acc[j][7]);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RB_PB; ++j)
#pragma unroll
            for (int n = 0; n < RB_NB; ++n) red[((size_t)sl * RB_PB * RB_NB + j * RB_NB + n) * 64 + cl] = acc[j][n];
        __syncthreads();
        {   // wave j adds ring pixel j of the group (old values and mask bits loaded together, then stored)
            static_assert(RB_PB == 4, "one wave per pixel of the group");
            const int r = blockIdx.x * RB_PB + sl;
            if (r < R && ci < p.Cin) {
                int sy, sx;
                rb_ring_pixel(r, p.H, p.W, sy, sx);
                const size_t pix0 = (size_t)n0 * p.H * p.W + (size_t)sy * p.W + sx, img = (size_t)p.H * p.W;
                float old[RB_NB];
                bool keep[RB_NB];
#pragma unroll
                for (int n = 0; n < RB_NB; ++n) {
                    old[n] = n < nn ? p.out[(pix0 + n * img) * p.Cin + ci] : 0.f;
                    keep[n] = !(p.mask_hi && n < nn) || (short)p.mask_hi[(pix0 + n * img) * p.mask_c8 + ci] > 0;      // planes of relu(x): positive <=> x > 0 (bf16 and fp16 alike)
                }
#pragma unroll
                for (int n = 0; n < RB_NB; ++n) {
                    const int e = (sl * RB_NB + n) * 64 + cl, q = RB_PB * RB_NB * 64;
                    const float sum = a * ((red[e] + red[q + e]) + (red[2 * q + e] + red[3 * q + e]));
                    if (n < nn && keep[n]) p.out[(pix0 + n * img) * p.Cin + ci] = old[n] + sum;
                }
            }
        }
        __syncthreads();
    }
}

// ---- weight gradient of the border terms: gw[co][ci][t] = sum_n sum_{q: q + t outside} dy[n][q][co] * x[n][mirror(q + t)][ci]   (centre tap: 0)
// grid (9 taps x S item slices, Cout / 16, Cin / 64).  The (image, pixel) items of a tap -- up to N (W + H - 1) of them, against 16 x 64 outputs per
// block -- are cut into S slices so that the thin layers (64 x 64 channels on 256 x 256 maps: 36 tiles) still fill the chip; with S > 1 every slice
// writes its partial tile to the workspace [S][Cout][Cin][9] and reflect_wgrad_reduce_kernel adds the slices in a fixed order.
#define RB_WI 64         // items per LDS stage
struct RbWgradSplit { float* ws; int S; };
template <int PREC>
__global__ __launch_bounds__(256) void reflect_wgrad_kernel(RbParams p, RbWgradSplit sp) {
    __shared__ float dys[RB_WI][16];
    __shared__ float xs[RB_WI][64];
    const int t = blockIdx.x / sp.S, slice = blockIdx.x % sp.S, ky = t / 3, kx = t % 3;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, ci = blockIdx.z * 64 + cl, co0 = blockIdx.y * 16;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int Hs = p.H >> p.up, Ws = p.W >> p.up;
    // output pixels where tap t leaves the image: the row oy_r (ky != 1), the column ox_c (kx != 1), their union for a corner tap
    const int oy_r = ky == 0 ? 0 : p.H - 1, ox_c = kx == 0 ? 0 : p.W - 1;
    const int nrow = ky != 1 ? p.W : 0, ncol = kx != 1 ? (ky != 1 ? p.H - 1 : p.H) : 0, Q = nrow + ncol;
    const int items = p.N * Q, per = (items + sp.S - 1) / sp.S;
    const int i_begin = slice * per, i_end = min(items, i_begin + per);
    for (int i0 = i_begin; i0 < i_end; i0 += RB_WI) {
        __syncthreads();
        {   // stage RB_WI (image, pixel) items: dy[.][16 output channels of this block], x[.][64 input channels of this block]
            const int it = threadIdx.x >> 2, part = threadIdx.x & 3, i = i0 + it;
            const bool live = i < i_end;
            int n = 0, oy = 0, ox = 0;
            if (live) {
                n = i / Q; const int q = i % Q;
                if (q < nrow) { oy = oy_r; ox = q; }
                else { const int j = q - nrow; ox = ox_c; oy = (ky != 1) ? (oy_r == 0 ? j + 1 : j) : j; }
            }
            const float* dyp = p.dy + (((size_t)n * p.H + oy) * p.W + ox) * p.Cout + co0 + part * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) dys[it][part * 4 + j] = (live && co0 + part * 4 + j < p.Cout) ? dyp[j] : 0.f;
            const int my = rb_mirror(oy + ky - 1, p.H) >> p.up, mx = rb_mirror(ox + kx - 1, p.W) >> p.up;
            const size_t xb = (((size_t)n * Hs + my) * Ws + mx) * p.C8 + blockIdx.z * 64 + part * 16;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                xs[it][part * 16 + j] = (live && blockIdx.z * 64 + part * 16 + j < p.Cin) ? rb_plane<PREC>(p.x_hi, p.x_lo, xb + j) : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int it = 0; it < RB_WI; ++it) {
            const float xv = xs[it][cl];
            const float4 d = *reinterpret_cast<const float4*>(&dys[it][g * 4]);
            acc[0] = fmaf(d.x, xv, acc[0]); acc[1] = fmaf(d.y, xv, acc[1]); acc[2] = fmaf(d.z, xv, acc[2]); acc[3] = fmaf(d.w, xv, acc[3]);
        }
    }
    float* out = sp.S > 1 ? sp.ws + (size_t)slice * p.Cout * p.Cin * 9 : p.out;
    if (ci < p.Cin)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (co0 + g * 4 + j < p.Cout) out[((size_t)(co0 + g * 4 + j) * p.Cin + ci) * 9 + t] = acc[j];
}
__global__ __launch_bounds__(256) void reflect_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw, int total, int S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += ws[(size_t)s * total + i];
    gw[i] = a;
}
// slices of the item list: enough blocks for ~8 per CU, at least RB_WI items per slice
static int rb_wgrad_slices(int N, int H, int W, int Cin, int Cout) {
    const int tiles = 9 * ((Cout + 15) / 16) * ((Cin + 63) / 64), items = N * (W + H - 1);
    int S = 2048 / tiles;
    S = S < 1 ? 1 : S;
    const int cap = (items + RB_WI - 1) / RB_WI;
    return S > cap ? (cap < 1 ? 1 : cap) : S;
}
extern "C" long long lp_reflect_border_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    const int S = rb_wgrad_slices(N, H, W, Cin, Cout);
    return S > 1 ? (long long)S * Cout * Cin * 9 * (long long)sizeof(float) : 0;
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
    const dim3 grid((2 * W + 2 * (H - 2) + RB_PB - 1) / RB_PB, (Cout + 63) / 64);
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
    const int ring = 2 * W + 2 * (H - 2);
    hipLaunchKernelGGL(reflect_dgrad_kernel, dim3((ring + RB_PB - 1) / RB_PB, (Cin + 63) / 64), dim3(256), 0, (hipStream_t)stream, p);
    return lp_check_launch("lp_reflect_border_dgrad");
}

extern "C" int lp_reflect_border_wgrad(const uint16_t* x_hi, const uint16_t* x_lo, int prec, int N, int H, int W, int Cin, int C8, int upsample,
                                       const float* dy, int Cout, float* gw, float* workspace, void* stream) {
    if (!x_hi || !dy || !gw || (prec == LP_PREC_BF16X3 && !x_lo)) return lp_set_error(LP_ERR_ARG, "lp_reflect_border_wgrad: null argument");
    if (int e = rb_check("lp_reflect_border_wgrad: bad geometry", N, H, W, Cin, Cout, upsample)) return e;
    RbParams p{}; p.x_hi = x_hi; p.x_lo = x_lo; p.dy = dy; p.out = gw;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.C8 = C8; p.up = upsample ? 1 : 0;
    RbWgradSplit sp{workspace, rb_wgrad_slices(N, H, W, Cin, Cout)};
    if (sp.S > 1 && !workspace) return lp_set_error(LP_ERR_ARG, "lp_reflect_border_wgrad: workspace of lp_reflect_border_wgrad_workspace_bytes() needed");
    const dim3 grid(9 * sp.S, (Cout + 15) / 16, (Cin + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    if (prec == LP_PREC_F16) hipLaunchKernelGGL(reflect_wgrad_kernel<LP_PREC_F16>, grid, dim3(256), 0, st, p, sp);
    else if (prec == LP_PREC_BF16X3) hipLaunchKernelGGL(reflect_wgrad_kernel<LP_PREC_BF16X3>, grid, dim3(256), 0, st, p, sp);
    else if (prec == LP_PREC_BF16) hipLaunchKernelGGL(reflect_wgrad_kernel<LP_PREC_BF16>, grid, dim3(256), 0, st, p, sp);
    else return lp_set_error(LP_ERR_ARG, "lp_reflect_border_wgrad: unknown operand mode");
    if (sp.S > 1) {
        const int total = Cout * Cin * 9;
        hipLaunchKernelGGL(reflect_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, gw, total, sp.S);
    }
    return lp_check_launch("lp_reflect_border_wgrad");
}
