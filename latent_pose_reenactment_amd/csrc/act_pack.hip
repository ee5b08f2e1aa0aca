// Operand packing for conv_dma.hip / conv_wgrad.hip (gfx950): the activated input of a conv is written ONCE as 16-bit planes
//   hi[p][c] (, lo[p][c])  =  split( act( x[p][c] ) * in_scale ),   [N*HW][C8] with C8 = C rounded up to 8 (pad channels = 0)
// act = identity | relu(x*scale[n,c]+shift[n,c]) (AdaIN + ReLU, generators/common/blocks.py:18-26,70-73) | relu(x) | relu6 / relu /
// identity of a per-channel BatchNorm affine x*scale[c]+shift[c] (the embedder's conv -> BatchNorm -> ReLU chains).
// Bandwidth-bound: 4 B read, 2 B (bf16 / f16) or 4 B (bf16x3: hi + lo) written per element; every thread converts 8 channels of
// one pixel (two 16-B loads, one 16-B store per plane).  fp16 gradients are scaled by a power of two taken from the tensor's
// amax (lp_amax_scale) so that they sit in the fp16 normal range; the consumer multiplies its result by 1/in_scale.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

// grid = (blocks, N): blockIdx.y = image; a thread walks the (pixel, 8-channel group) items of its image with a stride that is a
// multiple of the group count G whenever G is a power of two, so its group -- and with it the AdaIN scale/shift of its 8 channels
// -- is fixed and loaded once; 4 items are in flight per thread (8 x 16-B loads) to cover the HBM latency.
#define AMAX_BLOCKS 1024
// amax -> power-of-two input scale of an fp16 gradient operand: s = 2^(13 - e) with amax = f * 2^e, f in [0.5, 1): the scaled
// tensor's amax lands in [2^12, 2^13) -- 3 binades of headroom below the fp16 maximum, 26 binades of normal range below
__device__ __forceinline__ void amax_to_scale(float m, float& s, float& inv) {
    s = 1.f; inv = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        int e;
        (void)frexpf(m, &e);
        int k = 13 - e;
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        s = ldexpf(1.f, k); inv = ldexpf(1.f, -k);
    }
}

template <int PREC>
__global__ __launch_bounds__(256) void act_pack_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int pro, uint16_t* __restrict__ hi,
                                                       uint16_t* __restrict__ lo, int HW, int C, int C8,
                                                       const float* __restrict__ in_scale, const float* __restrict__ amax_part,
                                                       int amax_count, int amax_stride, float* __restrict__ scale_out) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    constexpr int U = 4;
    float isc = in_scale ? in_scale[0] : 1.f;
    if (amax_part) {
        // gradient operand: every block finishes the amax reduction itself (AMAX_BLOCKS partials from lp_amax_partial, L2 resident)
        // and derives the same power-of-two scale; block (0,0) publishes {s, 1/s} for the consumers of the planes
        __shared__ float sh[4];
        float m = 0.f;
        for (int j = threadIdx.x; j < amax_count; j += 256) m = fmaxf(m, amax_part[(size_t)j * amax_stride]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        float inv;
        amax_to_scale(m, isc, inv);
        if (scale_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { scale_out[0] = isc; scale_out[1] = inv; }
    }
    const unsigned G = (unsigned)C8 >> 3;
    const unsigned items = (unsigned)HW * G;
    const unsigned stride = gridDim.x * 256u;
    const int n = blockIdx.y;
    const bool vec = (C & 3) == 0;
    const float lo_clamp = (pro != 0 && pro != 5) ? 0.f : -3.0e38f;
    const float hi_clamp = (pro == 3) ? 6.f : 3.0e38f;          // pro 3 = ReLU6, 4 = ReLU, 5 = identity of a per-channel (batch-norm) affine
    const bool aff = (pro == 1 || pro >= 3);
    const size_t aoff = (pro == 1) ? (size_t)n * C : 0;
    const float* xn = x + (size_t)n * HW * C;
    uint16_t* hn = hi + (size_t)n * HW * C8;
    uint16_t* ln = SPLIT ? lo + (size_t)n * HW * C8 : nullptr;
    const bool fixed_g = (stride % G) == 0;
    float sc[8], sf[8];
    auto load_affine = [&](unsigned g) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = (int)g * 8 + j;
            sc[j] = (aff && c < C) ? scale[aoff + c] : 1.f;
            sf[j] = (aff && c < C) ? shift[aoff + c] : 0.f;
        }
    };
    unsigned i0 = blockIdx.x * 256u + threadIdx.x;
    if (fixed_g) load_affine(i0 % G);
    for (; i0 < items; i0 += U * stride) {
        float v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u * stride;
            const unsigned ic = i < items ? i : i0;                 // clamped: loads stay unconditional
            const unsigned g = ic % G, pix = ic / G;
            const int c = (int)g * 8;
            const float* src = xn + (size_t)pix * C + c;
            if (vec && c + 8 <= C) {
                const float4 p0 = *(const float4*)src, p1 = *(const float4*)(src + 4);
                v[u][0] = p0.x; v[u][1] = p0.y; v[u][2] = p0.z; v[u][3] = p0.w; v[u][4] = p1.x; v[u][5] = p1.y; v[u][6] = p1.z; v[u][7] = p1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[u][j] = (c + j < C) ? src[j] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u * stride;
            if (i < items) {
                const unsigned g = i % G;
                const int c = (int)g * 8;
                if (!fixed_g) load_affine(g);
                s16x8_t h, l;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float q = fminf(fmaxf(fmaf(v[u][j], sc[j], sf[j]), lo_clamp), hi_clamp) * isc;
                    q = (c + j < C) ? q : 0.f;
                    const uint16_t hb = lp_f32_to_op16<F16>(q);
                    h[j] = (short)hb;
                    if (SPLIT) l[j] = (short)lp_f32_to_op16<false>(q - lp_op16_to_f32<false>(hb));
                }
                *(s16x8_t*)(hn + (size_t)i * 8) = h;
                if (SPLIT) *(s16x8_t*)(ln + (size_t)i * 8) = l;
            }
        }
    }
}

extern "C" int lp_act_pack(const float* x, const float* scale, const float* shift, int pro, uint16_t* hi, uint16_t* lo,
                           int N, int HW, int C, int prec, const float* in_scale, const float* amax_part, int amax_count,
                           int amax_stride, float* scale_out, void* stream) {
    if (!x || !hi) return lp_set_error(LP_ERR_ARG, "lp_act_pack: null pointer");
    if (pro < 0 || pro > 5) return lp_set_error(LP_ERR_ARG, "lp_act_pack: pro must be 0..5");
    if ((pro == 1 || pro >= 3) && (!scale || !shift)) return lp_set_error(LP_ERR_ARG, "lp_act_pack: pro=1|3|4|5 needs scale/shift");
    if (prec == LP_PREC_BF16X3 && !lo) return lp_set_error(LP_ERR_ARG, "lp_act_pack: bf16x3 needs the lo plane");
    if (amax_part && (amax_count < 1 || amax_stride < 1)) return lp_set_error(LP_ERR_ARG, "lp_act_pack: amax_count / amax_stride must be >= 1 with amax_part");
    const int C8 = (C + 7) & ~7;
    const long long items = (long long)HW * (C8 >> 3);          // per image
    if (items == 0 || N == 0) return LP_OK;
    if (items >= (1ll << 31)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_act_pack: image too large");
    long long bx = (items + 1023) / 1024;                       // 4 items per thread
    const long long cap = (4096 + N - 1) / N; if (bx > cap) bx = cap; if (bx < 1) bx = 1;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)bx, (unsigned)N);
#define LP_AP(P) hipLaunchKernelGGL(act_pack_kernel<P>, grid, dim3(256), 0, st, x, scale, shift, pro, hi, lo, HW, C, C8, in_scale, amax_part, amax_count, amax_stride, scale_out)
    if (prec == LP_PREC_BF16) LP_AP(LP_PREC_BF16);
    else if (prec == LP_PREC_BF16X3) LP_AP(LP_PREC_BF16X3);
    else if (prec == LP_PREC_F16) LP_AP(LP_PREC_F16);
    else return lp_set_error(LP_ERR_ARG, "lp_act_pack: unknown precision mode");
#undef LP_AP
    return lp_check_launch("act_pack");
}

// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// Backward of conv + AvgPool2d(2) [+ the next block's ReLU] as ONE pass over the pooled gradient (round 6, nn.ConvPoolFn.backward):
//   dm = dy * [ymask > 0]                      the ReLU behind the pool (ymask = the stored relu(pooled) output | NULL: no ReLU)
//   lo planes [N][h][w][C]   = split(dm * s)   operand of the data-gradient launch (and of the skip conv's backward)
//   up planes [N][2h][2w][C] = the same 16-bit values at the four positions of every 2 x 2 window: the operand planes of 0.25 * nearest_up2(dm) --
//                              the pool's adjoint, operand of the weight-gradient launch -- under the scale 4 s (fp16) / as split(0.25 dm) (bf16 modes:
//                              a power-of-two factor commutes with the hi / lo split)
// It replaces gt + where (2 ATen launches), lp_act_pack(dm), lp_avgpool2_bwd (a full-resolution fp32 tensor written and read back) and
// lp_act_pack of that tensor.  fp16: s = the power of two lp_act_pack would take from amax(dy) (>= amax(dm): no overflow);
// scale_out = {s, 1/s, 4 s, 1/(4 s)}.  Every thread owns 8 channels of one pooled pixel.
// ------------------------------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(256) void pool_grad_pack_kernel(const float* __restrict__ dy, const float* __restrict__ ymask, float* __restrict__ dm,
                                                             uint16_t* __restrict__ lo_hi, uint16_t* __restrict__ lo_lo,
                                                             uint16_t* __restrict__ up_hi, uint16_t* __restrict__ up_lo, long long items, int h, int w, int C,
                                                             const float* __restrict__ amax_part, int amax_count, int amax_stride, float* __restrict__ scale_out) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    float isc = 1.f;
    if (F16 && amax_part) {
        __shared__ float sh[4];
        float m = 0.f;
        for (int j = threadIdx.x; j < amax_count; j += 256) m = fmaxf(m, amax_part[(size_t)j * amax_stride]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        float inv;
        amax_to_scale(m, isc, inv);
        if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) { scale_out[0] = isc; scale_out[1] = inv; scale_out[2] = 4.f * isc; scale_out[3] = 0.25f * inv; }
    }
    const int G = C >> 3;
    const float upf = F16 ? 1.f : 0.25f;          // fp16: the factor lives in the planes' scale; bf16 modes: in the values
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        const long long pix = i / G;
        const size_t e = (size_t)pix * C + (size_t)g * 8;
        float4 a = *(const float4*)(dy + e), b = *(const float4*)(dy + e + 4);
        if (ymask) {
            const float4 ma = *(const float4*)(ymask + e), mb = *(const float4*)(ymask + e + 4);
            a.x = ma.x > 0.f ? a.x : 0.f; a.y = ma.y > 0.f ? a.y : 0.f; a.z = ma.z > 0.f ? a.z : 0.f; a.w = ma.w > 0.f ? a.w : 0.f;
            b.x = mb.x > 0.f ? b.x : 0.f; b.y = mb.y > 0.f ? b.y : 0.f; b.z = mb.z > 0.f ? b.z : 0.f; b.w = mb.w > 0.f ? b.w : 0.f;
        }
        if (dm) { *(float4*)(dm + e) = a; *(float4*)(dm + e + 4) = b; }
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        s16x8_t hl, ll, hu, lu;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float q = v[j] * isc;
            const uint16_t hb = lp_f32_to_op16<F16>(q);
            hl[j] = (short)hb;
            if (SPLIT) ll[j] = (short)lp_f32_to_op16<false>(q - lp_op16_to_f32<false>(hb));
            if (F16) { hu[j] = hl[j]; }
            else {
                const float qu = q * upf;
                const uint16_t hbu = lp_f32_to_op16<false>(qu);
                hu[j] = (short)hbu;
                if (SPLIT) lu[j] = (short)lp_f32_to_op16<false>(qu - lp_op16_to_f32<false>(hbu));
            }
        }
        if (lo_hi) { *(s16x8_t*)(lo_hi + e) = hl; if (SPLIT) *(s16x8_t*)(lo_lo + e) = ll; }
        if (up_hi) {
            const int x = (int)(pix % w); const long long t = pix / w; const int y = (int)(t % h); const long long n = t / h;
            const size_t row = ((size_t)(n * 2 * h + 2 * y) * (2 * w) + 2 * x) * C + (size_t)g * 8;
            const size_t dn = (size_t)2 * w * C;
            *(s16x8_t*)(up_hi + row) = hu; *(s16x8_t*)(up_hi + row + C) = hu; *(s16x8_t*)(up_hi + row + dn) = hu; *(s16x8_t*)(up_hi + row + dn + C) = hu;
            if (SPLIT) { *(s16x8_t*)(up_lo + row) = lu; *(s16x8_t*)(up_lo + row + C) = lu; *(s16x8_t*)(up_lo + row + dn) = lu; *(s16x8_t*)(up_lo + row + dn + C) = lu; }
        }
    }
}

extern "C" int lp_pool_grad_pack(const float* dy, const float* ymask, float* dm, uint16_t* lo_hi, uint16_t* lo_lo, uint16_t* up_hi, uint16_t* up_lo,
                                 int N, int h, int w, int C, int prec, const float* amax_part, int amax_count, int amax_stride, float* scale_out,
                                 void* stream) {
    if (!dy || (!lo_hi && !up_hi && !dm)) return lp_set_error(LP_ERR_ARG, "lp_pool_grad_pack: null pointer");
    if (C & 7) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_pool_grad_pack: C must be a multiple of 8");
    if (prec == LP_PREC_BF16X3 && ((lo_hi && !lo_lo) || (up_hi && !up_lo))) return lp_set_error(LP_ERR_ARG, "lp_pool_grad_pack: bf16x3 needs the lo planes");
    if (prec == LP_PREC_F16 && (!amax_part || amax_count < 1 || amax_stride < 1 || !scale_out))
        return lp_set_error(LP_ERR_ARG, "lp_pool_grad_pack: the fp16 mode needs the amax partials of dy (lp_amax_partial) and scale_out[4]");
    const long long items = (long long)N * h * w * (C >> 3);
    if (items == 0) return LP_OK;
    long long blocks = (items + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
#define LP_PG(P) hipLaunchKernelGGL(pool_grad_pack_kernel<P>, dim3((unsigned)blocks), dim3(256), 0, st, dy, ymask, dm, lo_hi, lo_lo, up_hi, up_lo, items, h, w, C, amax_part, amax_count, amax_stride, scale_out)
    if (prec == LP_PREC_BF16) LP_PG(LP_PREC_BF16);
    else if (prec == LP_PREC_BF16X3) LP_PG(LP_PREC_BF16X3);
    else if (prec == LP_PREC_F16) LP_PG(LP_PREC_F16);
    else return lp_set_error(LP_ERR_ARG, "lp_pool_grad_pack: unknown precision mode");
#undef LP_PG
    return lp_check_launch("pool_grad_pack");
}

// amax partials of a gradient tensor: part[AMAX_BLOCKS] block maxima of |x| (every entry written); lp_act_pack finishes the
// reduction (amax_part argument), so a gradient operand costs one extra streaming read and no extra finalize launch
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_partial_kernel(const float4* __restrict__ x, long long total4, const float* __restrict__ tail,
                                                           int ntail, float* __restrict__ part) {
    __shared__ float sh[4];
    float m = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < total4; i += 4 * stride) {          // 4 independent 16-B loads in flight per thread
        const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))), fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
    }
    for (; i < total4; i += stride) {
        const float4 v = x[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) m = fmaxf(m, fabsf(tail[threadIdx.x]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

extern "C" int lp_amax_blocks(void) { return AMAX_BLOCKS; }
extern "C" int lp_amax_slots(void) { return LP_AMAX_SLOTS; }
extern "C" int lp_amax_slot_stride(void) { return LP_AMAX_STRIDE; }

extern "C" int lp_amax_partial(const float* x, long long numel, float* part, void* stream) {
    if (!x || !part) return lp_set_error(LP_ERR_ARG, "lp_amax_partial: null pointer");
    const long long total4 = numel / 4;
    hipLaunchKernelGGL(amax_partial_kernel, dim3(AMAX_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const float4*)x, total4, x + total4 * 4,
                       (int)(numel - total4 * 4), part);
    return lp_check_launch("amax_partial");
}
