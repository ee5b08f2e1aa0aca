// Shared device helpers for the latent-pose gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) short s16x8_t;     // 8 x 16-bit lanes = one MFMA 16x16x32 A/B fragment
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LP_WAVE 64

// precision modes of the contraction operands (accumulation is always fp32 in the MFMA)
#define LP_PREC_BF16 0      // operands rounded to bf16 (RNE), 1 MFMA per k-step
#define LP_PREC_BF16X3 1    // operands split hi+lo bf16, 3 MFMAs per k-step (~2^-16 relative operand error)
#define LP_PREC_F16 2       // operands rounded to IEEE fp16 (RNE, 2^-12), 1 MFMA per k-step; gradients need a power-of-two input scale

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// 16-bit operand element of a precision mode -> fp32 (used by the thin-channel kernels and bias-gradient sums)
template <bool F16> __device__ __forceinline__ float lp_op16_to_f32(uint16_t b) {
    if (F16) return (float)__builtin_bit_cast(_Float16, b);
    return __uint_as_float((unsigned)b << 16);
}
// fp32 -> 16-bit operand (hi) and residual (lo, bf16x3 only); fp16 saturates finite values instead of overflowing to inf (a NaN
// propagates: a diverging run must not be masked at the operand)
template <bool F16> __device__ __forceinline__ uint16_t lp_f32_to_op16(float v) {
    if (F16) { const float c = fminf(fmaxf(v, -65504.f), 65504.f); v = (v != v) ? v : c; return __builtin_bit_cast(uint16_t, (_Float16)v); }   // NaN stays NaN
    return __builtin_bit_cast(uint16_t, (__bf16)v);
}

__device__ __forceinline__ f32x4_t mfma16(s16x8_t a, s16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// element-type generic form: F16 operands use the fp16 MFMA (same rate and fragment layout as the bf16 one)
template <bool F16> __device__ __forceinline__ f32x4_t mfma16t(s16x8_t a, s16x8_t b, f32x4_t c) {
    if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// LDS-DMA (16 B per lane, wave-uniform LDS destination + lane*16) issued through inline asm: hipcc's waitcnt pass would
// otherwise put `s_waitcnt vmcnt(0)` in front of the next ds_read (it cannot prove the DMA target does not alias it), which
// serialises the prefetch with the MFMAs.  The caller owns the completion: `lp_wait_vm0()` + barrier before reading the tile.
// M0 is compiler-reserved, so it is saved/restored inside the same statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void lp_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Producer-side amax of a gradient tensor: kernels that WRITE a tensor which lp_act_pack will turn into fp16 gradient planes fold
// max|v| of what they write into LP_AMAX_SLOTS pre-zeroed floats, LP_AMAX_STRIDE floats (one 128-B line, so the slots spread over
// the L2 channels) apart: per wave one atomic max on the IEEE bit pattern (non-negative floats order like unsigned integers; the
// result does not depend on the order of arrival), fire-and-forget.  lp_act_pack(amax_part =
// slots, amax_count = LP_AMAX_SLOTS, amax_stride = LP_AMAX_STRIDE) then needs no separate pass over the tensor.  Call from every
// lane of the wave.
#define LP_AMAX_SLOTS 64
#define LP_AMAX_STRIDE 32
__device__ __forceinline__ float lp_amax4(float m, const float4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ void lp_amax_commit(float m, float* slots, unsigned key) {
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) {
        unsigned* s = (unsigned*)slots + ((key + (threadIdx.x >> 6)) & (LP_AMAX_SLOTS - 1)) * LP_AMAX_STRIDE;
        if (m > 0.f) atomicMax(s, __float_as_uint(m));          // no return value: the wave does not wait for it
    }
}

__device__ __forceinline__ void lp_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N> __device__ __forceinline__ void lp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

// Geometry of one output tile: NB images x (TH x TW) pixel patch, all powers of two, TH,TW >= 2.
struct TileGeom {
    int lTH, lTW, lNB;        // log2
    int tiles_x, tiles_y;     // patches per image
};

// Row index m (0..BM) of a tile -> (image-in-tile, py, px).  Rows 4q..4q+3 form the 2x2 pixel block q so that the four
// accumulator registers of a lane in the 16x16 MFMA C layout are one 2x2 block (used by fused pooling epilogues).
__device__ __forceinline__ void tile_row_decode(int m, int lTH, int lTW, int& nb, int& py, int& px) {
    int bq = m >> 2, sub = m & 3;
    int lbw = lTW - 1, lbh = lTH - 1;
    int bx = bq & ((1 << lbw) - 1);
    int by = (bq >> lbw) & ((1 << lbh) - 1);
    nb = bq >> (lbw + lbh);
    py = 2 * by + (sub >> 1);
    px = 2 * bx + (sub & 1);
}

// Row index m of a tile in LINEAR order (row-major inside the patch) -> (image-in-tile, py, px): 16 consecutive rows are 16
// consecutive pixels of one patch row (TW = 16), i.e. consecutive halo pixels -- what the lane-linear LDS-DMA image of
// conv_dma.hip and its XOR key are designed for (scripts/lds_swizzle_sim.py).
__device__ __forceinline__ void tile_row_linear(int m, int lTH, int lTW, int& nb, int& py, int& px) {
    px = m & ((1 << lTW) - 1);
    py = (m >> lTW) & ((1 << lTH) - 1);
    nb = m >> (lTW + lTH);
}

// Linear pixel index kpix (row-major inside the patch) -> (image-in-tile, py, px); used for the wgrad k dimension.
__device__ __forceinline__ void tile_lin_decode(int kpix, int lTH, int lTW, int& nb, int& py, int& px) {
    px = kpix & ((1 << lTW) - 1);
    py = (kpix >> lTW) & ((1 << lTH) - 1);
    nb = kpix >> (lTW + lTH);
}


// Pixels per stage-1 workgroup of the per-channel reductions (norm statistics, norm-backward sums): a workgroup covers 64 channels x
// `pix` pixels of one image.  1024 pixels where that already gives >= 2048 workgroups (the 256^2 maps); fewer for the smaller tensors
// (multiples of 16, >= 64) so that the launch still covers the 256 CUs -- the embedder's 8 x 64 x 64 x 256 BatchNorm had 128 workgroups
// of 64 serial iterations at the fixed split (profiles/r03_bn_bwd16_micro.txt).
static inline int lp_stat_split_pix(long long images, long long HW, int C) {
    static const bool fixed = getenv("LP_STAT_SPLIT_FIXED") != nullptr;       // A/B knob: the fixed 1024-pixel split of rounds 1-2
    if (fixed) return 1024;
    const long long cb = (C + 63) / 64;
    long long pb = images * HW * cb / 2048;
    pb = pb / 16 * 16;
    if (pb < 64) pb = 64;
    if (pb > 1024) pb = 1024;
    return (int)pb;
}
