// Diagnosis only (not part of liblp_hip.so): canary kernels that run beside the convolution kernels on another stream and report whether
// their registers, their LDS, or the data returned by their global loads were disturbed.  scripts/canary_probe.py drives them.
#include <hip/hip_runtime.h>
#include <stdint.h>

// out[0] = mismatches, out[1..] = histogram: [1 + lane/16] by quarter-wave, [8 + comp] by dword of the 16-byte load, out[16] first bad index, out[17] got, out[18] want
__device__ __forceinline__ void report(unsigned* out, int lane, int comp, unsigned idx, unsigned got, unsigned want) {
    if (atomicAdd(&out[0], 1u) == 0u) { out[16] = idx; out[17] = got; out[18] = want; out[19] = (unsigned)lane; }
    atomicAdd(&out[1 + (lane >> 4)], 1u);
    if (comp >= 0) atomicAdd(&out[8 + comp], 1u);
}

// C: the power iteration's access pattern over a buffer whose dword at index i holds i: rows x C dwords, 32-row blocks, 4 x 16-byte loads in flight
extern "C" __global__ __launch_bounds__(256) void load_canary(const uint4* __restrict__ w4, int rows, int C, unsigned* out, int repeats) {
    const int r0 = blockIdx.x * 32, C4 = C >> 2;
    for (int rep = 0; rep < repeats; ++rep)
        for (int c = threadIdx.x; c < C4; c += 256)
            for (int r = r0; r + 4 <= r0 + 32 && r + 4 <= rows; r += 4) {
                uint4 x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = w4[(size_t)(r + k) * C4 + c];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned base = (unsigned)((r + k) * C + 4 * c);
                    if (x[k].x != base)     report(out, threadIdx.x & 63, 0, base, x[k].x, base);
                    if (x[k].y != base + 1) report(out, threadIdx.x & 63, 1, base + 1, x[k].y, base + 1);
                    if (x[k].z != base + 2) report(out, threadIdx.x & 63, 2, base + 2, x[k].z, base + 2);
                    if (x[k].w != base + 3) report(out, threadIdx.x & 63, 3, base + 3, x[k].w, base + 3);
                }
            }
}

// B: a small static LDS array (like the power iteration's 32 floats) checked over and over
extern "C" __global__ __launch_bounds__(256) void lds_canary(unsigned* out, int repeats) {
    __shared__ unsigned s[256];
    s[threadIdx.x] = 0xA5000000u + blockIdx.x * 256u + threadIdx.x;
    __syncthreads();
    for (int rep = 0; rep < repeats; ++rep) {
        __builtin_amdgcn_s_sleep(32);
        const unsigned want = 0xA5000000u + blockIdx.x * 256u + ((threadIdx.x + rep) & 255);
        const unsigned got = ((volatile unsigned*)s)[(threadIdx.x + rep) & 255];
        if (got != want) report(out, threadIdx.x & 63, -1, (threadIdx.x + rep) & 255, got, want);
    }
}

// A: 32 registers per lane holding known values across a long sleep
extern "C" __global__ __launch_bounds__(256) void reg_canary(unsigned* out, int repeats) {
    unsigned r[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { r[i] = 0x5A000000u + (blockIdx.x * 256u + threadIdx.x) * 32u + i; asm volatile("" : "+v"(r[i])); }
    for (int rep = 0; rep < repeats; ++rep) {
        __builtin_amdgcn_s_sleep(64);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            asm volatile("" : "+v"(r[i]));
            const unsigned want = 0x5A000000u + (blockIdx.x * 256u + threadIdx.x) * 32u + i;
            if (r[i] != want) { report(out, threadIdx.x & 63, i & 3, i, r[i], want); r[i] = want; }
        }
    }
}

// D: the arithmetic of the power iteration's inner loop from registers only, twice, compared with itself
extern "C" __global__ __launch_bounds__(256) void fma_canary(unsigned* out, int repeats) {
    __shared__ float us[32];
    if (threadIdx.x < 32) us[threadIdx.x] = 0.01f * (float)(threadIdx.x + 1);
    __syncthreads();
    float4 first = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rep = 0; rep < repeats; ++rep) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        for (int r = 0; r < 32; r += 4) {
            float4 x0 = make_float4(threadIdx.x * 0.5f + r, threadIdx.x * 0.25f - r, 1.f + r, 2.f - r), x1 = x0, x2 = x0, x3 = x0;
            x1.x += 1.f; x2.y += 2.f; x3.z += 3.f;
            asm volatile("" : "+v"(x0.x), "+v"(x1.y), "+v"(x2.z), "+v"(x3.w));
            const float u0 = us[r], u1 = us[r + 1], u2 = us[r + 2], u3 = us[r + 3];
            a.x = fmaf(x0.x, u0, a.x); a.y = fmaf(x0.y, u0, a.y); a.z = fmaf(x0.z, u0, a.z); a.w = fmaf(x0.w, u0, a.w);
            b.x = fmaf(x1.x, u1, b.x); b.y = fmaf(x1.y, u1, b.y); b.z = fmaf(x1.z, u1, b.z); b.w = fmaf(x1.w, u1, b.w);
            a.x = fmaf(x2.x, u2, a.x); a.y = fmaf(x2.y, u2, a.y); a.z = fmaf(x2.z, u2, a.z); a.w = fmaf(x2.w, u2, a.w);
            b.x = fmaf(x3.x, u3, b.x); b.y = fmaf(x3.y, u3, b.y); b.z = fmaf(x3.z, u3, b.z); b.w = fmaf(x3.w, u3, b.w);
        }
        const float4 s = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        if (rep == 0) first = s;
        else {
            if (__float_as_uint(s.x) != __float_as_uint(first.x)) report(out, threadIdx.x & 63, 0, rep, __float_as_uint(s.x), __float_as_uint(first.x));
            if (__float_as_uint(s.y) != __float_as_uint(first.y)) report(out, threadIdx.x & 63, 1, rep, __float_as_uint(s.y), __float_as_uint(first.y));
            if (__float_as_uint(s.z) != __float_as_uint(first.z)) report(out, threadIdx.x & 63, 2, rep, __float_as_uint(s.z), __float_as_uint(first.z));
            if (__float_as_uint(s.w) != __float_as_uint(first.w)) report(out, threadIdx.x & 63, 3, rep, __float_as_uint(s.w), __float_as_uint(first.w));
        }
    }
}


// E / F: the same loads through a pointer that was itself loaded from device memory (a descriptor table, as lp_sn_power_iter does): the compiler
// cannot prove the address space and emits FLAT loads (E); F casts the pointer to the global address space first (global_load)
typedef unsigned u4v __attribute__((ext_vector_type(4)));
struct Desc { const u4v* w; unsigned* out; int rows, C; };
template <bool GLOBAL_AS>
__device__ __forceinline__ void table_body(const Desc* __restrict__ table, int repeats) {
    const Desc d = table[blockIdx.y];
    typedef const u4v __attribute__((address_space(1)))* gptr;
    const int r0 = blockIdx.x * 32, C4 = d.C >> 2;
    __shared__ float us[32];
    if (threadIdx.x < 32) us[threadIdx.x] = 1.f;
    __syncthreads();
    for (int rep = 0; rep < repeats; ++rep)
        for (int c = threadIdx.x; c < C4; c += 256)
            for (int r = r0; r + 4 <= r0 + 32 && r + 4 <= d.rows; r += 4) {
                u4v x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { if (GLOBAL_AS) x[k] = ((gptr)d.w)[(size_t)(r + k) * C4 + c]; else x[k] = d.w[(size_t)(r + k) * C4 + c]; }
                const float keep = us[r - r0] + us[r - r0 + 1] + us[r - r0 + 2] + us[r - r0 + 3];
                if (keep != 4.f) report(d.out, threadIdx.x & 63, -1, 0xFFFFFFFFu, __float_as_uint(keep), 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned base = (unsigned)((r + k) * d.C + 4 * c);
                    if (x[k].x != base)     report(d.out, threadIdx.x & 63, 0, base, x[k].x, base);
                    if (x[k].y != base + 1) report(d.out, threadIdx.x & 63, 1, base + 1, x[k].y, base + 1);
                    if (x[k].z != base + 2) report(d.out, threadIdx.x & 63, 2, base + 2, x[k].z, base + 2);
                    if (x[k].w != base + 3) report(d.out, threadIdx.x & 63, 3, base + 3, x[k].w, base + 3);
                }
            }
}
extern "C" __global__ __launch_bounds__(256) void flat_canary(const Desc* __restrict__ table, int repeats) { table_body<false>(table, repeats); }
extern "C" __global__ __launch_bounds__(256) void global_canary(const Desc* __restrict__ table, int repeats) { table_body<true>(table, repeats); }
extern "C" int canary_table_launch(int global_as, const void* table, int rows, int repeats, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (global_as) hipLaunchKernelGGL(global_canary, dim3((rows + 31) / 32, 1), dim3(256), 0, st, (const Desc*)table, repeats);
    else hipLaunchKernelGGL(flat_canary, dim3((rows + 31) / 32, 1), dim3(256), 0, st, (const Desc*)table, repeats);
    return (int)hipGetLastError();
}

// G: v_pk_fma_f32 against two v_fmac_f32 on the same register inputs.  FORM 0: no operand selection; 1: op_sel:[0,1,0] (both halves take src1's high
// dword); 2: op_sel_hi:[1,0,1] (both halves take src1's low dword) -- the two forms the compiler emits for "vector times broadcast scalar".
typedef float f2v __attribute__((ext_vector_type(2)));
template <int FORM>
__device__ __forceinline__ void pk_body(unsigned* out, int repeats) {
    for (int rep = 0; rep < repeats; ++rep) {
        f2v acc; acc.x = 0.f; acc.y = 0.f;
        float lo = 0.f, hi = 0.f;
        for (int i = 0; i < 32; ++i) {
            f2v x, u;
            x.x = 0.001f * (float)((threadIdx.x * 7 + i * 13 + rep) & 1023) - 0.5f; x.y = 0.002f * (float)((threadIdx.x * 5 + i * 11 + rep) & 511) - 0.5f;
            u.x = 0.03f * (float)(i + 1); u.y = -0.02f * (float)(i + 3);
            asm volatile("" : "+v"(x), "+v"(u));
            if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(u));
            if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(x), "v"(u));
            if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(u));
            const float ul = FORM == 1 ? u.y : u.x, uh = FORM == 2 ? u.x : u.y;
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(lo) : "v"(x.x), "v"(ul));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(hi) : "v"(x.y), "v"(uh));
        }
        if (__float_as_uint(acc.x) != __float_as_uint(lo)) report(out, threadIdx.x & 63, 0, rep, __float_as_uint(acc.x), __float_as_uint(lo));
        if (__float_as_uint(acc.y) != __float_as_uint(hi)) report(out, threadIdx.x & 63, 1, rep, __float_as_uint(acc.y), __float_as_uint(hi));
    }
}
extern "C" __global__ __launch_bounds__(256) void pk_canary0(unsigned* out, int repeats) { pk_body<0>(out, repeats); }
extern "C" __global__ __launch_bounds__(256) void pk_canary1(unsigned* out, int repeats) { pk_body<1>(out, repeats); }
extern "C" __global__ __launch_bounds__(256) void pk_canary2(unsigned* out, int repeats) { pk_body<2>(out, repeats); }
extern "C" int canary_pk_launch(int form, unsigned* out, int blocks, int repeats, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (form == 0) hipLaunchKernelGGL(pk_canary0, dim3(blocks), dim3(256), 0, st, out, repeats);
    else if (form == 1) hipLaunchKernelGGL(pk_canary1, dim3(blocks), dim3(256), 0, st, out, repeats);
    else hipLaunchKernelGGL(pk_canary2, dim3(blocks), dim3(256), 0, st, out, repeats);
    return (int)hipGetLastError();
}

// H: every operand-selection form of the packed fp32 instructions that occurs in liblp_hip.so, one kernel per form.  sel bit = 0: low dword, 1: high dword.
#define SEL(v, bit) ((bit) ? (v).y : (v).x)
#define PK_FMA_KERNEL(name, mods, l0, l1, l2, h0, h1, h2)                                                                          \
extern "C" __global__ __launch_bounds__(256) void name(unsigned* out, int repeats) {                                              \
    for (int rep = 0; rep < repeats; ++rep)                                                                                        \
        for (int i = 0; i < 16; ++i) {                                                                                             \
            f2v x, u, a, r;                                                                                                        \
            x.x = 0.001f * (float)((threadIdx.x * 7 + i * 13 + rep) & 1023) - 0.5f; x.y = 0.002f * (float)((threadIdx.x * 5 + i * 11 + rep) & 511) - 0.5f; \
            u.x = 0.03f * (float)(i + 1); u.y = -0.02f * (float)(i + 3); a.x = 0.125f * (float)(threadIdx.x & 15); a.y = -0.25f * (float)(i & 7);     \
            asm volatile("" : "+v"(x), "+v"(u), "+v"(a));                                                                          \
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " mods : "=&v"(r) : "v"(x), "v"(u), "v"(a));                                 \
            float lo, hi;                                                                                                          \
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(lo) : "v"(SEL(x, l0)), "v"(SEL(u, l1)), "v"(SEL(a, l2)));              \
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(hi) : "v"(SEL(x, h0)), "v"(SEL(u, h1)), "v"(SEL(a, h2)));              \
            if (__float_as_uint(r.x) != __float_as_uint(lo)) report(out, threadIdx.x & 63, 0, rep, __float_as_uint(r.x), __float_as_uint(lo)); \
            if (__float_as_uint(r.y) != __float_as_uint(hi)) report(out, threadIdx.x & 63, 1, rep, __float_as_uint(r.y), __float_as_uint(hi)); \
        }                                                                                                                          \
}
#define PK_2OP_KERNEL(name, op, sop, mods, l0, l1, h0, h1)                                                                         \
extern "C" __global__ __launch_bounds__(256) void name(unsigned* out, int repeats) {                                              \
    for (int rep = 0; rep < repeats; ++rep)                                                                                        \
        for (int i = 0; i < 16; ++i) {                                                                                             \
            f2v x, u, r;                                                                                                           \
            x.x = 0.001f * (float)((threadIdx.x * 7 + i * 13 + rep) & 1023) - 0.5f; x.y = 0.002f * (float)((threadIdx.x * 5 + i * 11 + rep) & 511) - 0.5f; \
            u.x = 0.03f * (float)(i + 1); u.y = -0.02f * (float)(i + 3);                                                           \
            asm volatile("" : "+v"(x), "+v"(u));                                                                                   \
            asm volatile(op " %0, %1, %2 " mods : "=&v"(r) : "v"(x), "v"(u));                                                      \
            float lo, hi;                                                                                                          \
            asm volatile(sop " %0, %1, %2" : "=&v"(lo) : "v"(SEL(x, l0)), "v"(SEL(u, l1)));                                        \
            asm volatile(sop " %0, %1, %2" : "=&v"(hi) : "v"(SEL(x, h0)), "v"(SEL(u, h1)));                                        \
            if (__float_as_uint(r.x) != __float_as_uint(lo)) report(out, threadIdx.x & 63, 0, rep, __float_as_uint(r.x), __float_as_uint(lo)); \
            if (__float_as_uint(r.y) != __float_as_uint(hi)) report(out, threadIdx.x & 63, 1, rep, __float_as_uint(r.y), __float_as_uint(hi)); \
        }                                                                                                                          \
}
PK_FMA_KERNEL(pkf_plain,   "",                      0, 0, 0, 1, 1, 1)
PK_FMA_KERNEL(pkf_s100,    "op_sel:[1,0,0]",        1, 0, 0, 1, 1, 1)
PK_FMA_KERNEL(pkf_s010,    "op_sel:[0,1,0]",        0, 1, 0, 1, 1, 1)
PK_FMA_KERNEL(pkf_s001,    "op_sel:[0,0,1]",        0, 0, 1, 1, 1, 1)
PK_FMA_KERNEL(pkf_h011,    "op_sel_hi:[0,1,1]",     0, 0, 0, 0, 1, 1)
PK_FMA_KERNEL(pkf_h101,    "op_sel_hi:[1,0,1]",     0, 0, 0, 1, 0, 1)
PK_FMA_KERNEL(pkf_h110,    "op_sel_hi:[1,1,0]",     0, 0, 0, 1, 1, 0)
PK_FMA_KERNEL(pkf_h010,    "op_sel_hi:[0,1,0]",     0, 0, 0, 0, 1, 0)
PK_2OP_KERNEL(pkm_plain, "v_pk_mul_f32", "v_mul_f32", "",                 0, 0, 1, 1)
PK_2OP_KERNEL(pkm_s10,   "v_pk_mul_f32", "v_mul_f32", "op_sel:[1,0]",     1, 0, 1, 1)
PK_2OP_KERNEL(pkm_s01,   "v_pk_mul_f32", "v_mul_f32", "op_sel:[0,1]",     0, 1, 1, 1)
PK_2OP_KERNEL(pkm_h01,   "v_pk_mul_f32", "v_mul_f32", "op_sel_hi:[0,1]",  0, 0, 0, 1)
PK_2OP_KERNEL(pkm_h10,   "v_pk_mul_f32", "v_mul_f32", "op_sel_hi:[1,0]",  0, 0, 1, 0)
PK_2OP_KERNEL(pka_plain, "v_pk_add_f32", "v_add_f32", "",                 0, 0, 1, 1)
PK_2OP_KERNEL(pka_s10,   "v_pk_add_f32", "v_add_f32", "op_sel:[1,0]",     1, 0, 1, 1)
PK_2OP_KERNEL(pka_s01,   "v_pk_add_f32", "v_add_f32", "op_sel:[0,1]",     0, 1, 1, 1)
PK_2OP_KERNEL(pka_h01,   "v_pk_add_f32", "v_add_f32", "op_sel_hi:[0,1]",  0, 0, 0, 1)
PK_2OP_KERNEL(pka_h10,   "v_pk_add_f32", "v_add_f32", "op_sel_hi:[1,0]",  0, 0, 1, 0)

// v_pk_mov_b32 (as printed by the disassembler: op_sel:[a,b] gives dst = {src0[a], src1[b]}) and v_fma_mix_f32 (f16 halves of 32-bit registers as fp32 operands), the other
// instructions in liblp_hip.so that carry an op_sel
#define PK_MOV_KERNEL(name, mods, l0, h1)                                                                                          \
extern "C" __global__ __launch_bounds__(256) void name(unsigned* out, int repeats) {                                              \
    for (int rep = 0; rep < repeats; ++rep)                                                                                        \
        for (int i = 0; i < 16; ++i) {                                                                                             \
            f2v x, u, r;                                                                                                           \
            x.x = (float)((threadIdx.x * 7 + i * 13 + rep) & 1023); x.y = -(float)((threadIdx.x * 5 + i * 11 + rep) & 511) - 1.f;   \
            u.x = 4096.f + (float)(i + threadIdx.x); u.y = -8192.f - (float)(i + 3 * threadIdx.x);                                 \
            asm volatile("" : "+v"(x), "+v"(u));                                                                                   \
            asm volatile("v_pk_mov_b32 %0, %1, %2 " mods : "=&v"(r) : "v"(x), "v"(u));                                             \
            const float lo = SEL(x, l0), hi = SEL(u, h1);                                                                          \
            if (__float_as_uint(r.x) != __float_as_uint(lo)) report(out, threadIdx.x & 63, 0, rep, __float_as_uint(r.x), __float_as_uint(lo)); \
            if (__float_as_uint(r.y) != __float_as_uint(hi)) report(out, threadIdx.x & 63, 1, rep, __float_as_uint(r.y), __float_as_uint(hi)); \
        }                                                                                                                          \
}
PK_MOV_KERNEL(pkmov_plain, "op_sel:[0,0]",                    0, 0)
PK_MOV_KERNEL(pkmov_s10,   "op_sel:[1,0]",                    1, 0)
PK_MOV_KERNEL(pkmov_s01,   "op_sel:[0,1]",                    0, 1)
PK_MOV_KERNEL(pkmov_h10,   "op_sel:[1,1]",                    1, 1)
// v_fma_mix_f32 d, a, b, c with a = the HIGH (op_sel:[1,0,0]) or b = the HIGH (op_sel:[0,1,0]) f16 half of a register, checked against v_cvt_f32_f16 + v_fma_f32
#define MIX_KERNEL(name, mods, a_hi, b_hi)                                                                                         \
extern "C" __global__ __launch_bounds__(256) void name(unsigned* out, int repeats) {                                              \
    for (int rep = 0; rep < repeats; ++rep)                                                                                        \
        for (int i = 0; i < 16; ++i) {                                                                                             \
            unsigned ha = 0x3C003800u + ((threadIdx.x * 7u + i * 13u + rep) & 0x3FFu) * 0x10001u;                                  \
            unsigned hb = 0xB8003400u + ((threadIdx.x * 5u + i * 11u + rep) & 0x1FFu) * 0x10003u;                                  \
            float c = 0.125f * (float)(threadIdx.x & 15), r;                                                                       \
            asm volatile("" : "+v"(ha), "+v"(hb), "+v"(c));                                                                        \
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 " mods : "=&v"(r) : "v"(ha), "v"(hb), "v"(c));                              \
            const _Float16 fa = __builtin_bit_cast(_Float16, (unsigned short)((a_hi) ? ha >> 16 : ha & 0xFFFFu));                  \
            const _Float16 fb = __builtin_bit_cast(_Float16, (unsigned short)((b_hi) ? hb >> 16 : hb & 0xFFFFu));                  \
            float want;                                                                                                            \
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(want) : "v"((float)fa), "v"((float)fb), "v"(c));                       \
            if (__float_as_uint(r) != __float_as_uint(want)) report(out, threadIdx.x & 63, 0, rep, __float_as_uint(r), __float_as_uint(want)); \
        }                                                                                                                          \
}
MIX_KERNEL(mix_s100, "op_sel:[1,0,0] op_sel_hi:[1,1,0]", 1, 0)
MIX_KERNEL(mix_s010, "op_sel:[0,1,0] op_sel_hi:[1,1,0]", 0, 1)
MIX_KERNEL(mix_s000, "op_sel_hi:[1,1,0]", 0, 0)
typedef void (*pk_kernel_t)(unsigned*, int);
extern "C" int canary_form_launch(int form, unsigned* out, int blocks, int repeats, void* stream) {
    static const pk_kernel_t k[] = {pkf_plain, pkf_s100, pkf_s010, pkf_s001, pkf_h011, pkf_h101, pkf_h110, pkf_h010, pkm_plain, pkm_s10, pkm_s01, pkm_h01, pkm_h10,
                                    pka_plain, pka_s10, pka_s01, pka_h01, pka_h10,
                                    pkmov_plain, pkmov_s10, pkmov_s01, pkmov_h10, mix_s100, mix_s010, mix_s000};
    if (form < 0 || form >= (int)(sizeof(k) / sizeof(k[0]))) return -1;
    hipLaunchKernelGGL(k[form], dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, repeats);
    return (int)hipGetLastError();
}

extern "C" int canary_launch(int which, const void* w, int rows, int C, unsigned* out, int blocks, int repeats, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (which == 0) hipLaunchKernelGGL(load_canary, dim3((rows + 31) / 32 * 1, 1), dim3(256), 0, st, (const uint4*)w, rows, C, out, repeats);
    else if (which == 1) hipLaunchKernelGGL(lds_canary, dim3(blocks), dim3(256), 0, st, out, repeats);
    else if (which == 2) hipLaunchKernelGGL(reg_canary, dim3(blocks), dim3(256), 0, st, out, repeats);
    else hipLaunchKernelGGL(fma_canary, dim3(blocks), dim3(256), 0, st, out, repeats);
    return (int)hipGetLastError();
}
