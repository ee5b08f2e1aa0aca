// Hardware layout probe for gfx950 (test infrastructure, not product code).
// Verifies the MFMA fragment layouts and dumps the ds_read_b64_tr_b16 lane/element mapping that the
// conv kernels in latent_pose_reenactment_amd/csrc rely on.  Build: hipcc --offload-arch=gfx950 -O2 probe_layout.hip -o probe_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(2);} } while (0)

// A: [16][32] row-major bf16, B: [32][16] row-major bf16, C: [16][16] f32
__global__ void k_mfma16(const uint16_t* A, const uint16_t* B, float* C) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 15) * 32 + (l >> 4) * 8 + j];
        b[j] = (short)B[((l >> 4) * 8 + j) * 16 + (l & 15)];
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

// A: [32][16], B: [16][32], C: [32][32]
__global__ void k_mfma32(const uint16_t* A, const uint16_t* B, float* C) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 31) * 16 + (l >> 5) * 8 + j];
        b[j] = (short)B[((l >> 5) * 8 + j) * 32 + (l & 31)];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

// ds_read_b64_tr_b16: LDS u16[i] = i; lane l reads at byte offset addr[l]; out[l*4+j] = element j.
__global__ void k_trread(const int* addr, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[16384];
    int l = threadIdx.x;
    for (int i = l; i < 16384; i += 64) lds[i] = (short)i;
    __syncthreads();
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr_t;
    lds_ptr_t p = (lds_ptr_t)((__attribute__((address_space(3))) char*)lds + addr[l]);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

// Transposed-operand GEMM check: T1 = [32 k][16 m] row-major, T2 = [32 k][16 n] row-major (k = "pixel" major),
// C[m][n] = sum_k T1[k][m] * T2[k][n], fragments fetched with two tr-reads each using the address rule
// addr(lane, half) = ((half*16 + 4*(l>>4) + (l&3)) * 16 + 4*((l&15)>>2)) * 2 bytes  -- CANDIDATE rule, verified below.
__global__ void k_trgemm(const uint16_t* T1, const uint16_t* T2, float* C, int rule) {
    __shared__ __attribute__((aligned(16))) short lds[2 * 32 * 16];
    int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) { lds[i] = (short)T1[i]; lds[512 + i] = (short)T2[i]; }
    __syncthreads();
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr_t;
    __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)lds;
    bf16x8 a, b;
    for (int half = 0; half < 2; ++half) {
        int off;
        if (rule == 0) off = (half * 256 + l * 4) * 2;                                   // linear: lane*8 bytes
        else           off = ((half * 16 + 4 * (l >> 4) + (l & 3)) * 16 + 4 * ((l & 15) >> 2)) * 2;
        s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(base + off));
        s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(base + 1024 + off));
        for (int j = 0; j < 4; ++j) { a[half * 4 + j] = va[j]; b[half * 4 + j] = vb[j]; }
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  arch=%s  CUs=%d  clock=%d kHz  LDS/block=%zu\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock);
    srand(7);
    // ---- MFMA 16x16x32
    {
        std::vector<uint16_t> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16), C(256), R(256, 0.f);
        for (int i = 0; i < 512; ++i) { Af[i] = (float)(rand() % 17 - 8); A[i] = f2bf(Af[i]); Bf[i] = (float)(rand() % 13 - 6) * 0.5f; B[i] = f2bf(Bf[i]); }
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int k = 0; k < 32; ++k) R[m * 16 + n] += Af[m * 32 + k] * Bf[k * 16 + n];
        uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
        double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(C[i] - R[i]));
        printf("MFMA16x16x32_bf16 layout: %s (max err %g)\n", e < 1e-3 ? "PASS" : "FAIL", e);
    }
    // ---- MFMA 32x32x16
    {
        std::vector<uint16_t> A(512), B(512); std::vector<float> Af(512), Bf(512), C(1024), R(1024, 0.f);
        for (int i = 0; i < 512; ++i) { Af[i] = (float)(rand() % 17 - 8); A[i] = f2bf(Af[i]); Bf[i] = (float)(rand() % 13 - 6) * 0.5f; B[i] = f2bf(Bf[i]); }
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[m * 32 + n] += Af[m * 16 + k] * Bf[k * 32 + n];
        uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(C[i] - R[i]));
        printf("MFMA32x32x16_bf16 layout: %s (max err %g)\n", e < 1e-3 ? "PASS" : "FAIL", e);
    }
    // ---- tr-read dumps
    {
        int* dAddr; short* dOut; CK(hipMalloc(&dAddr, 256)); CK(hipMalloc(&dOut, 512));
        const char* names[4] = {"linear lane*8B", "lane*64B", "rows16: ((4*(l>>4)+(l&3))*16+4*((l&15)>>2))*2B", "row=l&15 stride 80B, col4=(l>>4)*8B"};
        for (int pat = 0; pat < 4; ++pat) {
            int addr[64];
            for (int l = 0; l < 64; ++l) {
                if (pat == 0) addr[l] = l * 8;
                else if (pat == 1) addr[l] = l * 64;
                else if (pat == 2) addr[l] = ((4 * (l >> 4) + (l & 3)) * 16 + 4 * ((l & 15) >> 2)) * 2;
                else addr[l] = (l & 15) * 80 + (l >> 4) * 8;
            }
            CK(hipMemcpy(dAddr, addr, 256, hipMemcpyHostToDevice));
            k_trread<<<1, 64>>>(dAddr, dOut); CK(hipDeviceSynchronize());
            short out[256]; CK(hipMemcpy(out, dOut, 512, hipMemcpyDeviceToHost));
            printf("TRREAD pattern %d (%s): per lane [e0 e1 e2 e3] = u16 index read; (srcLane.srcElem) decoded from addresses\n", pat, names[pat]);
            for (int l = 0; l < 64; ++l) {
                printf("  l%02d a=%4d:", l, addr[l]);
                for (int j = 0; j < 4; ++j) {
                    int idx = out[l * 4 + j]; int sl = -1, se = -1;
                    for (int q = 0; q < 64; ++q) if (idx * 2 >= addr[q] && idx * 2 < addr[q] + 8) { sl = q; se = (idx * 2 - addr[q]) / 2; }
                    printf(" %5d(%02d.%d)", idx, sl, se);
                }
                printf("%s", (l & 1) ? "\n" : "   |");
            }
        }
    }
    // ---- transposed-operand GEMM through tr-read
    for (int rule = 0; rule < 2; ++rule) {
        std::vector<uint16_t> T1(512), T2(512); std::vector<float> F1(512), F2(512), C(256), R(256, 0.f);
        for (int i = 0; i < 512; ++i) { F1[i] = (float)(rand() % 17 - 8); T1[i] = f2bf(F1[i]); F2[i] = (float)(rand() % 13 - 6) * 0.5f; T2[i] = f2bf(F2[i]); }
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int k = 0; k < 32; ++k) R[m * 16 + n] += F1[k * 16 + m] * F2[k * 16 + n];
        uint16_t *d1, *d2; float* dC; CK(hipMalloc(&d1, 1024)); CK(hipMalloc(&d2, 1024)); CK(hipMalloc(&dC, 1024));
        CK(hipMemcpy(d1, T1.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, T2.data(), 1024, hipMemcpyHostToDevice));
        k_trgemm<<<1, 64>>>(d1, d2, dC, rule); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
        double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(C[i] - R[i]));
        printf("TR-GEMM rule %d: %s (max err %g)\n", rule, e < 1e-3 ? "PASS" : "FAIL", e);
    }
    return 0;
}
