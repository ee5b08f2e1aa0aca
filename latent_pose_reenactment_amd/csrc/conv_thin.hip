// Weight gradient of the convs with a THIN side (<= 4 channels) for gfx950: the image-side convs of the critics and the VGG
// stacks (Cin = 3 -> 64) and the generator head (64 -> 4).  On the MFMA wgrad kernel these pad the thin side to 64 channels
// (21x / 16x wasted matrix work plus the generic halo staging) and ran at 6-12 TFLOP/s; the work is really a bandwidth-bound
// reduction over pixels of (wide operand, 64 lanes) x (thin operand, <= 4 values x taps), so it is done in fp32 on the VALU:
//
//   dw[co][ci][tap] = sum_{n,y,x} dy[n,y,x,co] * act(x)[n, y+ky-1, x+kx-1, ci]
//
// One workgroup walks a strided set of 4-row groups.  The thin operand's rows (with the 1-pixel border, zero padded, channels
// padded to 4 = one 16-byte LDS word per pixel) are staged in LDS and read back as wave-wide BROADCASTS; the wide operand is
// streamed from HBM exactly once, 64 consecutive channels per wave (256-byte rows); wave w takes the pixels x = w, w+4, ...
//   THIN_X:  thin = input x (Cin <= 4, no prologue),  wide = dy;       acc[tap][ci] += dy[p][co] * x[p + tap][ci]
//   !THIN_X: thin = dy (Cout <= 4),  wide = act(x) (AdaIN/ReLU prologue applied on the fly);
//                                                                       acc[tap][co] += act(x)[q][ci] * dy[q - tap][co]
// Partial sums per workgroup go to the workspace and are reduced by wgrad_thin_reduce_kernel (deterministic, no atomics).
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

struct ThinParams {
    const float* x; const float* dy; float* part; float* dw;
    const uint16_t* x16;             // !THIN_X only: the wide operand as a 16-bit-resident fp16 plane [N][H][W][Cin] instead of fp32 x (NULL: fp32)
    float* bpart; float* dbias;      // NULL | [G][Cout] partial column sums of dy (THIN_X only) and the bias gradient they reduce to
    const float* scale; const float* shift;
    int N, H, W, Cin, Cout, pro, G;
};

// C4: the thin side really has 4 channels (else <= 3: the fourth FMA per tap is skipped)
template <int KS, bool THIN_X, bool C4>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(ThinParams p) {
    constexpr int T = KS * KS, PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, ph = tid >> 6;
    const int thinC = THIN_X ? p.Cin : p.Cout, wideC = THIN_X ? p.Cout : p.Cin;
    const int wc = blockIdx.y * 64 + lane;                               // this lane's channel of the wide operand
    const float* thin = THIN_X ? p.x : p.dy;
    const float* wide = THIN_X ? p.dy : p.x;
    const int RW = p.W + 2 * PAD;                                        // staged row width (pixels)
    float bsum = 0.f;                                                    // THIN_X: column sum of dy (bias gradient)
    float acc[T][4];              // (scalar FMAs: v_pk_fma_f32 pairs measured 1.4x SLOWER here)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;

    // a block iteration = RB consecutive rows of one image (+ border rows of the thin operand); the wide operand's loads are
    // issued UB at a time before they are used, so every wave keeps UB x 256 B in flight (the kernel is a pure HBM stream)
    constexpr int RB = 4, UB = (KS == 1) ? 16 : 8;
    const int gpi = (p.H + RB - 1) / RB;                                 // row groups per image
    for (int grp = blockIdx.x; grp < p.N * gpi; grp += gridDim.x) {
        const int n = grp / gpi, y0 = (grp % gpi) * RB;
        __syncthreads();                                                 // previous group fully consumed
        for (int i = tid; i < (RB + 2 * PAD) * RW; i += 256) {
            const int r = i / RW, hx = i % RW;
            const int iy = y0 + r - PAD, ix = hx - PAD;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                const float* src = thin + ((size_t)(n * p.H + iy) * p.W + ix) * thinC;
                v.x = src[0];
                if (thinC > 1) v.y = src[1];
                if (thinC > 2) v.z = src[2];
                if (thinC > 3) v.w = src[3];
            }
            *(float4*)(sm + (size_t)i * 4) = v;
        }
        __syncthreads();
        float sc = 1.f, sh = 0.f;
        if (!THIN_X && p.pro == 1) { sc = p.scale[(size_t)n * p.Cin + wc]; sh = p.shift[(size_t)n * p.Cin + wc]; }
        const int nrow = min(RB, p.H - y0);
        const int npx = nrow * p.W;                                      // pixels of the group, row-major; this wave: ph, ph+4, ...
        const float* wbase = wide + (size_t)(n * p.H + y0) * p.W * wideC + wc;
        const uint16_t* wbase16 = (!THIN_X && p.x16) ? p.x16 + (size_t)(n * p.H + y0) * p.W * wideC + wc : nullptr;
        for (int j0 = ph; j0 < npx; j0 += 4 * UB) {
            float av[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int j = j0 + 4 * u;
                if (!THIN_X && wbase16) av[u] = lp_op16_to_f32<true>(wbase16[(size_t)(j < npx ? j : 0) * wideC]);
                else av[u] = wbase[(size_t)(j < npx ? j : 0) * wideC];          // unconditional load, masked below
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int j = j0 + 4 * u;
                float a = av[u];
                if (!THIN_X && p.pro != 0) a = fmaxf(fmaf(a, sc, sh), 0.f);
                a = j < npx ? a : 0.f;
                if (THIN_X) bsum += a;
                const int rr = j / p.W, xx = j - rr * p.W;
                const int jr = j < npx ? rr : 0;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        // THIN_X: x pixel (y+ky-1, xx+kx-1) -> staged row rr+ky, column xx+kx.   else: dy pixel (y-ky+1, xx-kx+1)
                        // -> staged row rr+2-ky (rows y0-1..), column xx-kx+2
                        const int r = jr + (THIN_X ? ky : (KS - 1 - ky)), c = THIN_X ? (xx + kx) : (xx + KS - 1 - kx);
                        const float4 v = *(const float4*)(sm + ((size_t)r * RW + c) * 4);
                        acc[ky * KS + kx][0] = fmaf(a, v.x, acc[ky * KS + kx][0]);
                        acc[ky * KS + kx][1] = fmaf(a, v.y, acc[ky * KS + kx][1]);
                        acc[ky * KS + kx][2] = fmaf(a, v.z, acc[ky * KS + kx][2]);
                        if (C4) acc[ky * KS + kx][3] = fmaf(a, v.w, acc[ky * KS + kx][3]);
                    }
            }
        }
    }
    // sum the four pixel phases (waves) through LDS, then one partial slab per workgroup: part[wg][tap][thin c][wide c]
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) sm[((ph * T + t) * 4 + c) * 64 + lane] = acc[t][c];
    __syncthreads();
    for (int i = tid; i < T * 4 * 64; i += 256) {
        const float s = (sm[i] + sm[T * 256 + i]) + (sm[2 * T * 256 + i] + sm[3 * T * 256 + i]);
        const int l = i & 63, c = (i >> 6) & 3, t = i >> 8;
        if (c < thinC) p.part[(((size_t)blockIdx.x * T + t) * thinC + c) * wideC + blockIdx.y * 64 + l] = s;
    }
    if (THIN_X && p.bpart) {
        __syncthreads();
        sm[tid] = bsum;
        __syncthreads();
        if (tid < 64) p.bpart[(size_t)blockIdx.x * wideC + blockIdx.y * 64 + tid] = (sm[tid] + sm[64 + tid]) + (sm[128 + tid] + sm[192 + tid]);
    }
}

// dw[co][ci][tap] = sum_wg part[wg][tap][thin][wide]; a block = 16 outputs x 16 slices of the G workgroup slabs (4 loads in
// flight per thread), combined through LDS; trailing blocks do the same for the bias partials
template <bool THIN_X>
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(ThinParams p, int T, int wblocks) {
    __shared__ float red[16][17];
    const int thinC = THIN_X ? p.Cin : p.Cout, wideC = THIN_X ? p.Cout : p.Cin;
    const bool is_bias = (int)blockIdx.x >= wblocks;
    const int total = is_bias ? wideC : T * thinC * wideC;
    const float* src = is_bias ? p.bpart : p.part;
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int idx = (is_bias ? (int)blockIdx.x - wblocks : (int)blockIdx.x) * 16 + o;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (idx < total) {
        int g = sl;
        for (; g + 48 < p.G; g += 64) {
            a0 += src[(size_t)g * total + idx]; a1 += src[(size_t)(g + 16) * total + idx];
            a2 += src[(size_t)(g + 32) * total + idx]; a3 += src[(size_t)(g + 48) * total + idx];
        }
        for (; g < p.G; g += 16) a0 += src[(size_t)g * total + idx];
    }
    red[sl][o] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && idx < total) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][o];
        if (is_bias) { p.dbias[idx] = s; return; }
        const int w = idx % wideC, c = (idx / wideC) % thinC, t = idx / (wideC * thinC);
        const int co = THIN_X ? w : c, ci = THIN_X ? c : w;
        p.dw[((size_t)co * p.Cin + ci) * T + t] = s;
    }
}

bool lp_wgrad_thin_supported(int Cin, int Cout, int ksize, int upsample, int pro, int W) {
    if (upsample || (ksize != 1 && ksize != 3)) return false;
    if ((size_t)(4 + 2 * (ksize / 2)) * (W + 2 * (ksize / 2)) * 16 > 64 * 1024) return false;      // staged rows must fit LDS
    if (Cin <= 4 && Cout % 64 == 0 && pro == 0) return true;
    if (Cout <= 4 && Cin % 64 == 0 && ksize == 3) return true;
    return false;
}

// workspace: the caller's lp_conv_wgrad workspace (splits * T * 64 * 64k floats) holds G <= 16 * splits slabs of T * 4 * wide
static int wgrad_thin_impl(const float* x, const uint16_t* x16, const float* dy, float* dw, float* workspace, const float* scale, const float* shift,
                           int N, int H, int W, int Cin, int Cout, int ksize, int pro, int splits, float* dbias, hipStream_t stream);

int lp_wgrad_thin(const float* x, const float* dy, float* dw, float* workspace, const float* scale, const float* shift, int N, int H,
                  int W, int Cin, int Cout, int ksize, int pro, int splits, float* dbias, hipStream_t stream) {
    return wgrad_thin_impl(x, nullptr, dy, dw, workspace, scale, shift, N, H, W, Cin, Cout, ksize, pro, splits, dbias, stream);
}

static int wgrad_thin_impl(const float* x, const uint16_t* x16, const float* dy, float* dw, float* workspace, const float* scale, const float* shift,
                           int N, int H, int W, int Cin, int Cout, int ksize, int pro, int splits, float* dbias, hipStream_t stream) {
    const bool thin_x = (Cin <= 4 && Cout % 64 == 0 && pro == 0);
    ThinParams p;
    p.x16 = thin_x ? nullptr : x16;
    p.x = x; p.dy = dy; p.part = workspace; p.dw = dw; p.scale = scale; p.shift = shift;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.pro = pro;
    int G = N * ((H + 3) / 4);                  // row groups (RB = 4 rows each)
    if (G > 16 * splits) G = 16 * splits;
    if (G > 2048) G = 2048;
    const bool want_b = thin_x && dbias != nullptr;
    if (want_b && G > splits) G = splits;          // the [G][Cout] bias partials live in the workspace's [splits][CoP] tail
    p.G = G;
    p.bpart = want_b ? workspace + (size_t)splits * ksize * ksize * ((Cout + 63) / 64 * 64) * ((Cin + 63) / 64 * 64) : nullptr;
    p.dbias = dbias;
    const int T = ksize * ksize, wide = thin_x ? Cout : Cin, thinC = thin_x ? Cin : Cout;
    const size_t stage = (size_t)(4 + 2 * (ksize / 2)) * (W + 2 * (ksize / 2)) * 16, red = (size_t)4 * T * 4 * 64 * 4;
    const size_t lds = stage > red ? stage : red;
    if (lds > 64 * 1024) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_conv_wgrad (thin): image row too wide for LDS");
    dim3 grid(G, wide / 64);
    const bool c4 = thinC == 4;
    if (thin_x) {
        if (ksize == 3) { if (c4) hipLaunchKernelGGL((wgrad_thin_kernel<3, true, true>), grid, dim3(256), lds, stream, p);
                          else hipLaunchKernelGGL((wgrad_thin_kernel<3, true, false>), grid, dim3(256), lds, stream, p); }
        else { if (c4) hipLaunchKernelGGL((wgrad_thin_kernel<1, true, true>), grid, dim3(256), lds, stream, p);
               else hipLaunchKernelGGL((wgrad_thin_kernel<1, true, false>), grid, dim3(256), lds, stream, p); }
    } else {
        if (c4) hipLaunchKernelGGL((wgrad_thin_kernel<3, false, true>), grid, dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((wgrad_thin_kernel<3, false, false>), grid, dim3(256), lds, stream, p);
    }
    int rc = lp_check_launch("wgrad_thin");
    if (rc) return rc;
    const int total = T * thinC * wide;
    const int wblocks = (total + 15) / 16, bblocks = p.bpart ? (wide + 15) / 16 : 0;
    if (thin_x) hipLaunchKernelGGL(wgrad_thin_reduce_kernel<true>, dim3(wblocks + bblocks), dim3(256), 0, stream, p, T, wblocks);
    else hipLaunchKernelGGL(wgrad_thin_reduce_kernel<false>, dim3(wblocks + bblocks), dim3(256), 0, stream, p, T, wblocks);
    return lp_check_launch("wgrad_thin_reduce");
}

// ------------------------------------------------------------------------------------------------------------------
// Forward conv with <= 4 input channels (RGB -> 64: first conv of the critics and of both VGG stacks).  27 MACs per output:
// the MFMA kernel pads K to 9 x 32 and stages a 32-channel halo for 3 real channels.  Here the image rows are staged in LDS
// (one 16-byte word per pixel), every lane owns one output channel with its 9 x 4 weights in registers (decoded from the
// bf16 (hi [+ lo]) pack, i.e. the same weight values the MFMA path would use), pixels are broadcast from LDS, and the output
// row is written 256 bytes per wave-store.  fp32 activations are used as they are (more exact than the bf16 operand path).
// ------------------------------------------------------------------------------------------------------------------
struct ThinFwdParams {
    const float* x; const uint16_t* w_hi; const uint16_t* w_lo; float* y; const float* bias; const float* alpha;
    int N, H, W, Cin, Cout, CinP, CoutP, f16;
};

template <int KS, bool C4>
__global__ __launch_bounds__(256) void conv_thin_fwd_kernel(ThinFwdParams p) {
    constexpr int T = KS * KS, PAD = KS / 2, RB = 4, UB = 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, ph = tid >> 6;
    const int co = blockIdx.y * 64 + lane;
    const int RW = p.W + 2 * PAD;
    float w[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const size_t idx = ((size_t)t * p.CoutP + co) * p.CinP + c;         // pack layout [tap][CoutP][CinP], zero padded
            float v = p.f16 ? lp_op16_to_f32<true>(p.w_hi[idx]) : lp_op16_to_f32<false>(p.w_hi[idx]);
            if (p.w_lo) v += lp_op16_to_f32<false>(p.w_lo[idx]);
            w[t][c] = v;
        }
    const float alpha = p.alpha ? *p.alpha : 1.f;
    const float bias = p.bias ? p.bias[co] : 0.f;
    const int gpi = (p.H + RB - 1) / RB;
    for (int grp = blockIdx.x; grp < p.N * gpi; grp += gridDim.x) {
        const int n = grp / gpi, y0 = (grp % gpi) * RB;
        __syncthreads();
        for (int i = tid; i < (RB + 2 * PAD) * RW; i += 256) {
            const int r = i / RW, hx = i % RW;
            const int iy = y0 + r - PAD, ix = hx - PAD;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                const float* src = p.x + ((size_t)(n * p.H + iy) * p.W + ix) * p.Cin;
                v.x = src[0];
                if (p.Cin > 1) v.y = src[1];
                if (p.Cin > 2) v.z = src[2];
                if (p.Cin > 3) v.w = src[3];
            }
            *(float4*)(sm + (size_t)i * 4) = v;
        }
        __syncthreads();
        const int nrow = min(RB, p.H - y0), npx = nrow * p.W;
        float* ybase = p.y + (size_t)(n * p.H + y0) * p.W * p.Cout + co;
        for (int j0 = ph; j0 < npx; j0 += 4 * UB) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int j = j0 + 4 * u;
                if (j < npx) {
                    const int rr = j / p.W, xx = j - rr * p.W;
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            const float4 v = *(const float4*)(sm + ((size_t)(rr + ky) * RW + xx + kx) * 4);
                            const int t = ky * KS + kx;
                            a0 = fmaf(v.x, w[t][0], a0); a1 = fmaf(v.y, w[t][1], a1);
                            a0 = fmaf(v.z, w[t][2], a0); if (C4) a1 = fmaf(v.w, w[t][3], a1);
                        }
                    ybase[(size_t)j * p.Cout] = fmaf(a0 + a1, alpha, bias);
                }
            }
        }
    }
}

// RGB -> 64 3x3 on the matrix cores (fp16 / bf16 operand modes): the whole contraction is 27 MACs per output = ONE 16x16x32 MFMA
// k-step.  k = 3 * tap + channel (27 used, 5 zero); a wave takes 16 consecutive pixels of an image row as the M side and 64 output
// channels as 4 N tiles.  A fragment: 8 u16 LDS reads per lane from the fp16/bf16 halo rows (one 8-byte word per pixel); B fragments:
// gathered once per workgroup from the 16-bit weight pack.  The fp32 VALU kernel above spends 27 FMAs per output per lane (31 us of
// pure VALU time on a 8 x 256 x 256 image against 25 us of output writes); this one is bound by the writes.  The 16 x 64 block goes
// through LDS so that a pixel's 64 channels leave as one 256-byte run, together with the optional operand planes of (relu?)(y)
// for the next conv (saves that layer's lp_act_pack pass over y).
struct RgbMfmaParams {
    const float* x; const uint16_t* w_hi; float* y; const float* bias; const float* alpha; uint16_t* o_hi;
    int N, H, W, Cout, CinP, CoutP, o_relu;
};

template <bool F16>
__global__ __launch_bounds__(256) void conv_rgb_mfma_kernel(RgbMfmaParams p) {
    constexpr int RB = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int RW = p.W + 2;
    uint16_t* halo = (uint16_t*)smraw;                                             // [RB + 2][RW][4] 16-bit
    float* tbase = (float*)(smraw + (((size_t)(RB + 2) * RW * 8 + 15) & ~(size_t)15));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* T = tbase + wave * (16 * 68);                                            // this wave's 16 x 64 output block (+4 pad)
    const int co0 = blockIdx.y * 64;
    const int kg = lane >> 4, m = lane & 15;
    // per-lane k slice kk = kg*8 + j -> (tap, channel) -> halo offset (in 16-bit units) relative to (row r, pixel x0 + m)
    int aoff[8];
    s16x8_t bfrag[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = kg * 8 + j, t = kk / 3, c = kk - t * 3;
        aoff[j] = kk < 27 ? ((t / 3) * RW + (t % 3)) * 4 + c : -1;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int co = co0 + nt * 16 + m;
            bfrag[nt][j] = kk < 27 ? (short)p.w_hi[((size_t)t * p.CoutP + co) * p.CinP + c] : (short)0;
        }
    }
    const float alpha = p.alpha ? *p.alpha : 1.f;
    const int pq = lane >> 2, cq = (lane & 3) * 16;                                  // store phase: pixel of the block, first of 16 channels
    float bias16[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bias16[j] = p.bias ? p.bias[co0 + cq + j] : 0.f;
    const int gpi = (p.H + RB - 1) / RB, segs = p.W >> 4;
    for (int grp = blockIdx.x; grp < p.N * gpi; grp += gridDim.x) {
        const int n = grp / gpi, y0 = (grp % gpi) * RB;
        __syncthreads();
        for (int i = tid; i < (RB + 2) * RW; i += 256) {
            const int r = i / RW, hx = i - r * RW;
            const int iy = y0 + r - 1, ix = hx - 1;
            ushort4 v = make_ushort4(0, 0, 0, 0);
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                const float* src = p.x + ((size_t)(n * p.H + iy) * p.W + ix) * 3;
                v.x = lp_f32_to_op16<F16>(src[0]); v.y = lp_f32_to_op16<F16>(src[1]); v.z = lp_f32_to_op16<F16>(src[2]);
            }
            *(ushort4*)(halo + (size_t)i * 4) = v;
        }
        __syncthreads();
        const int nrow = min(RB, p.H - y0);
        for (int sgi = wave; sgi < nrow * segs; sgi += 4) {
            const int r = sgi / segs, x0 = (sgi - r * segs) << 4;
            const uint16_t* hb = halo + ((size_t)r * RW + x0 + m) * 4;
            s16x8_t a;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = aoff[j] >= 0 ? (short)hb[aoff[j]] : (short)0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4_t acc = mfma16t<F16>(a, bfrag[nt], (f32x4_t){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int q = 0; q < 4; ++q) T[(kg * 4 + q) * 68 + nt * 16 + m] = acc[q];      // C: row = (lane>>4)*4 + q (pixel), col = lane&15
            }
            // (wave-private block: the LDS writes above are ordered before the reads below by the compiler's lgkmcnt wait)
            const size_t pix = (size_t)(n * p.H + y0 + r) * p.W + x0 + pq;
            float* dst = p.y + pix * p.Cout + co0 + cq;
            float o[16];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const float4 v = *(const float4*)(T + pq * 68 + cq + j4 * 4);
                o[j4 * 4 + 0] = fmaf(v.x, alpha, bias16[j4 * 4 + 0]); o[j4 * 4 + 1] = fmaf(v.y, alpha, bias16[j4 * 4 + 1]);
                o[j4 * 4 + 2] = fmaf(v.z, alpha, bias16[j4 * 4 + 2]); o[j4 * 4 + 3] = fmaf(v.w, alpha, bias16[j4 * 4 + 3]);
                if (p.y) *(float4*)(dst + j4 * 4) = make_float4(o[j4 * 4], o[j4 * 4 + 1], o[j4 * 4 + 2], o[j4 * 4 + 3]);
            }
            if (p.o_hi) {
                s16x8_t h0, h1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    h0[j] = (short)lp_f32_to_op16<F16>(p.o_relu ? fmaxf(o[j], 0.f) : o[j]);
                    h1[j] = (short)lp_f32_to_op16<F16>(p.o_relu ? fmaxf(o[8 + j], 0.f) : o[8 + j]);
                }
                uint16_t* od = p.o_hi + pix * p.Cout + co0 + cq;
                *(s16x8_t*)od = h0; *(s16x8_t*)(od + 8) = h1;
            }
        }
    }
}

static bool rgb_mfma_ok(int Cin, int Cout, int ksize, int W, int prec) {
    static const int env = getenv("LP_THIN_MFMA") ? atoi(getenv("LP_THIN_MFMA")) : 1;       // LP_THIN_MFMA=0: fp32 VALU kernel always
    return env && Cin == 3 && ksize == 3 && (Cout % 64) == 0 && (W % 16) == 0 && prec != LP_PREC_BF16X3 &&
           (size_t)6 * (W + 2) * 8 + 16 + 4 * 16 * 68 * 4 <= 64 * 1024;
}

bool lp_conv_thin_fwd_supported(int Cin, int Cout, int ksize, int upsample, int pro, bool has_res, int W) {
    return Cin <= 4 && (Cout % 64) == 0 && !upsample && pro == 0 && !has_res && (ksize == 1 || ksize == 3) &&
           (size_t)(4 + 2) * (W + 2) * 16 <= 64 * 1024;
}

int lp_conv_thin_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, float* y, const float* bias, const float* alpha, int N,
                     int H, int W, int Cin, int Cout, int CinP, int CoutP, int ksize, int f16, hipStream_t stream) {
    ThinFwdParams p;
    p.x = x; p.w_hi = w_hi; p.w_lo = w_lo; p.y = y; p.bias = bias; p.alpha = alpha;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.CinP = CinP; p.CoutP = CoutP; p.f16 = f16;
    int G = N * ((H + 3) / 4);
    if (G > 2048) G = 2048;
    const size_t lds = (size_t)(4 + 2 * (ksize / 2)) * (W + 2 * (ksize / 2)) * 16;
    dim3 grid(G, Cout / 64);
    if (ksize == 3) { if (Cin == 4) hipLaunchKernelGGL((conv_thin_fwd_kernel<3, true>), grid, dim3(256), lds, stream, p);
                      else hipLaunchKernelGGL((conv_thin_fwd_kernel<3, false>), grid, dim3(256), lds, stream, p); }
    else { if (Cin == 4) hipLaunchKernelGGL((conv_thin_fwd_kernel<1, true>), grid, dim3(256), lds, stream, p);
           else hipLaunchKernelGGL((conv_thin_fwd_kernel<1, false>), grid, dim3(256), lds, stream, p); }
    return lp_check_launch("conv_thin_fwd");
}

// ---- C ABI of the thin-channel kernels (fp32 activations in, no operand planes) --------------------------------------------------
extern "C" int lp_thin_conv_supported(int Cin, int Cout, int ksize, int W) {
    return lp_conv_thin_fwd_supported(Cin, Cout, ksize, 0, 0, false, W) ? 1 : 0;
}

extern "C" int lp_thin_conv_emits_planes(int Cin, int Cout, int ksize, int W, int prec) { return rgb_mfma_ok(Cin, Cout, ksize, W, prec) ? 1 : 0; }

extern "C" int lp_thin_conv_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, float* y, const float* bias, const float* alpha,
                                int N, int H, int W, int Cin, int Cout, int CinP, int CoutP, int ksize, int prec, uint16_t* out_hi,
                                int out_relu, void* stream) {
    if (!x || !w_hi) return lp_set_error(LP_ERR_ARG, "lp_thin_conv_fwd: null pointer");
    if (!lp_conv_thin_fwd_supported(Cin, Cout, ksize, 0, 0, false, W)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_thin_conv_fwd: unsupported shape");
    // y == NULL: planes-only output, where the MFMA kernel emits them (lp_thin_conv_emits_planes)
    if (!y && !(out_hi && rgb_mfma_ok(Cin, Cout, ksize, W, prec))) return lp_set_error(LP_ERR_ARG, "lp_thin_conv_fwd: y may only be NULL where the operand planes are emitted");
    if (rgb_mfma_ok(Cin, Cout, ksize, W, prec)) {
        RgbMfmaParams q;
        q.x = x; q.w_hi = w_hi; q.y = y; q.bias = bias; q.alpha = alpha; q.o_hi = out_hi; q.o_relu = out_relu;
        q.N = N; q.H = H; q.W = W; q.Cout = Cout; q.CinP = CinP; q.CoutP = CoutP;
        int G = N * ((H + 3) / 4); if (G > 2048) G = 2048;
        const size_t lds = (((size_t)6 * (W + 2) * 8 + 15) & ~(size_t)15) + (size_t)4 * 16 * 68 * sizeof(float);
        dim3 grid(G, Cout / 64);
        if (prec == LP_PREC_F16) hipLaunchKernelGGL(conv_rgb_mfma_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, q);
        else hipLaunchKernelGGL(conv_rgb_mfma_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, q);
        return lp_check_launch("conv_rgb_mfma");
    }
    if (out_hi) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_thin_conv_fwd: operand planes are only emitted where lp_thin_conv_emits_planes() says so");
    return lp_conv_thin_fwd(x, w_hi, prec == LP_PREC_BF16X3 ? w_lo : nullptr, y, bias, alpha, N, H, W, Cin, Cout, CinP, CoutP, ksize,
                            prec == LP_PREC_F16, (hipStream_t)stream);
}

extern "C" int lp_thin_wgrad_supported(int Cin, int Cout, int ksize, int pro, int W) {
    return lp_wgrad_thin_supported(Cin, Cout, ksize, 0, pro, W) ? 1 : 0;
}

extern "C" int lp_thin_wgrad_has_dbias(int Cin, int Cout) { (void)Cin; return Cout > 4; }

// lp_thin_wgrad for a conv with <= 4 OUTPUT channels (the generator head, 64 -> 4) whose wide input is a 16-bit-resident conv output:
// x16 = the unscaled fp16 plane [N][H][W][Cin] (Cin % 64 == 0), AdaIN + ReLU prologue (pro 1) applied on the fly as in the fp32 form
extern "C" int lp_thin_wgrad16(const uint16_t* x16, const float* dy, float* dw, float* workspace, const float* scale, const float* shift,
                               int N, int H, int W, int Cin, int Cout, int ksize, int pro, int splits, void* stream) {
    if (!x16 || !dy || !dw || !workspace) return lp_set_error(LP_ERR_ARG, "lp_thin_wgrad16: null pointer");
    if (pro == 1 && (!scale || !shift)) return lp_set_error(LP_ERR_ARG, "lp_thin_wgrad16: pro=1 needs scale/shift");
    if (!(Cout <= 4 && Cin % 64 == 0 && ksize == 3) || !lp_wgrad_thin_supported(Cin, Cout, ksize, 0, pro, W))
        return lp_set_error(LP_ERR_UNSUPPORTED, "lp_thin_wgrad16: needs Cout <= 4, Cin % 64 == 0, 3x3, rows that fit LDS");
    return wgrad_thin_impl(nullptr, x16, dy, dw, workspace, scale, shift, N, H, W, Cin, Cout, ksize, pro, splits, nullptr, (hipStream_t)stream);
}

extern "C" int lp_thin_wgrad(const float* x, const float* dy, float* dw, float* workspace, const float* scale, const float* shift,
                             int N, int H, int W, int Cin, int Cout, int ksize, int pro, int splits, float* dbias, void* stream) {
    if (!x || !dy || !dw || !workspace) return lp_set_error(LP_ERR_ARG, "lp_thin_wgrad: null pointer");
    if (pro == 1 && (!scale || !shift)) return lp_set_error(LP_ERR_ARG, "lp_thin_wgrad: pro=1 needs scale/shift");
    if (!lp_wgrad_thin_supported(Cin, Cout, ksize, 0, pro, W)) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_thin_wgrad: unsupported shape");
    return lp_wgrad_thin(x, dy, dw, workspace, scale, shift, N, H, W, Cin, Cout, ksize, pro, splits, dbias, (hipStream_t)stream);
}
