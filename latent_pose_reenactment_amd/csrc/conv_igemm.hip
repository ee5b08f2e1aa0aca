// Fused implicit-GEMM convolution for gfx950 (MFMA 16x16x32 bf16, fp32 accumulate).
//
//   y[n,oy,ox,co] = alpha * sum_{tap,ci} act(x)[n, oy+dy, ox+dx, ci] * w[tap][co][ci]  (+ bias[co]) (+ res[n,oy>>rs,ox>>rs,co])
//
// act() is the *prologue* of the consumer conv (pre-activation ResBlock, reference generators/common/blocks.py:70-88):
// AdaIN affine (x*scale[n,c]+shift[n,c]) -> ReLU -> optional nearest x2 upsample, all applied while the input halo tile
// is staged into LDS, so the normalised/upsampled tensor never exists in HBM.  The same kernel is the data-gradient
// kernel when fed dY and the flipped/transposed weight pack (lp_pack_weights mode 1).
//
// Tiling: one workgroup (256 threads = 4 waves) owns BM = WM*MR*16 output pixels (NB images x TH x TW patch) and
// BN = WN*NR*16 output channels.  K loop = input-channel chunks of CC; per chunk the activated halo is staged ONCE and
// reused by all KS*KS taps; the weight tile of each tap is double-buffered through registers -> LDS.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"

struct ConvParams {
    const float* x; const uint16_t* w_hi; const uint16_t* w_lo; float* y;
    const float* scale; const float* shift; const float* bias; const float* res; const float* alpha;
    int N, H, W, Hin, Win, Cin, Cout, CinP, CoutP;
    int res_shift, pro;
    int lTH, lTW, lNB, tiles_x, tiles_y;
};

template <int KS, bool UPS, int WM, int WN, int MR, int NR, int CC, int PREC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
    constexpr int T = KS * KS;
    constexpr int SA = CC * 2 + 16, SB = CC * 2 + 16;     // padded LDS row strides (bytes)
    constexpr int CG = CC / 8;
    constexpr int B_ITEMS = BN * CG;                       // 16-byte weight items per tap tile
    constexpr int B_PER_THREAD = (B_ITEMS + 255) / 256;
    static_assert(!(UPS && KS == 1), "1x1 convs commute with nearest upsampling: run them at low resolution");
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;

    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int ng = t / p.tiles_y;
    const int n0 = ng << p.lNB, y0 = ty << p.lTH, x0 = tx << p.lTW;
    const int co0 = blockIdx.y * BN;

    // halo geometry (input coordinates)
    int HH, HW, oy, ox;
    if (KS == 1) { HH = TH; HW = TW; oy = y0; ox = x0; }
    else if (UPS) { HH = (TH >> 1) + 2; HW = (TW >> 1) + 2; oy = (y0 >> 1) - 1; ox = (x0 >> 1) - 1; }
    else { HH = TH + 2; HW = TW + 2; oy = y0 - 1; ox = x0 - 1; }
    const int a_bytes = NBv * HH * HW * SA;
    unsigned char* A_hi = smem;
    unsigned char* A_lo = smem + a_bytes;
    unsigned char* B_base = smem + (SPLIT ? 2 : 1) * a_bytes;         // [2 buffers][hi|lo][BN rows][SB]
    constexpr int B_TILE = BN * SB;
    constexpr int B_BUF = B_TILE * (SPLIT ? 2 : 1);

    // per-lane A-fragment rows
    int a_nbbase[MR], a_py[MR], a_px[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        int m = wm * (MR * 16) + mr * 16 + (lane & 15);
        int nb, py, px;
        tile_row_decode(m, p.lTH, p.lTW, nb, py, px);
        a_nbbase[mr] = nb * HH * HW; a_py[mr] = py; a_px[mr] = px;
    }
    const int kb16 = (lane >> 4) * 16;
    int b_off[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) b_off[nr] = (wn * (NR * 16) + nr * 16 + (lane & 15)) * SB + kb16;

    f32x4_t acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    s16x8_t breg_hi[B_PER_THREAD], breg_lo[B_PER_THREAD];
    auto load_b = [&](int c0, int tap) {
#pragma unroll
        for (int k = 0; k < B_PER_THREAD; ++k) {
            int i = tid + k * 256;
            if (B_ITEMS % 256 == 0 || i < B_ITEMS) {
                int cg = i % CG, n = i / CG;
                size_t off = ((size_t)(tap * p.CoutP + co0 + n) * p.CinP + c0 + cg * 8);
                breg_hi[k] = *(const s16x8_t*)(p.w_hi + off);
                if (SPLIT) breg_lo[k] = *(const s16x8_t*)(p.w_lo + off);
            }
        }
    };
    auto store_b = [&](int buf) {
        unsigned char* dst = B_base + buf * B_BUF;
#pragma unroll
        for (int k = 0; k < B_PER_THREAD; ++k) {
            int i = tid + k * 256;
            if (B_ITEMS % 256 == 0 || i < B_ITEMS) {
                int cg = i % CG, n = i / CG;
                *(s16x8_t*)(dst + n * SB + cg * 16) = breg_hi[k];
                if (SPLIT) *(s16x8_t*)(dst + B_TILE + n * SB + cg * 16) = breg_lo[k];
            }
        }
    };

    int step = 0;
    load_b(0, 0);
    for (int c0 = 0; c0 < p.CinP; c0 += CC) {
        __syncthreads();                        // all waves finished reading the previous chunk's halo
        stage_act_halo<CC, SPLIT>(A_hi, A_lo, SA, p.x, p.scale, p.shift, p.pro, p.N, p.Hin, p.Win, p.Cin,
                                  n0, NBv, HH, HW, oy, ox, c0, tid);
#pragma unroll
        for (int tap = 0; tap < T; ++tap, ++step) {
            const int buf = step & 1;
            store_b(buf);
            __syncthreads();
            // prefetch the next weight tile (next tap, or tap 0 of the next chunk) while computing this one
            if (tap + 1 < T) load_b(c0, tap + 1);
            else if (c0 + CC < p.CinP) load_b(c0 + CC, 0);

            const int dy = (KS == 3) ? tap / 3 : 0, dx = (KS == 3) ? tap % 3 : 0;
            int a_off[MR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                int hy, hx;
                if (KS == 1) { hy = a_py[mr]; hx = a_px[mr]; }
                else if (UPS) { hy = ((a_py[mr] + dy - 1) >> 1) + 1; hx = ((a_px[mr] + dx - 1) >> 1) + 1; }
                else { hy = a_py[mr] + dy; hx = a_px[mr] + dx; }
                a_off[mr] = (a_nbbase[mr] + hy * HW + hx) * SA + kb16;
            }
            const unsigned char* Bc = B_base + buf * B_BUF;
#pragma unroll
            for (int kk = 0; kk < CC / 32; ++kk) {
                s16x8_t a[MR], b[NR], al[MR], bl[NR];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    a[mr] = *(const s16x8_t*)(A_hi + a_off[mr] + kk * 64);
                    if (SPLIT) al[mr] = *(const s16x8_t*)(A_lo + a_off[mr] + kk * 64);
                }
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    b[nr] = *(const s16x8_t*)(Bc + b_off[nr] + kk * 64);
                    if (SPLIT) bl[nr] = *(const s16x8_t*)(Bc + B_TILE + b_off[nr] + kk * 64);
                }
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        if (SPLIT) {
                            acc[mr][nr] = mfma16(al[mr], b[nr], acc[mr][nr]);
                            acc[mr][nr] = mfma16(a[mr], bl[nr], acc[mr][nr]);
                        }
                        acc[mr][nr] = mfma16(a[mr], b[nr], acc[mr][nr]);
                    }
            }
        }
    }

    // epilogue: C layout of mfma 16x16: col = lane&15 (channel), row = (lane>>4)*4 + reg (tile row)
    const float alpha = p.alpha ? *p.alpha : 1.f;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        int m0 = wm * (MR * 16) + mr * 16 + (lane >> 4) * 4;
        int nb, py0, px0;
        tile_row_decode(m0, p.lTH, p.lTW, nb, py0, px0);
        const int n = n0 + nb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oyy = y0 + py0 + (r >> 1), oxx = x0 + px0 + (r & 1);
            if (n >= p.N || oyy >= p.H || oxx >= p.W) continue;
            const size_t pix = ((size_t)(n * p.H + oyy) * p.W + oxx) * p.Cout;
            size_t rpix = 0;
            if (p.res) rpix = ((size_t)(n * (p.H >> p.res_shift) + (oyy >> p.res_shift)) * (p.W >> p.res_shift) + (oxx >> p.res_shift)) * p.Cout;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int co = co0 + wn * (NR * 16) + nr * 16 + (lane & 15);
                if (co < p.Cout) {
                    float v = acc[mr][nr][r] * alpha;
                    if (p.bias) v += p.bias[co];
                    if (p.res) v += p.res[rpix + co];
                    p.y[pix + co] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side: tile selection + dispatch
// ------------------------------------------------------------------------------------------------------------------
static int ilog2_floor(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }

// choose NB x TH x TW = BM with TH<=H, TW<=W (powers of two), preferring wide patches (TW up to 16)
static void choose_tile(int BM, int N, int H, int W, int* lTH, int* lTW, int* lNB) {
    int lbm = ilog2_floor(BM);
    int ltw = ilog2_floor(W); if (ltw > 4) ltw = 4;
    int lth = ilog2_floor(H); if (lth > lbm - ltw) lth = lbm - ltw;
    if (lth < 1) lth = 1;
    if (ltw < 1) ltw = 1;
    int lnb = lbm - ltw - lth; if (lnb < 0) lnb = 0;
    while (lnb > 0 && (1 << (lnb - 1)) >= N) --lnb;      // no more images per tile than exist (keeps the LDS halo small)
    *lTH = lth; *lTW = ltw; *lNB = lnb;
}

template <int KS, bool UPS, int WM, int WN, int MR, int NR, int CC, int PREC>
static int launch_conv(ConvParams& p, hipStream_t stream) {
    constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
    constexpr int SA = CC * 2 + 16, SB = CC * 2 + 16;
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    choose_tile(BM, p.N, p.H, p.W, &p.lTH, &p.lTW, &p.lNB);
    const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;
    p.tiles_x = (p.W + TW - 1) / TW; p.tiles_y = (p.H + TH - 1) / TH;
    int HH, HW;
    if (KS == 1) { HH = TH; HW = TW; } else if (UPS) { HH = TH / 2 + 2; HW = TW / 2 + 2; } else { HH = TH + 2; HW = TW + 2; }
    size_t lds = (size_t)NBv * HH * HW * SA * (SPLIT ? 2 : 1) + (size_t)2 * BN * SB * (SPLIT ? 2 : 1);
    if (lds > 160 * 1024) return lp_set_error(LP_ERR_UNSUPPORTED, "conv tile needs too much LDS");
    auto kern = conv_igemm_kernel<KS, UPS, WM, WN, MR, NR, CC, PREC>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_set = true;
    }
    dim3 grid(p.tiles_x * p.tiles_y * ((p.N + NBv - 1) / NBv), (p.Cout + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return lp_check_launch("conv_igemm");
}

template <int PREC>
static int dispatch_conv(ConvParams& p, int ks, int ups, hipStream_t s) {
    const bool small_cin = p.CinP % 64 != 0;   // packs of tiny-Cin layers are padded to 32 only
    const bool big_img = p.H * p.W >= 256;     // a 256-pixel patch fits inside one image
    if (small_cin) {
        if (ks == 3 && !ups) return launch_conv<3, false, 4, 1, 4, 4, 32, PREC>(p, s);
        return lp_set_error(LP_ERR_UNSUPPORTED, "CinP%64!=0 only supported for 3x3 non-upsampled convs");
    }
    if (ks == 3 && !ups) {
        if (p.Cout <= 16 && big_img) return launch_conv<3, false, 4, 1, 4, 1, 64, PREC>(p, s);
        if (p.Cout <= 64 && big_img) return launch_conv<3, false, 4, 1, 4, 4, 64, PREC>(p, s);
        return launch_conv<3, false, 2, 2, 4, 4, 64, PREC>(p, s);
    }
    if (ks == 3 && ups) {
        if (p.Cout <= 64 && big_img) return launch_conv<3, true, 4, 1, 4, 4, 64, PREC>(p, s);
        return launch_conv<3, true, 2, 2, 4, 4, 64, PREC>(p, s);
    }
    if (ks == 1 && !ups) {
        if (p.Cout <= 64 && big_img) return launch_conv<1, false, 4, 1, 4, 4, 64, PREC>(p, s);
        return launch_conv<1, false, 2, 2, 4, 4, 64, PREC>(p, s);
    }
    return lp_set_error(LP_ERR_UNSUPPORTED, "unsupported conv configuration");
}

extern "C" int lp_conv_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                           const float* scale, const float* shift, const float* bias, const float* res, const float* alpha,
                           int N, int H, int W, int Cin, int Cout, int CinP, int CoutP,
                           int ksize, int upsample, int pro, int res_shift, int prec, void* stream) {
    if (!x || !w_hi || !y) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: null pointer");
    if (pro == 1 && (!scale || !shift)) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: pro=1 needs scale/shift");
    if (prec == LP_PREC_BF16X3 && !w_lo) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: bf16x3 needs w_lo");
    if (upsample && ((H | W) & 1)) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: upsampled output dims must be even");
    if (CinP % 32 || CinP < Cin || CoutP % 128 || CoutP < Cout) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: bad padded dims");
    if (H < 2 || W < 2) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_conv_fwd: H,W must be >= 2");
    ConvParams p;
    p.x = x; p.w_hi = w_hi; p.w_lo = w_lo; p.y = y; p.scale = scale; p.shift = shift; p.bias = bias; p.res = res; p.alpha = alpha;
    p.N = N; p.H = H; p.W = W; p.Hin = upsample ? H / 2 : H; p.Win = upsample ? W / 2 : W;
    p.Cin = Cin; p.Cout = Cout; p.CinP = CinP; p.CoutP = CoutP; p.res_shift = res_shift; p.pro = pro;
    hipStream_t s = (hipStream_t)stream;
    if (prec == LP_PREC_BF16) return dispatch_conv<LP_PREC_BF16>(p, ksize, upsample, s);
    if (prec == LP_PREC_BF16X3) return dispatch_conv<LP_PREC_BF16X3>(p, ksize, upsample, s);
    return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: unknown precision mode");
}
