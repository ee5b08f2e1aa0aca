// Fused implicit-GEMM convolution for gfx950 (MFMA 16x16x32 bf16, fp32 accumulate).
//
//   y[n,oy,ox,co] = alpha * sum_{tap,ci} act(x)[n, oy+dy, ox+dx, ci] * w[tap][co][ci]  (+ bias[co]) (+ res[n,oy>>rs,ox>>rs,co])
//
// act() is the *prologue* of the consumer conv (pre-activation ResBlock, reference generators/common/blocks.py:70-88):
// AdaIN affine (x*scale[n,c]+shift[n,c]) -> ReLU -> optional nearest x2 upsample, all applied while the input halo tile
// is staged into LDS, so the normalised/upsampled tensor never exists in HBM.  The same kernel is the data-gradient
// kernel when fed dY and the flipped/transposed weight pack (lp_pack_weights mode 1).
//
// Tiling: a group of WM*WN waves (4, or 8 for the small feature maps) owns BM = WM*MR*16 output pixels (NB images x TH x TW
// patch) and BN = WN*NR*16 output channels; a workgroup is one group, or two groups on adjacent pixel tiles in the ping-pong
// schedule (see conv_igemm_kernel).  K loop = input-channel chunks of CC; per chunk the activated halo is staged ONCE and
// reused by all KS*KS taps; the weight tile of each kernel row travels L2 -> LDS by LDS-DMA, double buffered.  Small feature
// maps additionally split K over gridDim.z (fp32 atomic epilogue).  Convs with <= 4 input channels never get here
// (conv_thin.hip).  Tuning / ablation knobs (read once per process): LP_CONV_PP, LP_CONV_W8, LP_CONV_KSPLIT, LP_CONV_NBUF,
// LP_CONV_CC, LP_CONV_THIN; -DLP_DBG adds the LP_CONV_DBG ablation bits and (with -DLP_PROF) s_memtime phase counters.
#include "lp_common.h"
#include "lp_hip.h"
#include "lp_internal.h"
#include <stdlib.h>

#ifdef LP_DBG
static unsigned long long* g_prof = nullptr;
extern "C" void lp_dbg_set_prof(unsigned long long* ptr) { g_prof = ptr; }
#endif

struct ConvParams {
    const float* x; const uint16_t* w_hi; const uint16_t* w_lo; float* y;
    const float* scale; const float* shift; const float* bias; const float* res; const float* alpha;
    const float* mask;        // epilogue: y = 0 where mask <= 0 (fused ReLU backward of the dgrad launch)
    int N, H, W, Hin, Win, Cin, Cout, CinP, CoutP;
    int res_shift, pro;
    int lTH, lTW, lNB, tiles_x, tiles_y;
#ifdef LP_DBG
    unsigned long long* prof; // [6] cycle counters of block 0 / wave 0: sync, dma issue, halo load issue, mfma, halo write, total
    int dbg;                  // ablation bitmask (LP_CONV_DBG): 1 skip halo restaging, 2 skip weight DMA, 4 skip MFMAs, 8 skip epilogue, 16 one fragment fetch per stage
#endif
    int ksplit;               // split-K: gridDim.z workgroups share one output tile (fp32 atomic epilogue onto a zeroed y)
    int a_dbuf;               // activation halo double-buffered in LDS (1) or single-buffered with an extra barrier (0)
};

// Main-loop structure (one workgroup = 4 waves, one wave per SIMD, up to 160 KiB LDS):
//   stage = one kernel ROW (KS taps) of one CC-channel chunk.  The weight tile of a stage ([KS*BN rows][CC] bf16) travels
//   HBM -> LDS by LDS-DMA (global_load_lds, 16 B/lane, no VGPRs) into one of two stage buffers, issued one stage ahead so the
//   DMA of stage s+1 overlaps the MFMAs of stage s.  The DMA destination is lane-linear, so the XOR bank swizzle is applied to
//   the per-lane SOURCE address (16-byte chunk index ^ row key) and undone on the ds_read_b128 side -- conflict-free B-fragment
//   reads without padding.  The activated input halo of a chunk is staged once (AdaIN/ReLU/upsample prologue in registers) into
//   one of two halo buffers while the previous chunk's last stage is still being multiplied.
//
// Ping-pong variant (PP): the workgroup has TWO groups of 4 waves (one wave of each group on every SIMD).  Each group owns
// its own output tile (adjacent M tiles, same N tile) and its own single halo buffer; the weight stages are shared, so they
// are fetched once for both tiles.  A stage slot has two phases separated by workgroup barriers: in phase A group 0 runs the
// stage's MFMAs while group 1 does its staging work (its share of the next stage's weight DMA, halo loads / prologue / LDS
// write), in phase B the roles swap.  The matrix pipe of every SIMD therefore always has one wave multiplying while the
// other wave's LDS-DMA issue, address math and fp32->bf16 conversion run beside it, instead of all waves staging (pipe idle)
// and then all waves multiplying as in the one-group schedule.
template <int KS, bool UPS, int WM, int WN, int MR, int NR, int CC, int PREC, bool FAST, int NBUF, bool PP = false>
__global__ __launch_bounds__(WM * WN * 64 * (PP ? 2 : 1), (!PP && WM * WN == 4 && CC == 32 && PREC == LP_PREC_BF16 && NBUF == 2) ? 2 : 1)
void conv_igemm_kernel(ConvParams p) {
    constexpr int NWAVE = WM * WN, NT = NWAVE * 64;      // waves / threads of one group: 4 waves (one per SIMD) or 8
    constexpr int NWD = PP ? 2 * NWAVE : NWAVE;          // waves that share the weight DMA of a stage
    static_assert(!PP || (FAST && KS == 3 && NBUF == 2 && NWAVE == 4), "ping-pong: two 4-wave groups, 3x3, fast staging");
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
    constexpr int SA = CC * 2 + 16;                 // padded halo row stride (bytes)
    constexpr int ROWB = CC * 2;                    // weight row bytes (unpadded, swizzled)
    constexpr int SLOTS = CC / 8;                   // 16-byte chunks per weight row
    constexpr int RPI = 64 / SLOTS;                 // weight rows covered by one wave-wide LDS-DMA (1 KiB)
    constexpr int B_STAGE = KS * BN * ROWB;         // bytes of one stage (hi part)
    constexpr int B_BUF = B_STAGE * (SPLIT ? 2 : 1);
    constexpr int NQ = B_STAGE / 1024;              // DMA instructions per stage (hi part)
    constexpr int DMA_PER_WAVE = ((NQ + NWD - 1) / NWD) * (SPLIT ? 2 : 1);   // LDS-DMA instructions one wave issues per stage
    static_assert(NBUF == 2 || (NBUF == 3 && NQ % NWD == 0), "3-deep ring needs a uniform DMA count per wave");
    static_assert(!(UPS && KS == 1), "1x1 convs commute with nearest upsampling: run them at low resolution");
    static_assert(NWAVE == 4 || NWAVE == 8, "4 or 8 waves per workgroup");
    static_assert(B_STAGE % 1024 == 0, "stage must be a whole number of 1 KiB DMA pieces");
    // FAST halo staging: every thread owns up to AIT (pixel, 8-channel group) items whose loads are issued together
    constexpr int CG = CC / 8, PPP = NT / CG;                        // pixels covered by one pass of the workgroup's threads
    constexpr int MAXHALO = (BM == 128) ? 10 * 18 : 18 * 18;        // 8x16 / 16x16 patch + 1-pixel border
    constexpr int AIT = FAST ? (MAXHALO + PPP - 1) / PPP : 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int grp = PP ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;     // ping-pong group of this wave
    const int tid = PP ? ((int)threadIdx.x & 255) : (int)threadIdx.x, lane = tid & 63;       // thread / wave index inside the group
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_d = PP ? grp * NWAVE + wave : wave;                                       // index among the waves sharing the DMA
    const int wm = wave / WN, wn = wave % WN;
    const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;

    int t = PP ? (int)blockIdx.x * 2 + grp : (int)blockIdx.x;     // (an odd tile count leaves group 1 of the last workgroup a
    const int tx = t % p.tiles_x; t /= p.tiles_x;                 //  tile past the last image: fully masked)
    const int ty = t % p.tiles_y; const int ng = t / p.tiles_y;
    const int n0 = ng << p.lNB, y0 = ty << p.lTH, x0 = tx << p.lTW;
    const int co0 = blockIdx.y * BN;

    int HH, HW, oy, ox;
    if (KS == 1) { HH = TH; HW = TW; oy = y0; ox = x0; }
    else if (UPS) { HH = (TH >> 1) + 2; HW = (TW >> 1) + 2; oy = (y0 >> 1) - 1; ox = (x0 >> 1) - 1; }
    else { HH = TH + 2; HW = TW + 2; oy = y0 - 1; ox = x0 - 1; }
    const int a_bytes = NBv * HH * HW * SA;
    const int a_buf = a_bytes * (SPLIT ? 2 : 1);                       // one halo buffer: [hi][lo]
    unsigned char* B_base = smem + a_buf * (FAST ? 2 : 1);             // FAST: two halo buffers; then two stage buffers [hi][lo]

    int a_nbbase[MR], a_py[MR], a_px[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        int m = wm * (MR * 16) + mr * 16 + (lane & 15);
        int nb, py, px;
        tile_row_decode(m, p.lTH, p.lTW, nb, py, px);
        a_nbbase[mr] = nb * HH * HW; a_py[mr] = py; a_px[mr] = px;
    }
    const int kb = lane >> 4, kb16 = kb * 16;
    const int bkey = (SLOTS == 8) ? (lane & 7) : ((lane >> 2) & 3);    // swizzle key of this lane's weight rows (row & 15 == lane & 15)
    int b_off[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) b_off[nr] = (wn * (NR * 16) + nr * 16 + (lane & 15)) * ROWB;

    f32x4_t acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- FAST halo staging descriptors (chunk independent): pixel index in x, -1 = zero padding, -2 = not an item
    const int halo_px = NBv * HH * HW;
    const int a_cg = tid % CG, a_hp0 = tid / CG;
    int a_pix[AIT];
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int hp = a_hp0 + k * PPP;
        const int hx = hp % HW, hy = hp / HW;
        const int iy = oy + hy, ix = ox + hx;
        const bool inb = (hp < halo_px) && (n0 < p.N) && (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
        a_pix[k] = inb ? ((n0 * p.Hin + iy) * p.Win + ix) : (hp < halo_px ? -1 : -2);
    }
    float4 a_ld[AIT][2];
    // issue the global loads of chunk `chunk` (no waits): they stay in flight across the MFMAs that follow.  The loads are
    // UNCONDITIONAL (out-of-image items read pixel 0 and are zeroed at write time): a branch per item would make hipcc wait
    // for every load separately.
    auto load_a = [&](int chunk) {
        int c = chunk * CC + a_cg * 8;
        c = c < p.Cin ? c : 0;
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int pix = a_pix[k] >= 0 ? a_pix[k] : 0;
            const float* src = p.x + (size_t)pix * p.Cin + c;
            a_ld[k][0] = *(const float4*)src; a_ld[k][1] = *(const float4*)(src + 4);
        }
    };
    // prologue (AdaIN affine / ReLU), bf16 (hi, lo) conversion and LDS write of the loaded items
    auto write_a = [&](int chunk, int buf) {
        unsigned char* A = smem + buf * a_buf;
        const int c = chunk * CC + a_cg * 8;
        const bool cok = c < p.Cin;
        const int cs = cok ? c : 0;
        float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0, t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
        if (p.pro == 1) {
            const float* sp = p.scale + (size_t)n0 * p.Cin + cs; const float* tp = p.shift + (size_t)n0 * p.Cin + cs;
            s0 = *(const float4*)sp; s1 = *(const float4*)(sp + 4); t0 = *(const float4*)tp; t1 = *(const float4*)(tp + 4);
        }
        const float lo_clamp = (p.pro != 0) ? 0.f : -3.0e38f;          // ReLU for pro 1|2, identity for pro 0
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            float v[8] = {a_ld[k][0].x, a_ld[k][0].y, a_ld[k][0].z, a_ld[k][0].w, a_ld[k][1].x, a_ld[k][1].y, a_ld[k][1].z, a_ld[k][1].w};
            v[0] = fmaxf(fmaf(v[0], s0.x, t0.x), lo_clamp); v[1] = fmaxf(fmaf(v[1], s0.y, t0.y), lo_clamp);
            v[2] = fmaxf(fmaf(v[2], s0.z, t0.z), lo_clamp); v[3] = fmaxf(fmaf(v[3], s0.w, t0.w), lo_clamp);
            v[4] = fmaxf(fmaf(v[4], s1.x, t1.x), lo_clamp); v[5] = fmaxf(fmaf(v[5], s1.y, t1.y), lo_clamp);
            v[6] = fmaxf(fmaf(v[6], s1.z, t1.z), lo_clamp); v[7] = fmaxf(fmaf(v[7], s1.w, t1.w), lo_clamp);
            const bool keep = (a_pix[k] >= 0) && cok;            // zero padding is applied AFTER the activation
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = keep ? v[j] : 0.f;
            s16x8_t hi, lo;
            cvt8<SPLIT>(v, hi, lo);
            const int off = (a_hp0 + k * PPP) * SA + a_cg * 16;
            if (a_pix[k] != -2) {
                *(s16x8_t*)(A + off) = hi;
                if (SPLIT) *(s16x8_t*)(A + a_bytes + off) = lo;
            }
        }
    };

    // LDS-DMA of the weight tile of stage (chunk, ky) into stage buffer `buf`
    auto issue_b = [&](int chunk, int ky, int buf) {
        const int c0 = chunk * CC;
        const unsigned dst_lds = (unsigned)(uintptr_t)(B_base + buf * B_BUF);      // LDS byte address (wave-uniform)
        const int rr = lane / SLOTS, slot = lane % SLOTS;
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += NWD) {
            const int q = q0 + wave_d;
            bool doit = (NQ % NWD == 0 || q < NQ);
#ifdef LP_DBG
            if ((p.dbg & 32) && ((q0 / NWD) & 1)) doit = false;      // ablation: half the DMA pieces
#endif
            if (doit) {
                const int r = q * RPI + rr;                                           // row inside the stage: kx * BN + n
                const int key = (SLOTS == 8) ? (r & 7) : ((r >> 2) & 3);
                const int kx = r / BN, n = r % BN;
                const size_t off = ((size_t)((ky * KS + kx) * p.CoutP + co0 + n) * p.CinP + c0 + ((slot ^ key) * 8));
                lp_glds16(p.w_hi + off, dst_lds + q * 1024);
                if (SPLIT) lp_glds16(p.w_lo + off, dst_lds + B_STAGE + q * 1024);
            }
        }
    };
    auto stage_a_slow = [&](int chunk, int buf) {
        unsigned char* A = smem + buf * a_buf;
        stage_act_halo<CC, SPLIT, NT>(A, A + a_bytes, SA, p.x, p.scale, p.shift, p.pro, p.N, p.Hin, p.Win, p.Cin,
                                  n0, NBv, HH, HW, oy, ox, chunk * CC, tid);
    };
    // MFMAs of one stage: kernel row ky of the chunk whose halo is in halo buffer `abuf`, weights in stage buffer `bbuf`
    auto compute = [&](int ky, int abuf, int bbuf) {
        const unsigned char* A_hi = smem + abuf * a_buf;
        const unsigned char* A_lo = A_hi + a_bytes;
        const unsigned char* Bc = B_base + bbuf * B_BUF;
        // k-steps of the stage = (kx, kk); fragments are double-buffered in registers and those of step s+1 are requested before
        // the MFMAs of step s, so with one wave per SIMD the LDS latency hides behind a full k-step of matrix work.
        constexpr int KK = CC / 32, STEPS = KS * KK;
        s16x8_t fa[2][MR], fb[2][NR], fal[2][MR], fbl[2][NR];
        auto fetch = [&](int st, int set) {
            const int kx = st / KK, kk = st % KK;
            const int dy = (KS == 3) ? ky : 0, dx = (KS == 3) ? kx : 0;
            const unsigned char* Bk = Bc + kx * (BN * ROWB);
            const int bslot = (((kk * 4 + kb) ^ bkey) * 16);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                int hy, hx;
                if (KS == 1) { hy = a_py[mr]; hx = a_px[mr]; }
                else if (UPS) { hy = ((a_py[mr] + dy - 1) >> 1) + 1; hx = ((a_px[mr] + dx - 1) >> 1) + 1; }
                else { hy = a_py[mr] + dy; hx = a_px[mr] + dx; }
                const int off = (a_nbbase[mr] + hy * HW + hx) * SA + kb16 + kk * 64;
                fa[set][mr] = *(const s16x8_t*)(A_hi + off);
                if (SPLIT) fal[set][mr] = *(const s16x8_t*)(A_lo + off);
            }
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                fb[set][nr] = *(const s16x8_t*)(Bk + b_off[nr] + bslot);
                if (SPLIT) fbl[set][nr] = *(const s16x8_t*)(Bk + B_STAGE + b_off[nr] + bslot);
            }
        };
        fetch(0, 0);
#ifdef LP_DBG
        if (p.dbg & 16) fetch(0, 1);                     // ablation: one fragment fetch per stage, MFMAs re-use it
#endif
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int cur = st & 1;
#ifdef LP_DBG
            if (st + 1 < STEPS && !(p.dbg & 16)) fetch(st + 1, cur ^ 1);
#else
            if (st + 1 < STEPS) fetch(st + 1, cur ^ 1);
#endif
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    if (SPLIT) {
                        acc[mr][nr] = mfma16(fal[cur][mr], fb[cur][nr], acc[mr][nr]);
                        acc[mr][nr] = mfma16(fa[cur][mr], fbl[cur][nr], acc[mr][nr]);
                    }
                    acc[mr][nr] = mfma16(fa[cur][mr], fb[cur][nr], acc[mr][nr]);
                }
        }
    };

    const int nch_total = p.CinP / CC;
    const int per = (nch_total + p.ksplit - 1) / p.ksplit;
    const int cbeg = blockIdx.z * per;
    const int nch = min(nch_total, cbeg + per) - cbeg;        // chunks of this workgroup: [cbeg, cbeg + nch)
    if (nch <= 0) return;                                       // (uniform) nothing to contribute
    issue_b(cbeg, 0, 0);
    if (FAST) { load_a(cbeg); write_a(cbeg, PP ? grp : 0); } else stage_a_slow(cbeg, 0);

    if constexpr (PP) {
        // slot k = stage k of both groups.  Phase A(k): group 0 multiplies stage k, group 1 stages; phase B(k): swapped.
        // Staging duties of a group in slot k:
        //   * its share of the weight DMA of stage k + 1 (buffer (k+1)&1: last read in phase B(k-1), which is over);
        //   * one phase before its compute of the first stage of the next chunk: prologue + LDS write of that chunk's halo into
        //     the group's (single) halo buffer -- the group itself is the only reader and it is not multiplying now;
        //   * one staging phase earlier: issue the global loads of that halo (they fly during the compute phase in between).
        // Weight DMA is untracked inline asm and is always issued BEFORE the halo loads of the same phase, so "at most
        // 2*AIT loads outstanding" == "all DMA landed".
        const int S = nch * KS;
        for (int chunk = 0; chunk < nch; ++chunk) {
            const bool has_next = chunk + 1 < nch;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int k = chunk * KS + ky;
                const int bbuf = k & 1;
                // halo loads issued in the previous slot may stay in flight: group 0 issues them in B(k) with ky == KS-2,
                // group 1 in A(k) with ky == KS-1 (both: only if a next chunk exists)
                const bool loads_in_flight = (grp == 0) ? (ky == KS - 1 && has_next) : (ky == 0 && chunk > 0);
                if (loads_in_flight) lp_wait_vm<2 * AIT>(); else lp_wait_vm0();
                __syncthreads();                                   // ---- phase A(k)
                if (grp == 0) {
                    compute(ky, 0, bbuf);
                } else {
                    if (ky == 0 && chunk > 0) write_a(cbeg + chunk, 1);               // halo of this chunk (loaded in A(k-1))
                    if (k + 1 < S) issue_b(cbeg + (k + 1) / KS, (k + 1) % KS, bbuf ^ 1);
                    if (ky == KS - 1 && has_next) load_a(cbeg + chunk + 1);
                }
                __syncthreads();                                   // ---- phase B(k)
                if (grp == 1) {
                    compute(ky, 1, bbuf);
                } else {
                    if (ky == KS - 1 && has_next) write_a(cbeg + chunk + 1, 0);       // halo of the next chunk (loaded in B(k-1))
                    if (k + 1 < S) issue_b(cbeg + (k + 1) / KS, (k + 1) % KS, bbuf ^ 1);
                    if (ky == KS - 2 && has_next) load_a(cbeg + chunk + 1);
                }
            }
        }
    } else if (FAST) {
        // Straight-line pipeline (no data-dependent control flow around memory ops, so hipcc places no early vmcnt waits):
        // stage = kernel row.  Top of stage: wait own DMA, barrier; issue next stage's DMA; [last row: issue next chunk's
        // halo loads]; MFMAs; [last row: prologue + LDS write of the next halo into the other halo buffer].
        // NBUF-deep weight ring: at the top of stage s the DMA of stage s+NBUF-1 is issued; only the DMA of stage s itself
        // must have landed, so with NBUF == 3 one stage's worth of DMA instructions stays in flight across the barrier.
        const int S = nch * KS;
        int abuf = 0;
#ifdef LP_DBG
        unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
        const unsigned long long tstart = __builtin_amdgcn_s_memtime();
#ifdef LP_PROF
#define LP_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#else
#define LP_T(var) const unsigned long long var = 0;
#endif
#endif
        if (NBUF == 3 && S > 1) issue_b(cbeg + (KS > 1 ? 0 : 1), KS > 1 ? 1 : 0, 1);
        for (int chunk = 0; chunk < nch; ++chunk) {
            const bool has_next = chunk + 1 < nch;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int s = chunk * KS + ky;
#ifdef LP_DBG
                LP_T(q0)
#endif
                if (NBUF == 3) { if (s + 1 < S) lp_wait_vm<DMA_PER_WAVE>(); else lp_wait_vm0(); } else lp_wait_vm0();
                __syncthreads();
                const int bbuf = s % NBUF;
                const int sp = s + NBUF - 1;                       // stage to prefetch now
#ifdef LP_DBG
                LP_T(q1)
                if (sp < S && !(p.dbg & 2)) issue_b(cbeg + sp / KS, sp % KS, sp % NBUF);
                LP_T(q2)
                if (ky == KS - 1 && has_next && !(p.dbg & 1)) load_a(cbeg + chunk + 1);
                LP_T(q3)
                if (!(p.dbg & 4)) compute(ky, abuf, bbuf);
                LP_T(q4)
                if (ky == KS - 1 && has_next && !(p.dbg & 1)) write_a(cbeg + chunk + 1, abuf ^ 1);
                LP_T(q5)
                pc[0] += q1 - q0; pc[1] += q2 - q1; pc[2] += q3 - q2; pc[3] += q4 - q3; pc[4] += q5 - q4;
#else
                if (sp < S) issue_b(cbeg + sp / KS, sp % KS, sp % NBUF);
                if (ky == KS - 1 && has_next) load_a(cbeg + chunk + 1);
                compute(ky, abuf, bbuf);
                if (ky == KS - 1 && has_next) write_a(cbeg + chunk + 1, abuf ^ 1);
#endif
            }
            abuf ^= 1;
        }
#ifdef LP_DBG
        pc[5] = __builtin_amdgcn_s_memtime() - tstart;
        if (p.prof && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0)
            for (int i = 0; i < 6; ++i) p.prof[i] = pc[i];
#endif
    } else {
        // generic path (several images per tile / odd channel counts): single halo buffer, synchronous staging
        for (int chunk = 0; chunk < nch; ++chunk) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                lp_wait_vm0();
                __syncthreads();
                const int bbuf = (ky + chunk * KS) & 1;
                if (ky + 1 < KS) issue_b(cbeg + chunk, ky + 1, bbuf ^ 1);
                else if (chunk + 1 < nch) issue_b(cbeg + chunk + 1, 0, bbuf ^ 1);
                compute(ky, 0, bbuf);
            }
            if (chunk + 1 < nch) { __syncthreads(); stage_a_slow(cbeg + chunk + 1, 0); }
        }
    }

    // epilogue: C layout of mfma 16x16: col = lane&15 (channel), row = (lane>>4)*4 + reg (tile row)
#ifdef LP_DBG
    if (p.dbg & 8) return;
#endif
    const float alpha = p.alpha ? *p.alpha : 1.f;
    if (p.ksplit == 1 && (p.Cout & 3) == 0) {
        // Coalesced path: every wave transposes its (MR*16) x (NR*16) accumulator block through LDS (the staging buffers are
        // dead now) and writes whole pixel rows -- NR*64 contiguous bytes per pixel, 16 B per lane -- instead of 64-B
        // fragments; bias and the residual are read the same way.
        constexpr int WR = MR * 16, WC = NR * 16, LDW = WC + 4;          // +4 floats: rows land on different banks
        __syncthreads();                                                  // all waves are done with the halo / weight buffers
        float* tile = (float*)smem + wave_d * (WR * LDW);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    tile[(mr * 16 + (lane >> 4) * 4 + r) * LDW + nr * 16 + (lane & 15)] = acc[mr][nr][r];
        // (a wave only reads back what it wrote itself: no workgroup barrier needed, just the LDS write->read order)
        constexpr int C4 = WC / 4;                 // float4 columns per row
        constexpr int RPP = 64 / C4;               // rows per pass of the wave
        const int c4 = lane % C4, rsub = lane / C4;
        const int co = co0 + wn * WC + c4 * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && co < p.Cout) bv = *(const float4*)(p.bias + co);
#pragma unroll 4
        for (int r0 = 0; r0 < WR; r0 += RPP) {
            const int row = r0 + rsub;             // row inside the wave block
            const int m = wm * WR + row;
            int nb, py, px;
            tile_row_decode(m, p.lTH, p.lTW, nb, py, px);
            const int n = n0 + nb, oyy = y0 + py, oxx = x0 + px;
            if (n < p.N && oyy < p.H && oxx < p.W && co < p.Cout) {
                float4 v = *(const float4*)(tile + row * LDW + c4 * 4);
                v.x = fmaf(v.x, alpha, bv.x); v.y = fmaf(v.y, alpha, bv.y); v.z = fmaf(v.z, alpha, bv.z); v.w = fmaf(v.w, alpha, bv.w);
                if (p.res) {
                    const float4 rv = *(const float4*)(p.res + ((size_t)(n * (p.H >> p.res_shift) + (oyy >> p.res_shift)) * (p.W >> p.res_shift)
                                                                + (oxx >> p.res_shift)) * p.Cout + co);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                if (p.mask) {
                    const float4 mv = *(const float4*)(p.mask + ((size_t)(n * p.H + oyy) * p.W + oxx) * p.Cout + co);
                    v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f; v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
                }
                *(float4*)(p.y + ((size_t)(n * p.H + oyy) * p.W + oxx) * p.Cout + co) = v;
            }
        }
        return;
    }
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
                    if (p.ksplit == 1 || blockIdx.z == 0) {
                        if (p.bias) v += p.bias[co];
                        if (p.res) v += p.res[rpix + co];
                    }
                    if (p.mask && !(p.mask[pix + co] > 0.f)) v = 0.f;          // (0/1 mask: commutes with the split-K sum)
                    if (p.ksplit == 1) p.y[pix + co] = v;
                    else unsafeAtomicAdd(p.y + pix + co, v);       // y was zeroed by lp_conv_fwd (hipMemsetAsync on the stream)
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

template <int KS, bool UPS, int WM, int WN, int MR, int NR, int CC, int PREC, bool FAST, int NBUF, bool PP = false>
static int launch_conv_v(ConvParams& p, size_t lds, dim3 grid, hipStream_t stream) {
    auto kern = conv_igemm_kernel<KS, UPS, WM, WN, MR, NR, CC, PREC, FAST, NBUF, PP>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64 * (PP ? 2 : 1)), lds, stream, p);
    return lp_check_launch("conv_igemm");
}

template <int KS, bool UPS, int WM, int WN, int MR, int NR, int CC, int PREC>
static int launch_conv(ConvParams& p, hipStream_t stream) {
    constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
    constexpr int SA = CC * 2 + 16;
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    constexpr size_t B_BUF = (size_t)KS * BN * CC * 2 * (SPLIT ? 2 : 1);
    constexpr size_t LDS_MAX = 160 * 1024;
    constexpr int PPP = WM * WN * 64 / (CC / 8);
    constexpr int AIT = (((BM == 128) ? 10 * 18 : 18 * 18) + PPP - 1) / PPP;
    choose_tile(BM, p.N, p.H, p.W, &p.lTH, &p.lTW, &p.lNB);
    const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;
    p.tiles_x = (p.W + TW - 1) / TW; p.tiles_y = (p.H + TH - 1) / TH;
    int HH, HW;
    if (KS == 1) { HH = TH; HW = TW; } else if (UPS) { HH = TH / 2 + 2; HW = TW / 2 + 2; } else { HH = TH + 2; HW = TW + 2; }
    const size_t a_buf = (size_t)NBv * HH * HW * SA * (SPLIT ? 2 : 1);
    const bool fast = (NBv == 1) && ((p.Cin & 7) == 0) && (HH * HW <= AIT * PPP) && (2 * a_buf + 2 * B_BUF <= LDS_MAX);
    p.a_dbuf = fast ? 1 : 0;
    static const int want_nbuf = getenv("LP_CONV_NBUF") ? atoi(getenv("LP_CONV_NBUF")) : 2;       // tuning knob: 2 | 3
    constexpr bool ring3_ok = (CC == 32) && (KS == 3) && (BN >= 64) && !SPLIT;
    const bool ring3 = ring3_ok && fast && want_nbuf == 3 && (2 * a_buf + 3 * B_BUF <= LDS_MAX);
    size_t lds = a_buf * (fast ? 2 : 1) + (ring3 ? 3 : 2) * B_BUF;
    size_t epi = (size_t)(WM * WN) * (MR * 16) * (NR * 16 + 4) * sizeof(float);       // LDS transpose of the coalesced epilogue
    // ping-pong schedule (two 4-wave groups, two adjacent M tiles per workgroup): default for bf16x3, whose single-group kernel
    // runs one wave per SIMD; the bf16 kernel already interleaves two workgroups per CU.  LP_CONV_PP = 0 | 1 overrides.
    constexpr bool pp_ok = (KS == 3) && (WM * WN == 4) && (CC == 32);
    static const int pp_env = getenv("LP_CONV_PP") ? atoi(getenv("LP_CONV_PP")) : -1;
    const int tiles = p.tiles_x * p.tiles_y * ((p.N + NBv - 1) / NBv);
    // (pairing tiles halves the workgroup count: only when the paired grid still covers the CUs -- measured on 8x32x32x512, 256
    //  tiles: 121 us single-group vs 125 us ping-pong + split-K 2)
    const long long pp_wgs = (long long)((tiles + 1) / 2) * ((p.Cout + BN - 1) / BN);
    const bool pp = pp_ok && fast && !ring3 && (pp_env >= 0 ? pp_env != 0 : (SPLIT && pp_wgs >= 200)) && tiles >= 2 &&
                    (2 * a_buf + 2 * B_BUF <= LDS_MAX);
    if (pp) { lds = 2 * a_buf + 2 * B_BUF; epi *= 2; }
    if (lds < epi) lds = epi;
    if (lds > LDS_MAX) return lp_set_error(LP_ERR_UNSUPPORTED, "conv tile needs too much LDS");
    dim3 grid(pp ? (tiles + 1) / 2 : tiles, (p.Cout + BN - 1) / BN);
    {   // split-K when the output tiling alone cannot fill the 256 CUs (4x4 ... 32x32 layers with K = 9*512)
        static const int max_split = getenv("LP_CONV_KSPLIT") ? atoi(getenv("LP_CONV_KSPLIT")) : 8;
        const int wgs = grid.x * grid.y, nch = p.CinP / CC;
        int ks = 1;
        while (ks < max_split && wgs * ks * 2 <= 256 && nch / (ks * 2) >= 2) ks *= 2;
        p.ksplit = ks;
        grid.z = ks;
        if (ks > 1 && hipMemsetAsync(p.y, 0, (size_t)p.N * p.H * p.W * p.Cout * sizeof(float), stream) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipMemsetAsync failed");
    }
    if constexpr (ring3_ok) { if (ring3) return launch_conv_v<KS, UPS, WM, WN, MR, NR, CC, PREC, true, 3>(p, lds, grid, stream); }
    if constexpr (pp_ok) { if (pp) return launch_conv_v<KS, UPS, WM, WN, MR, NR, CC, PREC, true, 2, true>(p, lds, grid, stream); }
    if (fast) return launch_conv_v<KS, UPS, WM, WN, MR, NR, CC, PREC, true, 2>(p, lds, grid, stream);
    return launch_conv_v<KS, UPS, WM, WN, MR, NR, CC, PREC, false, 2>(p, lds, grid, stream);
}

// Channel-chunk size: 64 for bf16; 32 for bf16x3 (hi+lo images double every LDS tile) and for tiny-Cin packs (CinP % 64 != 0).
template <int PREC, int CC>
static int dispatch_conv_cc(ConvParams& p, int ks, int ups, hipStream_t s) {
    const bool big_img = p.H * p.W >= 256;     // a 256-pixel patch fits inside one image
    // 8-wave workgroups (two waves per SIMD, each 64 x 32 of the 128 x 128 tile) for the small feature maps (<= 16x16): those
    // run split-K with few stages per workgroup, so the prologue (first halo + weight stage) and the epilogue dominate, and
    // twice the threads finish them sooner (measured 4x4: 43 -> 30 us, 16x16: 64 -> 51 us in bf16x3; 32x32 and up are better
    // with 4 waves: the phases of all waves are aligned by the stage barrier, so a second wave per SIMD hides nothing there).
    // LP_CONV_W8 = 0 | 1 forces it off / on for every shape.
    static const int w8_env = getenv("LP_CONV_W8") ? atoi(getenv("LP_CONV_W8")) : -1;
    const bool w8 = w8_env >= 0 ? (w8_env != 0) : (p.H * p.W <= 256);
    if (w8 && p.Cout > 64) {
        if (ks == 3 && !ups) return launch_conv<3, false, 2, 4, 4, 2, CC, PREC>(p, s);
        if (ks == 3 && ups) return launch_conv<3, true, 2, 4, 4, 2, CC, PREC>(p, s);
        if (ks == 1 && !ups) return launch_conv<1, false, 2, 4, 4, 2, CC, PREC>(p, s);
    }
    if (ks == 3 && !ups) {
        if (p.Cout <= 16 && big_img) return launch_conv<3, false, 4, 1, 4, 1, CC, PREC>(p, s);
        if (p.Cout <= 64 && big_img) return launch_conv<3, false, 4, 1, 4, 4, CC, PREC>(p, s);
        return launch_conv<3, false, 2, 2, 4, 4, CC, PREC>(p, s);
    }
    if (ks == 3 && ups) {
        if (p.Cout <= 64 && big_img) return launch_conv<3, true, 4, 1, 4, 4, CC, PREC>(p, s);
        return launch_conv<3, true, 2, 2, 4, 4, CC, PREC>(p, s);
    }
    if (ks == 1 && !ups) {
        if (p.Cout <= 64 && big_img) return launch_conv<1, false, 4, 1, 4, 4, CC, PREC>(p, s);
        return launch_conv<1, false, 2, 2, 4, 4, CC, PREC>(p, s);
    }
    return lp_set_error(LP_ERR_UNSUPPORTED, "unsupported conv configuration");
}

template <int PREC>
static int dispatch_conv(ConvParams& p, int ks, int ups, hipStream_t s) {
    static const int force_cc = getenv("LP_CONV_CC") ? atoi(getenv("LP_CONV_CC")) : 0;     // tuning knob: 32 | 64
    // default CC = 32: two workgroups fit per CU (78 KB LDS, <= 256 registers) and hide each other's staging latency
    // (bf16x3 doubles every LDS image: 64-channel chunks do not fit there, the knob only applies to the bf16 kernel)
    if constexpr (PREC == LP_PREC_BF16X3) {
        return dispatch_conv_cc<PREC, 32>(p, ks, ups, s);
    } else {
        if (force_cc != 64 || p.CinP % 64 != 0) return dispatch_conv_cc<PREC, 32>(p, ks, ups, s);
        return dispatch_conv_cc<PREC, 64>(p, ks, ups, s);
    }
}

extern "C" int lp_conv_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                           const float* scale, const float* shift, const float* bias, const float* res, const float* alpha,
                           int N, int H, int W, int Cin, int Cout, int CinP, int CoutP,
                           int ksize, int upsample, int pro, int res_shift, int prec, const float* relu_mask, void* stream) {
    if (!x || !w_hi || !y) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: null pointer");
    if (pro == 1 && (!scale || !shift)) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: pro=1 needs scale/shift");
    if (prec == LP_PREC_BF16X3 && !w_lo) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: bf16x3 needs w_lo");
    if (upsample && ((H | W) & 1)) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: upsampled output dims must be even");
    if (CinP % 32 || CinP < Cin || CoutP % 128 || CoutP < Cout) return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: bad padded dims");
    if (H < 2 || W < 2) return lp_set_error(LP_ERR_UNSUPPORTED, "lp_conv_fwd: H,W must be >= 2");
    {   // RGB -> 64 first convs: direct fp32 kernel (conv_thin.hip); LP_CONV_THIN=0 sends them through the MFMA kernel instead
        static const int thin_env = getenv("LP_CONV_THIN") ? atoi(getenv("LP_CONV_THIN")) : 1;
        if (thin_env && !relu_mask && lp_conv_thin_fwd_supported(Cin, Cout, ksize, upsample, pro, res != nullptr, W))
            return lp_conv_thin_fwd(x, w_hi, prec == LP_PREC_BF16X3 ? w_lo : nullptr, y, bias, alpha, N, H, W, Cin, Cout, CinP, CoutP, ksize,
                                    (hipStream_t)stream);
    }
    ConvParams p;
    p.x = x; p.w_hi = w_hi; p.w_lo = w_lo; p.y = y; p.scale = scale; p.shift = shift; p.bias = bias; p.res = res; p.alpha = alpha; p.mask = relu_mask;
    p.N = N; p.H = H; p.W = W; p.Hin = upsample ? H / 2 : H; p.Win = upsample ? W / 2 : W;
    p.Cin = Cin; p.Cout = Cout; p.CinP = CinP; p.CoutP = CoutP; p.res_shift = res_shift; p.pro = pro;
#ifdef LP_DBG
    static const int dbgv = getenv("LP_CONV_DBG") ? atoi(getenv("LP_CONV_DBG")) : 0;
    p.dbg = dbgv;
    p.prof = g_prof;
#endif

    hipStream_t s = (hipStream_t)stream;
    if (prec == LP_PREC_BF16) return dispatch_conv<LP_PREC_BF16>(p, ksize, upsample, s);
    if (prec == LP_PREC_BF16X3) return dispatch_conv<LP_PREC_BF16X3>(p, ksize, upsample, s);
    return lp_set_error(LP_ERR_ARG, "lp_conv_fwd: unknown precision mode");
}
