// 3x3 implicit-GEMM convolution for gfx950, second main loop: a TAP-granular LDS-DMA pipeline with counted `s_waitcnt vmcnt(N)` (round 4).
//
// Same contraction, operand formats, zero-page padding, fused x2 upsampling and epilogue as conv_dma.hip (conv_common.h); what differs is how
// the K loop is fed.  conv_dma_kernel stages one kernel ROW of weights per barrier, waits `vmcnt(0)` before every barrier (the DMA round trip
// of a stage is exposed in the phase that issued it) and gives every wave 64 x 64 outputs.  Here:
//   * the unit of the loop is one TAP of one 32-channel chunk: BN x 64 B of weights (8 KiB at BN = 128) in a ring of three slots; the weights
//     of tap g+3 are issued right after the barrier of tap g and are only waited for at the barrier of tap g+2 -- two full taps of MFMA work
//     (>= 1000 cycles) later -- with `s_waitcnt vmcnt(N)`, N = the DMA instructions that were issued AFTER the ones needed (a compile-time
//     count: every wave issues the same number of pieces per tap and per halo), so DMA is never drained inside the loop;
//   * the halo of the next chunk is issued at tap 0 of the current one into the other halo buffer (8 taps to land);
//   * every wave multiplies AND issues DMA (no producer / consumer roles), one raw `s_barrier` per tap;
//   * a wave owns 128 x 64 (MR = 8) or 64 x 64 (MR = 4) outputs; its A fragments are read in two halves per tap, each half one step ahead of the
//     MFMAs that consume it, the B fragments of the next tap behind the barrier that publishes them;
//   * 4-wave workgroups at <= 256 VGPRs and 72 KB of LDS: two of them share a CU, so one's epilogue (the fp32 / plane stores that nothing
//     overlapped in the one-workgroup-per-CU ping-pong kernel) runs beside the other's K loop.
// XOR swizzle of the lane-linear LDS images (applied to the DMA SOURCE address, undone by the fragment reads): weights: key (row >> 1) & 3 as
// in conv_dma.hip; halo: key (hx >> 1) & 3 with hx the halo COLUMN -- 16 consecutive pixels of a patch row see the same key sequence as with
// the pixel-index key (halo rows start at even pixel indices), and the address of a fragment read splits into a per-lane term per kernel
// column (3 VGPRs), a wave-uniform row term and an immediate: one VALU add per ds_read_b128.
// Covered: 3x3 (stride 1, optional fused x2 nearest upsampling), Cout >= 128 (128-channel tiles) or 32 < Cout <= 64 (64-channel tiles), maps
// whose tiling fills the chip, all three precision modes; everything else stays on conv_dma_kernel (lp_conv_pipe_launch returns 0).
#include "lp_common.h"
#include "conv_common.h"
#include "lp_hip.h"
#include "lp_internal.h"
#include <stdlib.h>

static __device__ __attribute__((aligned(64))) unsigned int lp_zero_page_pipe[16];      // what out-of-image DMA lanes read

__device__ __forceinline__ void lp_barrier_raw() { asm volatile("s_barrier" ::: "memory"); }

// MFMA with the accumulator TIED (D = C, the same four registers): hipcc's allocator otherwise rotates accumulators through fresh
// registers across the unrolled taps (40 accumulator quads for 16 in the 64 x 64-per-wave kernel; spills at 128 x 64 per wave).  An asm
// statement is opaque to the hazard recogniser (cdna_hip_programming.md 5.7): here the producers of a / b are ds_reads hipcc waits for
// itself, the next writer of c is an MFMA taking it whole as C (0 wait states), and the first non-MFMA reader sits behind a workgroup barrier.
template <bool F16> __device__ __forceinline__ void mfma16_tied(const s16x8_t& a, const s16x8_t& b, f32x4_t& c) {
    if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

#ifndef LP_PIPE_SCHED
#define LP_PIPE_SCHED 1          // 0: no order pinning (A/B builds)
#endif
#ifndef LP_PIPE_SETPRIO
#define LP_PIPE_SETPRIO 1        // s_setprio 1 around the first-half MFMA group (r04: 128^2 x 128 layer 49.8 -> 44 .. 47 us, others +-1 %); 0: A/B builds
#endif
#if LP_PIPE_SCHED
#define LP_PIPE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define LP_PIPE_SCHED_BARRIER() do {} while (0)
#endif

template <bool UPS, int WM, int WN, int MR, int NR, int PREC, int AIT>
__global__ __launch_bounds__(WM * WN * 64, (PREC == LP_PREC_BF16X3) ? 1 : 2)
void conv_pipe_kernel(Conv16Params p) {
    constexpr int ROWB = 64;                             // bytes per halo pixel / weight row of a 32-channel chunk
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    constexpr int NWAVE = WM * WN, BN = WN * NR * 16;
    constexpr int B_TAP = BN * ROWB;                     // one tap's weights (hi)
    constexpr int B_SLOT = B_TAP * (SPLIT ? 2 : 1);
    constexpr int NQW = B_TAP / 1024 / NWAVE;            // 1 KiB weight pieces per wave and tap (hi)
    constexpr int NBW = NQW * (SPLIT ? 2 : 1);           // weight DMA instructions per wave and tap
    constexpr int NAW = AIT * (SPLIT ? 2 : 1);           // halo DMA instructions per wave and chunk (always AIT pieces: a static count)
    constexpr int A_BYTES = AIT * NWAVE * 1024;          // one halo image (hi)
    constexpr int A_BUF = A_BYTES * (SPLIT ? 2 : 1);
    constexpr int MH = MR / 2;
    constexpr int HW = UPS ? 10 : 18;                    // halo width of a 16-pixel-wide patch
    static_assert(B_TAP % (1024 * NWAVE) == 0, "a tap's weights must be a whole number of DMA pieces per wave");
    static_assert(MR % 2 == 0, "the A fragments are read in two halves");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int TH = 1 << p.lTH;                           // (host: lTW == 4, lNB == 0)

    int tile_id, cob;
    if (!conv16_block(p, tile_id, cob)) return;          // (uniform: padding of the XCD-ordered grid)
    int t = tile_id;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int n0 = t / p.tiles_y;
    const int y0 = ty << p.lTH, x0 = tx << 4;
    const int co0 = cob * BN;
    const int HH = UPS ? (TH >> 1) + 2 : TH + 2;
    const int oy = UPS ? (y0 >> 1) - 1 : y0 - 1, ox = UPS ? (x0 >> 1) - 1 : x0 - 1;
    const int halo_px = HH * HW;

    unsigned char* const H_base = smem;                  // [halo 0][halo 1][slot 0][slot 1][slot 2]
    unsigned char* const B_base = smem + 2 * A_BUF;

    // ---- fragment addressing.  Tile rows are pixels of the TH x 16 patch in linear order: row m -> (py = m >> 4, px = m & 15); the 16 rows of
    // an MFMA row block are one patch row, so py is wave-uniform per block and px = lane & 15.
    const int kb = lane >> 4, l15 = lane & 15;
    int acol[3];                                         // per-lane byte offset of the fragment slice inside its halo row, per kernel column
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int hx = UPS ? ((l15 + dx - 1) >> 1) + 1 : l15 + dx;
        acol[dx] = hx * ROWB + ((kb ^ ((hx >> 1) & 3)) << 4);
    }
    // The address of an A fragment read = a per-lane base per kernel column (acol[dx] + the wave's first patch row) + a COMPILE-TIME offset
    // (row block, kernel row, halo width 18 | 10): three base registers, the rest rides in the ds_read offset field.  (wm * MR is even, so
    // the upsampled row ((py + dy - 1) >> 1) + 1 splits into wm*MR/2 and a constant, too.)
    const int py0 = (wm * MR) & (TH - 1);                // patch row of the wave's first row block (uniform)
    int abase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) abase[dx] = acol[dx] + (UPS ? (py0 >> 1) : py0) * HW * ROWB;
    int b_addr[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) b_addr[nr] = (wn * (NR * 16) + nr * 16 + l15) * ROWB + ((kb ^ ((l15 >> 1) & 3)) << 4);

    f32x4_t acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- DMA descriptors.  Halo piece q = k*NWAVE + wave covers halo pixels 16q .. 16q+15: lane -> pixel 16q + lane/4, LDS slot lane%4,
    // which holds the 8-channel group slot ^ key(hx).  Weight piece q = j*NWAVE + wave covers rows 16q .. 16q+15 of the tap's [BN][32] tile.
    int a_off[AIT];
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int hp = (k * NWAVE + wave) * 16 + (lane >> 2);
        const int hx = hp % HW, hy = hp / HW;
        const int iy = oy + hy, ix = ox + hx;
        const bool inb = (hp < halo_px) && (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
        a_off[k] = inb ? (((n0 * p.Hin + iy) * p.Win + ix) * p.C8 + (((lane & 3) ^ ((hx >> 1) & 3)) << 3)) : -1;
    }
    int w_off[NQW];
#pragma unroll
    for (int j = 0; j < NQW; ++j) {
        const int r = (j * NWAVE + wave) * 16 + (lane >> 2);
        w_off[j] = (co0 + r) * p.CinP + (((lane & 3) ^ ((r >> 1) & 3)) << 3);
    }
    const uint16_t* zero16 = (const uint16_t*)lp_zero_page_pipe;
    const size_t tap_stride = (size_t)p.CoutP * p.CinP;

    auto issue_a = [&](int chunk, int buf) {
        const int c0 = chunk * 32;
        const unsigned dst = (unsigned)(uintptr_t)(H_base + buf * A_BUF);
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const bool ok = a_off[k] >= 0;
            const size_t off = (size_t)(ok ? a_off[k] + c0 : 0);
            const unsigned d = dst + (unsigned)((k * NWAVE + wave) * 1024);
            lp_glds16(ok ? (p.a_hi + off) : zero16, d);
            if (SPLIT) lp_glds16(ok ? (p.a_lo + off) : zero16, d + A_BYTES);
        }
    };
    auto issue_w = [&](int chunk, int tap) {             // tap: compile-time after unrolling
        const size_t base = (size_t)tap * tap_stride + (size_t)chunk * 32;
        const unsigned dst = (unsigned)(uintptr_t)(B_base + (tap % 3) * B_SLOT);
#pragma unroll
        for (int j = 0; j < NQW; ++j) {
            const unsigned d = dst + (unsigned)((j * NWAVE + wave) * 1024);
            lp_glds16(p.w_hi + base + w_off[j], d);
            if (SPLIT) lp_glds16(p.w_lo + base + w_off[j], d + B_TAP);
        }
    };

    // Register plan (MR = 8, one-plane modes): 128 accumulators + the A fragments of the two half taps (fa0, fa1: 16 + 16) + ONE set of B
    // fragments (fbc: 16) = 176 of the 256 VGPRs two co-resident workgroups allow.  The B fragments of the next tap replace the current
    // ones IN PLACE while the second half multiplies (both halves run output-column-major: fbc[nr] is dead after its four MFMAs of the
    // second half and is needed again only nr groups into the next tap's first half).  A first version kept two B sets: 256 VGPRs + scratch
    // spills, whose reloads made hipcc put `vmcnt(0)` into the loop -- draining the DMA pipeline this kernel exists for.
    s16x8_t fbc[NR], fa0[MH], fa1[MH];
    s16x8_t fbcl[NR], fa0l[MH], fa1l[MH];
    auto load_b1 = [&](int nr, int tap) {
        const unsigned char* Bk = B_base + (tap % 3) * B_SLOT;
        fbc[nr] = *(const s16x8_t*)(Bk + b_addr[nr]);
        if (SPLIT) fbcl[nr] = *(const s16x8_t*)(Bk + B_TAP + b_addr[nr]);
    };
    auto load_a = [&](s16x8_t (&fa)[MH], s16x8_t (&fal)[MH], int half, int tap, const unsigned char* const (&Hb)[3]) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            const int k = half * MH + i;                                  // row block of the wave: compile-time
            const int off = (UPS ? (((k + dy - 1) >> 1) + 1) : (k + dy)) * HW * ROWB;
            fa[i] = *(const s16x8_t*)(Hb[dx] + off);
            if (SPLIT) fal[i] = *(const s16x8_t*)(Hb[dx] + A_BYTES + off);
        }
    };
    auto mm_col = [&](int half, int nr, const s16x8_t (&fa)[MH], const s16x8_t (&fal)[MH]) {
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            f32x4_t& c = acc[half * MH + i][nr];
            if (SPLIT) {
                // (bf16x3: 512-register budget, part of the fragments live in AGPRs -- the copies hipcc puts in front of an asm statement
                //  are VALU writes whose MFMA-operand wait states nobody would pad: the compiler-scheduled builtin here)
                c = mfma16(fal[i], fbc[nr], c);
                c = mfma16(fa[i], fbcl[nr], c);
                c = mfma16(fa[i], fbc[nr], c);
            } else {
                mfma16_tied<F16>(fa[i], fbc[nr], c);
            }
        }
    };

    const int nch = p.CinP / 32;
    // ---- prologue: three taps of weights and the first halo; everything lands before the first fragment read
    issue_w(0, 0); issue_w(0, 1); issue_w(0, 2);
    issue_a(0, 0);
    lp_wait_vm0();
    lp_barrier_raw();
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) load_b1(nr, 0);
    {
        const unsigned char* const H0[3] = {H_base + abase[0], H_base + abase[1], H_base + abase[2]};
        load_a(fa0, fa0l, 0, 0, H0);
    }

    for (int chunk = 0; chunk < nch; ++chunk) {
        const int hc = (chunk & 1) * A_BUF, hn = A_BUF - hc;
        const unsigned char* const Hc[3] = {H_base + hc + abase[0], H_base + hc + abase[1], H_base + hc + abase[2]};
        const unsigned char* const Hn[3] = {H_base + hn + abase[0], H_base + hn + abase[1], H_base + hn + abase[2]};
        const bool has_next = chunk + 1 < nch;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // first half of the tap's MFMAs; the second half's A fragments are read ahead of them.  The sched_barriers pin the ORDER of the
            // read groups and the MFMA groups (inside a group the compiler schedules freely): left alone, hipcc re-used one register quad
            // for successive A fragments and emitted read -> lgkmcnt(0) -> 4 MFMAs -> read ..., an exposed LDS round trip per 4 MFMAs.
            load_a(fa1, fa1l, 1, tap, Hc);
            LP_PIPE_SCHED_BARRIER();
            if (LP_PIPE_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) mm_col(0, nr, fa0, fa0l);
            if (LP_PIPE_SETPRIO) __builtin_amdgcn_s_setprio(0);
            LP_PIPE_SCHED_BARRIER();
            // ---- sync point of step g = (chunk, tap): the weights of tap g+1 (every wave's pieces) have landed, and every wave is done with
            // the slot of tap g (its fragments were read one step ago and consumed by the MFMAs above) and, at tap 0, with the other halo buffer.
            // Allowed outstanding = DMA instructions issued after those of tap g+1: the weights of tap g+2 (issued one step ago) and, at taps
            // 1 and 2, the next chunk's halo (issued at tap 0).
            if (tap == 1 || tap == 2) { if (has_next) lp_wait_vm<NBW + NAW>(); else lp_wait_vm<NBW>(); }
            else if (tap == 7) { if (has_next) lp_wait_vm<NBW>(); else lp_wait_vm0(); }
            else if (tap == 8) { if (has_next) lp_wait_vm<NBW>(); }
            else lp_wait_vm<NBW>();
            lp_barrier_raw();
            if (tap + 3 < 9) issue_w(chunk, tap + 3);
            else if (has_next) issue_w(chunk + 1, tap + 3 - 9);
            if (tap == 0 && has_next) issue_a(chunk + 1, (chunk + 1) & 1);
            const bool more = (tap < 8) || has_next;
            // (fa0 was consumed by the MFMAs above the barrier: the next tap's first half goes straight into it)
            if (tap < 8) load_a(fa0, fa0l, 0, tap + 1, Hc);
            else if (has_next) load_a(fa0, fa0l, 0, 0, Hn);
            LP_PIPE_SCHED_BARRIER();
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                mm_col(1, nr, fa1, fa1l);
                if (more) load_b1(nr, tap < 8 ? tap + 1 : 0);            // the next tap's B fragment into the registers that just became free
                LP_PIPE_SCHED_BARRIER();
            }
        }
    }
    // ---- epilogue (conv_common.h; 16 rows through LDS at a time: 4.3 KB of scratch per wave)
    conv16_epilogue<WM, WN, MR, NR, PREC, 1>(p, acc, smem, wave, wm, wn, lane, n0, y0, x0, co0, 1, tile_id);
}

// ------------------------------------------------------------------------------------------------------------------
// 1x1 convs (the encoders' pointwise layers on flattened pixels, the generator's / critic's skip convs): the same idea without taps.
// conv_dma_kernel<1> stages a 32-channel chunk ONE stage ahead and drains vmcnt(0) per chunk, so a workgroup's life is (Cin / 32) DMA
// round trips in sequence -- 2 .. 8 of them for the layers that carry the bytes (Cin 64 .. 256 at 65 k .. 262 k pixels), each 1 - 2 us under
// load against 0.1 - 0.35 us of MFMA work: the layers ran at 2 TB/s of algorithmic traffic.  Here a chunk = {BM x 64 B of activations, BN x 64 B
// of weights} goes into a ring of R slots, the first R chunks are issued back to back in the prologue (everything a small layer needs is in
// flight at once), chunk c+R-1 is issued behind the barrier of chunk c, and the only waits are counted `vmcnt`.  One barrier per chunk.
template <int WM, int WN, int PREC, int R>
__global__ __launch_bounds__(256, (PREC == LP_PREC_BF16X3) ? 1 : 2)
void conv1x1_pipe_kernel(Conv16Params p) {
    constexpr int ROWB = 64, MR = 4, NR = 4, NWAVE = 4;
    static_assert(WM * WN == NWAVE, "four waves");
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3), F16 = (PREC == LP_PREC_F16);
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;          // one chunk (hi)
    constexpr int NPA = A_BYTES / 1024 / NWAVE, NPB = B_BYTES / 1024 / NWAVE;
    constexpr int NPC = (NPA + NPB) * (SPLIT ? 2 : 1);              // DMA instructions per wave and chunk
    constexpr int SLOT = (A_BYTES + B_BYTES) * (SPLIT ? 2 : 1);    // [A hi][A lo][B hi][B lo]
    constexpr int B_OFF = A_BYTES * (SPLIT ? 2 : 1);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int NBv = 1 << p.lNB;
    int t = (int)blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int ng = t / p.tiles_y;
    const int n0 = ng << p.lNB, y0 = ty << p.lTH, x0 = tx << p.lTW;
    const int co0 = blockIdx.y * BN;

    const int kb = lane >> 4, l15 = lane & 15;
    const int fkey = (kb ^ ((l15 >> 1) & 3)) << 4;                   // swizzled slot of this lane's fragment slice (row & 15 == lane & 15)
    int a_addr[MR], b_addr[NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) a_addr[mr] = (wm * 64 + mr * 16 + l15) * ROWB + fkey;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) b_addr[nr] = B_OFF + (wn * 64 + nr * 16 + l15) * ROWB + fkey;

    f32x4_t acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // DMA descriptors: piece q = k*4 + wave covers tile rows 16q .. 16q+15 (lane -> row 16q + lane/4, slot lane%4 holding channel group
    // slot ^ key(row)); the key only depends on the lane (16q is a multiple of 8)
    const int g8 = ((lane & 3) ^ ((lane >> 3) & 3)) << 3;
    int a_off[NPA], w_off[NPB];
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
        const int m = (k * NWAVE + wave) * 16 + (lane >> 2);
        int nb, py, px;
        tile_row_linear(m, p.lTH, p.lTW, nb, py, px);
        const int n = n0 + nb, iy = y0 + py, ix = x0 + px;
        const bool inb = (nb < NBv) && (n < p.N) && (iy < p.H) && (ix < p.W);
        a_off[k] = inb ? (((n * p.H + iy) * p.W + ix) * p.C8 + g8) : -1;
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) w_off[j] = (co0 + (j * NWAVE + wave) * 16 + (lane >> 2)) * p.CinP + g8;
    const uint16_t* zero16 = (const uint16_t*)lp_zero_page_pipe;

    auto issue = [&](int chunk, int slot) {
        const int c0 = chunk * 32;
        const bool cok = (c0 + g8) < p.C8;
        const unsigned dst = (unsigned)(uintptr_t)(smem + slot * SLOT);
#pragma unroll
        for (int k = 0; k < NPA; ++k) {
            const bool ok = cok && a_off[k] >= 0;
            const size_t off = (size_t)(ok ? a_off[k] + c0 : 0);
            const unsigned d = dst + (unsigned)((k * NWAVE + wave) * 1024);
            lp_glds16(ok ? (p.a_hi + off) : zero16, d);
            if (SPLIT) lp_glds16(ok ? (p.a_lo + off) : zero16, d + A_BYTES);
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const unsigned d = dst + B_OFF + (unsigned)((j * NWAVE + wave) * 1024);
            lp_glds16(p.w_hi + w_off[j] + c0, d);
            if (SPLIT) lp_glds16(p.w_lo + w_off[j] + c0, d + B_BYTES);
        }
    };

    const int nch = p.CinP / 32;
    const int npro = nch < R - 1 ? nch : R - 1;                      // chunks 0 .. R-2 in the prologue; chunk c+R-1 behind the barrier of chunk c
    for (int c = 0; c < npro; ++c) issue(c, c);
    int slot = 0;
    for (int c = 0; c < nch; ++c) {
        // chunk c must have landed; issued after it so far: chunks c+1 .. min(c+R-2, nch-1)
        const int newer = min(R - 2, nch - 1 - c);
        if (newer >= R - 2) lp_wait_vm<(R - 2) * NPC>();
        else if (R > 3 && newer == 1) lp_wait_vm<NPC>();
        else if (R > 4 && newer == 2) lp_wait_vm<2 * NPC>();
        else lp_wait_vm0();
        lp_barrier_raw();                                            // every wave's pieces of chunk c are in; everyone is done with chunk c-1
        if (c + R - 1 < nch) issue(c + R - 1, (slot + R - 1) % R);   // into the slot chunk c-1 used
        const unsigned char* S = smem + slot * SLOT;
        s16x8_t fa[MR], fb[NR], fal[MR], fbl[NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            fa[mr] = *(const s16x8_t*)(S + a_addr[mr]);
            if (SPLIT) fal[mr] = *(const s16x8_t*)(S + A_BYTES + a_addr[mr]);
        }
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            fb[nr] = *(const s16x8_t*)(S + b_addr[nr]);
            if (SPLIT) fbl[nr] = *(const s16x8_t*)(S + B_BYTES + b_addr[nr]);
        }
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                if (SPLIT) {
                    acc[mr][nr] = mfma16(fal[mr], fb[nr], acc[mr][nr]);
                    acc[mr][nr] = mfma16(fa[mr], fbl[nr], acc[mr][nr]);
                }
                acc[mr][nr] = mfma16t<F16>(fa[mr], fb[nr], acc[mr][nr]);
            }
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    conv16_epilogue<WM, WN, MR, NR, PREC, 1>(p, acc, smem, wave, wm, wn, lane, n0, y0, x0, co0, NBv, (int)blockIdx.x);
}

template <int WM, int WN, int PREC, int R>
static int launch_pipe1x1(Conv16Params& p, hipStream_t stream) {
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    constexpr size_t SLOT = (size_t)(BM + BN) * 64 * (SPLIT ? 2 : 1);
    choose_tile(BM, p.N, p.H, p.W, &p.lTH, &p.lTW, &p.lNB);
    const int TH = 1 << p.lTH, TW = 1 << p.lTW, NBv = 1 << p.lNB;
    p.tiles_x = (p.W + TW - 1) / TW; p.tiles_y = (p.H + TH - 1) / TH;
    const int tiles = p.tiles_x * p.tiles_y * ((p.N + NBv - 1) / NBv);
    p.hit = 0; p.a_dbuf = 1; p.ksplit = 1;
    size_t lds = R * SLOT;
    const size_t epi = (size_t)4 * 16 * (64 + 4) * sizeof(float);
    if (lds < epi) lds = epi;
    if (lds > 160 * 1024) return 0;
    p.stats_rows = 0;
    if (p.stats) {
        const long long rows = (long long)tiles * WM;
        const bool ok = (NBv * TH * TW == BM) && (TH * TW >= 64) && (p.H % TH == 0) && (p.W % TW == 0) && ((p.Cout & 3) == 0) &&
                        rows * p.Cout * 3 <= p.stats_cap;
        if (ok) p.stats_rows = (p.H * p.W) / 64; else p.stats = nullptr;
    }
    auto kern = conv1x1_pipe_kernel<WM, WN, PREC, R>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipGetDevice failed");
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_dev = dev;
    }
    p.xcd_map = 0; p.ntiles = tiles; p.nco = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(tiles, (p.Cout + BN - 1) / BN), dim3(256), lds, stream, p);
    const int rc = lp_check_launch("conv1x1_pipe");
    return rc ? rc : 1;
}

template <int PREC>
static int pipe1x1_dispatch(Conv16Params& p, int bn, hipStream_t s) {
    constexpr int R = (PREC == LP_PREC_BF16X3) ? 3 : 4;      // 96 KB (one workgroup per CU at hi+lo planes) / 64 KB (two)
    return bn == 64 ? launch_pipe1x1<4, 1, PREC, R>(p, s) : launch_pipe1x1<2, 2, PREC, R>(p, s);
}

// -> 1: launched; 0: not covered; < 0: error
int lp_conv1x1_pipe_launch(Conv16Params& p, int prec, hipStream_t s) {
    // OFF by default (round 4, profiles/r04_conv1x1_pipe.txt): the pointwise layers turned out to be bound by HBM traffic and by how many
    // workgroups a CU holds, not by the chunk-by-chunk DMA round trips -- the ring costs LDS (64 / 96 KB per workgroup against 32 / 64 KB),
    // i.e. a resident workgroup per CU, and the kernel measured 5 - 25 % SLOWER than conv_dma_kernel<1> on the layers that carry the bytes
    // (fp16: 79 -> 87 us at 262 k pixels x 128 -> 256; bf16x3: 137 -> 167 us), equal on the long contractions.  1: on; 2: every shape (tests).
    static const int on = getenv("LP_CONV1X1_PIPE") ? atoi(getenv("LP_CONV1X1_PIPE")) : 0;
    if (!on || p.grouped || p.CinP % 32 || p.H < 2 || p.W < 2) return 0;
    const long long pix = (long long)p.N * p.H * p.W;
    const int bn = p.Cout <= 64 ? 64 : 128;
    const long long wgs = ((pix + (bn == 64 ? 255 : 127)) / (bn == 64 ? 256 : 128)) * ((p.Cout + bn - 1) / bn);
    static const int maxk = getenv("LP_CONV1X1_PIPE_MAXK") ? atoi(getenv("LP_CONV1X1_PIPE_MAXK")) : 256;
    // small layers: the 8-wave paths of conv_dma_kernel; long contractions (> maxk input channels): its 64-channel stages
    if (on != 2 && (wgs < 256 || p.CinP > maxk)) return 0;
    if (prec == LP_PREC_BF16) return pipe1x1_dispatch<LP_PREC_BF16>(p, bn, s);
    if (prec == LP_PREC_BF16X3) return pipe1x1_dispatch<LP_PREC_BF16X3>(p, bn, s);
    if (prec == LP_PREC_F16) return pipe1x1_dispatch<LP_PREC_F16>(p, bn, s);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
template <bool UPS, int WM, int WN, int MR, int NR, int PREC, int AIT>
static int launch_pipe(Conv16Params& p, hipStream_t stream) {
    constexpr int NWAVE = WM * WN, BM = WM * MR * 16, BN = WN * NR * 16;
    constexpr bool SPLIT = (PREC == LP_PREC_BF16X3);
    constexpr size_t A_BUF = (size_t)AIT * NWAVE * 1024 * (SPLIT ? 2 : 1), B_SLOT = (size_t)BN * 64 * (SPLIT ? 2 : 1);
    p.lTW = 4; p.lNB = 0;
    p.lTH = 0; while ((16 << (p.lTH + 1)) <= BM) ++p.lTH;               // TH = BM / 16
    const int TH = 1 << p.lTH;
    const int HH = UPS ? TH / 2 + 2 : TH + 2, HWc = UPS ? 10 : 18;
    const int pieces = (HH * HWc + 15) / 16;
    if ((pieces + NWAVE - 1) / NWAVE > AIT) return 0;
    p.tiles_x = (p.W + 15) / 16; p.tiles_y = (p.H + TH - 1) / TH;
    p.hit = AIT; p.a_dbuf = 1; p.ksplit = 1;
    const int tiles = p.tiles_x * p.tiles_y * p.N;
    size_t lds = 2 * A_BUF + 3 * B_SLOT;
    const size_t epi = (size_t)NWAVE * 16 * (NR * 16 + 4) * sizeof(float);
    if (lds < epi) lds = epi;
    if (lds > 160 * 1024) return 0;
    p.stats_rows = 0;
    if (p.stats) {      // as conv_dma.hip: full tiles, every wave's MR*16 rows inside one image, coalesced epilogue, room in the buffer
        constexpr int WR = MR * 16;
        const long long rows = (long long)tiles * WM;
        const bool ok = (p.H % TH == 0) && (p.W % 16 == 0) && ((p.Cout & 3) == 0) && rows * p.Cout * 3 <= p.stats_cap;
        if (ok) p.stats_rows = (p.H * p.W) / WR; else p.stats = nullptr;
    }
    auto kern = conv_pipe_kernel<UPS, WM, WN, MR, NR, PREC, AIT>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return lp_set_error(LP_ERR_HIP, "hipGetDevice failed");
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return lp_set_error(LP_ERR_HIP, "hipFuncSetAttribute failed");
        attr_dev = dev;
    }
    const dim3 grid = conv16_grid(p, tiles, (p.Cout + BN - 1) / BN);          // (sets p.xcd_map / ntiles / nco: BEFORE p is copied into the launch)
    hipLaunchKernelGGL(kern, grid, dim3(NWAVE * 64), lds, stream, p);
    const int rc = lp_check_launch("conv_pipe");
    return rc ? rc : 1;
}

template <int PREC>
static int pipe_dispatch(Conv16Params& p, int ups, int mr, int bn, hipStream_t s) {
    if (bn == 64) {      // <= 64 output channels: 256 x 64 tiles, four waves of 64 x 64 (the layers bound by their output store: two workgroups per CU)
        return ups ? launch_pipe<true, 4, 1, 4, 4, PREC, 2>(p, s) : launch_pipe<false, 4, 1, 4, 4, PREC, 6>(p, s);
    }
    if (ups) return mr == 8 ? launch_pipe<true, 2, 2, 8, 4, PREC, 2>(p, s) : launch_pipe<true, 2, 2, 4, 4, PREC, 1>(p, s);
    return mr == 8 ? launch_pipe<false, 2, 2, 8, 4, PREC, 6>(p, s) : launch_pipe<false, 2, 2, 4, 4, PREC, 3>(p, s);
}

// -> 1: launched; 0: this layer is not covered (the caller runs conv_dma_kernel); < 0: error.  p: as filled by lp_conv16_fwd_stats.
int lp_conv_pipe_launch(Conv16Params& p, int ups, int prec, hipStream_t s) {
    static const int on = getenv("LP_CONV_PIPE") ? atoi(getenv("LP_CONV_PIPE")) : 1;
    static const int mr_env = getenv("LP_CONV_PIPE_MR") ? atoi(getenv("LP_CONV_PIPE_MR")) : 0;      // 4 | 8 forces the rows per wave (test knob)
    static const int n64 = getenv("LP_CONV_PIPE_N64") ? atoi(getenv("LP_CONV_PIPE_N64")) : 1;       // 0: <= 64-channel outputs stay on conv_dma_kernel
    static const int x3 = getenv("LP_CONV_PIPE_X3") ? atoi(getenv("LP_CONV_PIPE_X3")) : 0;
    if (!on || p.grouped) return 0;
    // bf16x3 (hi + lo planes: 144 KB of LDS, one workgroup per CU, compiler-scheduled MFMAs): measured SLOWER than conv_dma_kernel on the
    // 64^2 .. 128^2 layers (115 vs 86 .. 93 us, profiles/r04_conv_pipe_micro.txt) -- the strict mode stays on conv_dma_kernel unless forced
    // (LP_CONV_PIPE_X3=1, or the test knob LP_CONV_PIPE_MR)
    if (prec == LP_PREC_BF16X3 && !x3 && !mr_env) return 0;
    if ((p.C8 & 31) || p.CinP % 32 || p.W < 16 || p.H < 16) return 0;
    int mr = 0, bn = 128;
    const long long px_tiles = (long long)((p.W + 15) / 16) * p.N;
    if (p.Cout <= 64) {
        if (!n64 || p.Cout < 32) return 0;                 // (the 4- / 3-channel head and RGB layers have their own kernels)
        const long long t = px_tiles * ((p.H + 15) / 16);
        if (t < 400 && !mr_env) return 0;
        bn = 64; mr = 4;
    } else {
        if (p.Cout < 128) return 0;
        const long long coblk = (p.Cout + 127) / 128;
        const long long t256 = px_tiles * ((p.H + 15) / 16) * coblk, t128 = px_tiles * ((p.H + 7) / 8) * coblk;
        // 256 x 128 tiles (128 x 64 outputs per wave: fewest LDS fragment bytes per MFMA) when they still give every CU two workgroups;
        // 128 x 128 tiles while those cover the chip; smaller layers keep the split-K path of conv_dma_kernel
        if (t256 >= 480) mr = 8;
        else if (t128 >= 200) mr = 4;
        if (mr_env == 8 || mr_env == 4) mr = mr_env;          // (test knob: any grid size)
        if (!mr) return 0;
    }
    if (prec == LP_PREC_BF16) return pipe_dispatch<LP_PREC_BF16>(p, ups, mr, bn, s);
    if (prec == LP_PREC_BF16X3) return pipe_dispatch<LP_PREC_BF16X3>(p, ups, mr, bn, s);
    if (prec == LP_PREC_F16) return pipe_dispatch<LP_PREC_F16>(p, ups, mr, bn, s);
    return 0;
}
