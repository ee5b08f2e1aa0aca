"""CPU emulation of the address arithmetic of wgrad3_pipe_kernel (csrc/conv_wgrad.hip): the DMA pieces with their XOR-swizzled source chunks, the
transposing LDS reads (ds_read_b64_tr_b16 lane/element mapping as documented at the top of conv_wgrad.hip), the MFMA 16x16x32 fragment
layouts and the slab write -- every formula restated literally -- against a direct fp64 weight gradient.  No GPU needed: run before
spending GPU time on a change of the layout.   python scripts/wgrad3_pipe_emu.py"""
import itertools
import sys

import numpy as np


def emulate(N, H, W, Cin, Cout, UPS, seed=0):
    rng = np.random.default_rng(seed)
    Hin, Win = (H // 2, W // 2) if UPS else (H, W)
    C8, Co8 = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8
    CoP, CiP = (Cout + 63) // 64 * 64, (Cin + 63) // 64 * 64
    a = np.zeros((N, Hin, Win, C8)); a[..., :Cin] = rng.integers(-3, 4, (N, Hin, Win, Cin))
    d = np.zeros((N, H, W, Co8)); d[..., :Cout] = rng.integers(-3, 4, (N, H, W, Cout))
    HH, HW = (6, 10) if UPS else (10, 18)
    HALO_PX = HH * HW
    NAH = (HALO_PX + 7) // 8
    NAW = (NAH + 3) // 4
    HALO_E = NAH * 512                                # 16-bit elements
    tiles_x, tiles_y = (W + 15) // 16, (H + 7) // 8
    num_tiles = tiles_x * tiles_y * N
    part = np.zeros((9, CoP, CiP))
    bsum = np.zeros(CoP)
    for tco, tci in itertools.product(range(CoP // 64), range(CiP // 64)):
        co0, ci0 = tco * 64, tci * 64
        acc = np.zeros((4, 9, 4, 16, 16))             # wave, tap, mf, row, col
        accb = np.zeros((4, 16))
        for t in range(num_tiles):
            r = t
            tx = r % tiles_x; r //= tiles_x
            ty = r % tiles_y; n0 = r // tiles_y
            y0, x0 = ty * 8, tx * 16
            oy, ox = ((y0 >> 1) - 1, (x0 >> 1) - 1) if UPS else (y0 - 1, x0 - 1)
            lds = np.full(HALO_E + 128 * 64, np.nan)
            for wave in range(4):
                for lane in range(64):
                    lrow, lslot = lane >> 3, lane & 7
                    for i in range(NAW):
                        q = i * 4 + wave
                        if q < NAH:
                            hp = q * 8 + lrow
                            hy = hp // HW; hx = hp - hy * HW
                            ch = ci0 + ((lslot ^ (((hp >> 1) & 3) << 1)) << 3)
                            iy, ix = oy + hy, ox + hx
                            ok = hp < HALO_PX and ch < C8 and 0 <= iy < Hin and 0 <= ix < Win
                            src = a[n0, iy, ix, ch:ch + 8] if ok else np.zeros(8)
                            o = q * 512 + lane * 8
                            lds[o:o + 8] = src
                    for i in range(4):
                        q = i * 4 + wave
                        kp = q * 8 + lrow
                        ch = co0 + ((lslot ^ (((kp >> 1) & 3) << 1)) << 3)
                        yy, xx = y0 + (kp >> 4), x0 + (kp & 15)
                        ok = ch < Co8 and yy < H and xx < W
                        src = d[n0, yy, xx, ch:ch + 8] if ok else np.zeros(8)
                        o = HALO_E + q * 512 + lane * 8
                        lds[o:o + 8] = src

            def tr_read(addr_of_lane):
                """addr_of_lane: 64 element offsets -> [64 lanes][4]"""
                out = np.zeros((64, 4))
                for lane in range(64):
                    g, i = lane >> 4, lane & 15
                    for j in range(4):
                        s = g * 16 + 4 * j + (i >> 2)
                        out[lane, j] = lds[addr_of_lane[s] + (i & 3)]
                return out

            for wave in range(4):
                lanes = np.arange(64)
                G, sj, sq = lanes >> 4, (lanes & 15) >> 2, lanes & 3
                pxl = G * 4 + sj
                swA = (pxl >> 1) & 3
                aoff = [pxl * 128 + ((mf ^ swA) << 5) + sq * 8 for mf in range(4)]        # bytes
                bl, bs = [], []
                for dx in range(3):
                    col = (((pxl + dx - 1) >> 1) + 1) if UPS else pxl + dx
                    bl.append(col * 128 + sq * 8); bs.append(col >> 1)
                for ks in range(4):
                    fa = []
                    for mf in range(4):
                        v0 = tr_read((HALO_E * 2 + (ks * 32) * 128 + aoff[mf]) // 2)
                        v1 = tr_read((HALO_E * 2 + (ks * 32 + 16) * 128 + aoff[mf]) // 2)
                        fa.append(np.concatenate([v0, v1], 1))                          # [64][8]
                    for tap in range(9):
                        dy, dx = tap // 3, tap % 3
                        r0 = (((2 * ks + dy - 1) >> 1) + 1) if UPS else 2 * ks + dy
                        r1 = (((2 * ks + dy) >> 1) + 1) if UPS else 2 * ks + dy + 1
                        v0 = tr_read((r0 * HW * 128 + bl[dx] + ((wave ^ ((bs[dx] + r0) & 3)) << 5)) // 2)
                        v1 = tr_read((r1 * HW * 128 + bl[dx] + ((wave ^ ((bs[dx] + r1) & 3)) << 5)) // 2)
                        fb = np.concatenate([v0, v1], 1)
                        # MFMA 16x16x32: A lane l -> row l & 15, k = (l >> 4) * 8 + e; B lane l -> col l & 15, same k
                        A = np.zeros((16, 32)); B = np.zeros((32, 16))
                        for mf in range(4):
                            for l in range(64):
                                A[l & 15, (l >> 4) * 8:(l >> 4) * 8 + 8] = fa[mf][l]
                                B[(l >> 4) * 8:(l >> 4) * 8 + 8, l & 15] = fb[l]
                            assert not np.isnan(A).any() and not np.isnan(B).any()
                            acc[wave, tap, mf] += A @ B
                            if tap == 0 and mf == wave:
                                accb[wave] += A.sum(1)
        for wave in range(4):
            for tap in range(9):
                for mf in range(4):
                    part[tap, co0 + mf * 16:co0 + mf * 16 + 16, ci0 + wave * 16:ci0 + wave * 16 + 16] = acc[wave, tap, mf]
            if tci == 0:
                bsum[co0 + wave * 16:co0 + wave * 16 + 16] = accb[wave]
    # direct reference
    au = a.repeat(2, 1).repeat(2, 2) if UPS else a
    ap = np.pad(au, ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = np.zeros((9, CoP, CiP))
    for tap in range(9):
        dy, dx = tap // 3, tap % 3
        ref[tap, :Co8, :C8] = np.einsum('nyxo,nyxi->oi', d, ap[:, dy:dy + H, dx:dx + W, :])
    eb = np.abs(bsum[:Co8] - d.sum((0, 1, 2))).max()
    return np.abs(part - ref).max(), eb


def bank_report(UPS):
    """worst number of distinct addresses per LDS bank over the 32 lanes of a half-wave, for every transposing read of a tile"""
    HW = 10 if UPS else 18
    lanes = np.arange(64)
    G, sj, sq = lanes >> 4, (lanes & 15) >> 2, lanes & 3
    pxl = G * 4 + sj
    worst = 0
    def ways(addr):
        w = 0
        for half in (addr[:32], addr[32:]):
            banks = {}
            for a in half:
                for b in (a // 4 % 64, (a // 4 + 1) % 64):
                    banks.setdefault(b, set()).add(a // 4 if b == a // 4 % 64 else a // 4 + 1)
            w = max(w, max(len(v) for v in banks.values()))
        return w
    swA = (pxl >> 1) & 3
    for mf in range(4):
        worst = max(worst, ways(pxl * 128 + ((mf ^ swA) << 5) + sq * 8))
    for wave in range(4):
        for ks in range(4):
            for tap in range(9):
                dy, dx = tap // 3, tap % 3
                col = (((pxl + dx - 1) >> 1) + 1) if UPS else pxl + dx
                for r in ((((2 * ks + dy - 1) >> 1) + 1) if UPS else 2 * ks + dy, (((2 * ks + dy) >> 1) + 1) if UPS else 2 * ks + dy + 1):
                    worst = max(worst, ways(r * HW * 128 + col * 128 + sq * 8 + ((wave ^ (((col >> 1) + r) & 3)) << 5)))
    return worst


if __name__ == '__main__':
    bad = 0
    for case in [(1, 8, 16, 64, 64, 0), (2, 16, 32, 64, 128, 0), (1, 12, 24, 72, 40, 0), (1, 16, 32, 64, 64, 1), (2, 8, 16, 128, 64, 1), (1, 20, 12, 64, 64, 1)]:
        e, eb = emulate(*case)
        print(case, 'max |slab - ref| =', e, ' bias', eb)
        bad += (e != 0) or (eb != 0)
    print('LDS bank conflicts (distinct dwords per bank and half-wave; 1 = conflict free): plain', bank_report(False), ' upsampled', bank_report(True))
    sys.exit(1 if bad else 0)
