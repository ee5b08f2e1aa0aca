"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel: mean counter value per launch.

usage: python scripts/pmc_summary.py <counter_collection.csv> [...]  > summary.csv
       python scripts/pmc_summary.py --json <kernel name prefix> [--exclude <regex>] <counter_collection.csv> [...]  > traffic.json
--exclude: kernels of the family whose name matches the regex are left out (round 6: the GROUPED instantiations `conv_dma_kernel<3, ..., true>` of
the identity encoder are a different family with their own roofline entry; round 5's 3x3 figure contained them)
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for
wide coalesced reads on gfx950.  The --json form sums over every launch of the kernels whose name starts with the prefix (all tile
variants of one kernel family; several prefixes may be joined with '|') -> average HBM bytes per launch, which bench.py reports as roofline.traffic."""
import collections
import csv
import json
import os
import sys

args = sys.argv[1:]
prefix = None
exclude = None
if args and args[0] == '--json':
    prefix, args = args[1], args[2:]
    if args and args[0] == '--exclude':
        import re
        exclude, args = re.compile(args[1]), args[2:]
acc = collections.defaultdict(lambda: [0, 0.0])
fam = collections.defaultdict(lambda: [0, 0.0])
for path in args:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = (r.get('Kernel_Name') or r.get('Kernel Name') or '?')
            k = (name[:100] + ' grid=' + str(r.get('Grid_Size', '')), r['Counter_Name'])
            acc[k][0] += 1
            acc[k][1] += float(r['Counter_Value'])
            if prefix and any(pf in name.split('(')[0] for pf in prefix.split('|')) and not (exclude and exclude.search(name.split('(')[0])):
                fam[r['Counter_Name']][0] += 1
                fam[r['Counter_Name']][1] += float(r['Counter_Value'])
if prefix:
    out = {'kernel_family': prefix, 'counters': {c: {'launches': n, 'mean_per_launch': tot / max(n, 1)} for c, (n, tot) in fam.items()}}
    f_, w_ = fam.get('FETCH_SIZE'), fam.get('WRITE_SIZE')
    if f_ and w_ and f_[0] and w_[0]:
        out['hbm_bytes_per_launch'] = int((2 * f_[1] / f_[0] + w_[1] / w_[0]) * 1024)
        out['hbm_read_bytes_per_launch'] = int(2 * f_[1] / f_[0] * 1024)
        out['hbm_write_bytes_per_launch'] = int(w_[1] / w_[0] * 1024)
        out['excluded'] = exclude.pattern if exclude else None
        out['note'] = 'mean over all launches of the family in the profiled command: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B'
    b, g, m = fam.get('SQ_VALU_MFMA_BUSY_CYCLES'), fam.get('GRBM_GUI_ACTIVE'), fam.get('SQ_INSTS_VALU_MFMA_MOPS_F16')
    if b and g and g[1] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES: clocks with an MFMA in a SIMD's matrix pipe, summed over all 1024 SIMDs (checked: 16 per
        # v_mfma_f32_16x16x32 = SQ_INSTS_VALU_MFMA_MOPS / 32 instructions); GRBM_GUI_ACTIVE: active clocks summed over the 8 XCDs
        out['mfma_busy_fraction'] = round(b[1] / (g[1] / 8 * 256 * 4), 4)
    if b and m and m[1] > 0:
        out['mfma_busy_clocks_per_instruction'] = round(b[1] / (m[1] / 32), 2)   # MOPS counts 512-FLOP units; one 16x16x32 MFMA = 32 of them
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench          # source_stamp: ties the counters to the tree they were measured on
    out['stamp'] = bench.source_stamp()
    print(json.dumps(out))
    sys.exit(0)
print('kernel,counter,launches,mean_per_launch_raw,mean_bytes_per_launch_corrected')
rows = []
for (name, ctr), (n, tot) in acc.items():
    mean = tot / n
    corr = mean * 1024 * (2 if ctr == 'FETCH_SIZE' else 1) if ctr in ('FETCH_SIZE', 'WRITE_SIZE') else ''
    rows.append((tot, name, ctr, n, mean, corr))
for tot, name, ctr, n, mean, corr in sorted(rows, reverse=True):
    print(f'"{name[:120]}",{ctr},{n},{mean:.3f},{corr if corr == "" else int(corr)}')
