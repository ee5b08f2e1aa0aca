"""3x3 dense conv family INSIDE a captured step: kernel durations from a one-step rocprofv3 breakdown (scripts/step_breakdown.py) against the
algorithmic FLOPs of the same launches (bench.py --shapes table, kind conv_igemm, two instrumented steps) -> the in-graph roofline figure
that bench.py prints beside its live eager-event one.
usage: python scripts/in_graph_conv.py <step_breakdown.csv> <conv_shapes.csv> > profiles/r06_conv3x3_in_graph.json
Family = conv_pipe_kernel (all variants) + conv_dma_kernel<3, ...> and <2, ...> (the phase forms of the x2-upsampled convs, round 6) except the instantiations whose last template flag (GH) is true -- the identity
encoder's GROUPED 3x3 convs (round 6: the dense bf16x3 <= 64-channel layers of the default assignment run conv_dma_kernel<3, false, 4, 1, 4, 4, ..,
false> and belong to the family) -- + the split-K finishes (splitk_reduce_kernel).  The figure prices ALGORITHMIC flops; bf16x3 launches execute three
MFMAs per MAC (bench.py prints the matrix-pipe work beside it)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (source_stamp: ties the figure to the tree it was measured on)

bd, shapes = sys.argv[1], sys.argv[2]
t_ms, launches, parts = 0.0, 0, {}
for line in open(bd):
    if line.startswith('#') or line.startswith('kernel,'):
        continue
    r = next(csv.reader([line]))
    name, n, ms = r[0], int(r[1]), float(r[2])
    fam = None
    if name.startswith('conv_pipe_kernel'):
        fam = 'conv_pipe_kernel'
    elif (name.startswith('conv_dma_kernel<3,') or name.startswith('conv_dma_kernel<2,')) and not name.rstrip().endswith('true>'):          # (last template flag GH: the grouped instantiations)
        fam = 'conv_dma_kernel<3> / <2> (bf16x3 layers, phase forms of the x2-upsampled convs, small maps, leftovers)'
    elif name.startswith('splitk_reduce_kernel'):
        fam = 'splitk_reduce_kernel'
    if fam:
        t_ms += ms; launches += n
        p = parts.setdefault(fam, [0, 0.0]); p[0] += n; p[1] += ms
fl = 0.0
for r in csv.DictReader(open(shapes)):
    if r['kind'] == 'conv_igemm':
        fl += float(r['TFLOPs']) * float(r['total_us']) * 1e-6      # TF/s x s = TFLOP (two instrumented steps)
tflop_step = fl / 2
ach = tflop_step / (t_ms * 1e-3)
print(json.dumps({'source': 'rocprofv3 kernel trace of one hipGraph replay (scripts/step_breakdown.py) + bench.py --shapes', 'ms_per_step': round(t_ms, 3),
                  'launches_per_step': launches, 'algorithmic_tflop_per_step': round(tflop_step, 3), 'achieved_tflops': round(ach, 1),
                  'frac': round(ach / 2500.0, 4), 'peak_tflops': 2500.0, 'stamp': bench.source_stamp(), 'by_kernel': {k: {'launches': v[0], 'ms': round(v[1], 3)} for k, v in parts.items()}}))
