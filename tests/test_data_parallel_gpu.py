"""Data-parallel path on the REAL train step (runners/holycow.py:241-242,249-250 replaced by parallel.GradReducer):
two ranks, each with half of a global batch, must produce the gradients of ONE rank on the whole batch -- generator + embedder
arena, discriminator arena including the row-sparse exchange of the label-embedding gradient -- and the re-cut hipGraph step
(asynchronous generator-side all-reduce overlapping the discriminator backward) must reproduce the eager data-parallel step.
Runs as two `gloo` processes sharing the box's single GPU (tests/dp_worker.py)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, 'tests', 'dp_worker.py')


def launch(world, mode, total, num_labels, steps, out, port):
    procs = []
    for r in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   LP_PREC='bf16x3', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(total), str(num_labels), str(steps), out], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return torch.load(out, weights_only=False)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('num_labels', [5, 6000])       # 6000 rows: the label-embedding gradient takes the row-sparse exchange
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path, num_labels):
    one = launch(1, 'eager', 4, num_labels, 1, str(tmp_path / 'one.pt'), 29611)
    two = launch(2, 'eager', 4, num_labels, 1, str(tmp_path / 'two.pt'), 29613)
    eg, ed = rel(two['gradG'], one['gradG']), rel(two['gradD'], one['gradD'])
    print(f'[dp] labels={num_labels}: 2 ranks vs 1 rank  generator-side gradient arena {eg:.2e}  discriminator arena {ed:.2e}')
    assert eg < 1e-4 and ed < 1e-4, (eg, ed)


def test_recut_graph_step_equals_eager_data_parallel_step(tmp_path):
    eager = launch(2, 'eager', 4, 6000, 3, str(tmp_path / 'e.pt'), 29615)
    graph = launch(2, 'graph', 4, 6000, 3, str(tmp_path / 'g.pt'), 29617)
    # (Adam moves elements whose true gradient is ~0 by +-lr on rounding noise, so single small tensors are not comparable between
    #  two runs; the last step's gradient arenas and the modules' whole state vectors are)
    eg, ed = rel(graph['gradG'], eager['gradG']), rel(graph['gradD'], eager['gradD'])
    vec = lambda sd: torch.cat([v.double().reshape(-1) for v in sd.values() if v.dtype == torch.float32])
    sg, sd_ = rel(vec(graph['G']), vec(eager['G'])), rel(vec(graph['D']), vec(eager['D']))
    print(f'[dp] graph vs eager after 3 steps on 2 ranks: gradient arenas {eg:.2e} / {ed:.2e}, state vectors {sg:.2e} / {sd_:.2e}')
    assert max(eg, ed) < 2e-3 and max(sg, sd_) < 2e-3, (eg, ed, sg, sd_)
    # issue order of the last re-cut step: G all-reduces asynchronous and BEFORE the discriminator-backward graph, waited for only before
    # optimizer_G's graph; the discriminator-side exchange between g2b and g3 (runners/holycow.py GraphedTrainStep.__call__)
    order = graph['order']
    names = [o if isinstance(o, str) else o[0] + ':' + o[1] for o in order]
    # (round 5) the generator-side exchange goes out as TWO buckets: the generator's slice of the arena right after g1 (its gradients are final
    # when the backward pass reaches the embedder's outputs), the encoders' slice after g1b (their backward), both before g2a
    i = {k: names.index(k) for k in ('g1', 'all_reduce:G-generator', 'g1b', 'all_reduce:G-embedder', 'g2a', 'wait_G', 'g2b', 'g3')}
    assert i['g1'] < i['all_reduce:G-generator'] < i['g1b'] < i['all_reduce:G-embedder'] < i['g2a'] < i['wait_G'] < i['g2b'] < i['g3'], order
    assert all(o[2] is True for o in order if not isinstance(o, str) and o[1].startswith('G')), order          # async_op=True: RCCL's own stream
    d_side = [j for j, n_ in enumerate(names) if n_ == 'all_reduce:D-side']
    assert d_side and all(i['g2b'] < j < i['g3'] for j in d_side), order


def test_bench_starts_its_own_ranks_from_plain_python():
    """`python bench.py --gpus 2` with NO launcher and NO WORLD_SIZE in the environment (how the driver starts the N = 1 point) must start
    its two ranks itself (bench.self_launch -> torch.distributed.run on 127.0.0.1) and print exactly ONE JSON line -- rank 0's -- on stdout.
    gloo backend: the two ranks share the box's single GPU (RCCL refuses that); the collectives' sequence is the same."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
                        '--image_size', '128', '--no-cpu-baseline', '--no-also', '--no-drive'], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['global_batch'] == 16 and j['config']['parallelism'] == 'dp2' and j['value'] > 0, j
    assert j['single_gpu_same_workload'].get('value', 0) > 0, j['single_gpu_same_workload']
    assert j['replicas']['bit_identical_parameters'] is True, j['replicas']          # the data-parallel invariant, checked on the parameters' bits


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: RCCL refuses two ranks on one device (the builder\'s box has one; the driver\'s node has eight)')
def test_bench_two_gpus_over_rccl_keeps_replicas_bit_identical():
    """ the FIRST execution of this code on the nccl (= RCCL) backend must not fail for a trivial reason: two ranks on
    two GPUs, the real meta-training step with the re-cut hipGraphs (generator bucket | encoders' bucket | discriminator-side exchange incl. the
    row-sparse label-embedding rows, ReduceOp.AVG), 1 warm-up + 2 timed + 2 instrumented steps -- the replicas' parameters must be bit-identical
    afterwards and the JSON line well-formed.  Skipped on a one-GPU box; no scaling figure is derived from it."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'nccl', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--no-also', '--no-drive'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['parallelism'] == 'dp2' and j['value'] > 0, j
    assert j['replicas']['bit_identical_parameters'] is True, j['replicas']
