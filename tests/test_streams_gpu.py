"""Concurrent branches (latent_pose_reenactment_amd/streams.py) must not change the result: one eager meta-training iteration (default.yaml
workload at 128 px: both encoders trained, 6 criterions) from the same initial state, once on ONE stream (LP_OVERLAP=0) and once with the
pose encoder beside the identity encoder and the VGG criterions beside the discriminator pass, must produce the same losses and the same
gradients of every parameter -- to the run-to-run spread of the one-stream step itself (the crop-and-resize backward scatters with float
atomics, like the reference's grid_sample, so two runs differ in the last bits; a missing cross-stream dependency or a tensor re-used
while another stream still reads it shows up as an O(1) error).  Also with every optional branch switched on."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
pytestmark = pytest.mark.gpu
KNOBS = ('LP_OVERLAP', 'LP_OVERLAP_GWGRAD', 'LP_OVERLAP_EARLY', 'LP_OVERLAP_ENCODERS', 'LP_OVERLAP_CRITERIONS', 'LP_OVERLAP_OPTIMIZER', 'LP_OVERLAP_WGRAD', 'LP_OVERLAP_TARGETS', 'LP_OVERLAP_PREPARE', 'LP_OVERLAP_DPASSES', 'LP_OVERLAP_REAL', 'LP_OVERLAP_EBWD')


def _iteration(monkeypatch, env, finetune=False):
    import bench
    from latent_pose_reenactment_amd.nn import fused_grad_accumulation
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    args = bench.make_args(128, 8, 'cuda:0', 1, 0, 'f16', finetune=finetune)
    args.num_labels = 100
    tm, opt_G, opt_D, holycow = bench.build(args)
    data, target = bench.synthetic_batch(args, 8, seed=500)
    all_data, losses_G, losses_D = tm(data, target)
    loss_G = sum(v for v in losses_G.values() if torch.is_tensor(v))
    loss_D = sum(v for v in losses_D.values() if torch.is_tensor(v))
    opt_G.zero_grad()
    with fused_grad_accumulation():
        loss_G.backward(retain_graph=True)
    out = {'loss.' + k: v.detach().clone() for k, v in {**losses_G, **losses_D}.items() if torch.is_tensor(v)}
    out.update({'fwd.' + k: all_data[k].detach().clone() for k in ('fake_rgbs', 'embeds', 'pose_embedding') if k in all_data})
    for name, mod in (('E', tm.embedder), ('G', tm.generator)):
        out.update({f'{name}.{k}': p.grad.detach().clone() for k, p in mod.named_parameters() if p.grad is not None})
    opt_D.zero_grad()
    with fused_grad_accumulation():
        loss_D.backward()
    out.update({f'D.{k}': p.grad.detach().clone() for k, p in tm.discriminator.named_parameters() if p.grad is not None})
    out.update({f'buf.{k}': b.detach().clone() for k, b in tm.embedder.named_buffers() if b.dtype.is_floating_point})
    torch.cuda.synchronize()
    return out


def _spread(a, b):
    """relative L2 difference of two runs per group (E / G / D gradient vectors, losses, forward outputs, BatchNorm buffers), each group taken
    as ONE vector: several BatchNorm biases of the encoders have a true gradient of exactly 0 (they feed a conv followed by a train-mode
    BatchNorm), so a per-tensor relative figure would compare rounding noise with rounding noise"""
    num, den = {}, {}
    for k in a:
        g = k.split('.')[0]
        num[g] = num.get(g, 0.0) + float((a[k].double() - b[k].double()).pow(2).sum())
        den[g] = den.get(g, 0.0) + float(b[k].double().pow(2).sum())
    return {g: (num[g] / max(den[g], 1e-60)) ** 0.5 for g in num}


def test_concurrent_branches_reproduce_the_one_stream_iteration(monkeypatch):
    one = _iteration(monkeypatch, {'LP_OVERLAP': '0'})
    again = _iteration(monkeypatch, {'LP_OVERLAP': '0'})
    floor = _spread(again, one)
    default = _iteration(monkeypatch, {})
    everything = _iteration(monkeypatch, {'LP_OVERLAP_WGRAD': '1', 'LP_OVERLAP_TARGETS': '1'})
    late = _iteration(monkeypatch, {'LP_OVERLAP_TARGETS': '2', 'LP_OVERLAP_REAL': '0'})      # target halves beside the generator; real pass beside the other two
    assert one.keys() == default.keys() == everything.keys() == late.keys()
    fmt = lambda sp: ', '.join(f'{g} {v:.1e}' for g, v in sorted(sp.items()))
    print(f'[streams] one stream, run to run: {fmt(floor)}')
    for name, run in (('default branches', default), ('all branches', everything), ('late targets, real pass with the others', late)):
        sp = _spread(run, one)
        print(f'[streams] {name} vs one stream: {fmt(sp)}')
        for g, d in sp.items():
            # (E, G: the run-to-run spread is itself random -- 1.7e-3 / 2.6e-4 typically; a missing dependency shows up as 0.1 .. 1)
            assert d <= max(20 * floor[g], 1e-2 if g in ('E', 'G') else 1e-5), (name, g, d, floor[g])


def test_fine_tuning_iteration_with_its_default_branches_reproduces_one_stream(monkeypatch):
    """(round 6) the fine-tuning step takes the criterions / prepare / dpasses / real branches by default: same losses, generator and critic
    gradients as on one stream"""
    one = _iteration(monkeypatch, {'LP_OVERLAP': '0'}, finetune=True)
    again = _iteration(monkeypatch, {'LP_OVERLAP': '0'}, finetune=True)
    floor = _spread(again, one)
    default = _iteration(monkeypatch, {}, finetune=True)
    assert one.keys() == default.keys()
    sp = _spread(default, one)
    fmt = lambda d: ', '.join(f'{g} {v:.1e}' for g, v in sorted(d.items()))
    print(f'[streams] fine-tuning, one stream run to run: {fmt(floor)}; default branches vs one stream: {fmt(sp)}')
    for g, d in sp.items():
        assert d <= max(20 * floor[g], 1e-2 if g in ('E', 'G') else 1e-5), (g, d, floor[g])


def _train_step_grads(monkeypatch, env):
    import bench
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    args = bench.make_args(128, 8, 'cuda:0', 1, 0, 'f16', finetune=False)
    args.num_labels = 100
    tm, opt_G, opt_D, holycow = bench.build(args)
    data, target = bench.synthetic_batch(args, 8, seed=500)
    _, losses_G, losses_D = holycow.train_step(tm, data, target, opt_G, opt_D, args)
    torch.cuda.synchronize()
    out = {'loss.' + k: v.detach().clone() for k, v in {**losses_G, **losses_D}.items() if torch.is_tensor(v)}
    for name, mod in (('E', tm.embedder), ('G', tm.generator), ('D', tm.discriminator)):
        out.update({f'{name}.{k}': p.grad.detach().clone() for k, p in mod.named_parameters() if p.grad is not None})
    return out


def test_embedder_backward_beside_the_discriminator_backward_reproduces_the_step(monkeypatch):
    """LP_OVERLAP_EBWD (default ON since round 6: -0.7 ms, profiles/r06_stream_overlap.txt): train_step cuts the autograd graph behind the embedder and runs the encoders' backward
    on a side stream beside loss_D.backward -- same gradients as the uncut step"""
    plain = _train_step_grads(monkeypatch, {'LP_OVERLAP_EBWD': '0'})
    again = _train_step_grads(monkeypatch, {'LP_OVERLAP_EBWD': '0'})
    cut = _train_step_grads(monkeypatch, {'LP_OVERLAP_EBWD': '1'})
    assert plain.keys() == cut.keys()
    floor, sp = _spread(again, plain), _spread(cut, plain)
    print('[streams] EBWD: run to run', {g: f'{v:.1e}' for g, v in floor.items()}, '| cut vs plain', {g: f'{v:.1e}' for g, v in sp.items()})
    for g, d in sp.items():
        assert d <= max(20 * floor[g], 1e-2 if g in ('E', 'G') else 1e-5), (g, d, floor[g])


def test_generator_weight_gradients_deferred_to_the_critic_backward_stream_reproduce_the_step(monkeypatch):
    """LP_OVERLAP_GWGRAD: the generator's weight-gradient launches are recorded during loss_G.backward (hipops.wgrad_defer) and issued on the critic-backward
    stream beside the encoders' backward (runners/holycow.py) -- the same kernels on the same operands, one contribution per parameter: same gradients"""
    plain = _train_step_grads(monkeypatch, {'LP_OVERLAP_GWGRAD': '0'})
    again = _train_step_grads(monkeypatch, {'LP_OVERLAP_GWGRAD': '0'})
    late = _train_step_grads(monkeypatch, {'LP_OVERLAP_GWGRAD': '1'})
    assert plain.keys() == late.keys()
    floor, sp = _spread(again, plain), _spread(late, plain)
    same = all(torch.equal(late[k], plain[k]) for k in plain if k.startswith('G.'))
    print('[streams] GWGRAD: run to run', {g: f'{v:.1e}' for g, v in floor.items()}, '| deferred vs plain', {g: f'{v:.1e}' for g, v in sp.items()}, '| generator gradients bit-identical:', same)
    for g, d in sp.items():
        assert d <= max(20 * floor[g], 1e-2 if g in ('E', 'G') else 1e-5), (g, d, floor[g])
