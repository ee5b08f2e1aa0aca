"""Full-size (256x256, real channel counts) parity of the HIP generator against the CPU oracle on identical inputs --
the gate of BASELINE.json's north_star: outputs within 1e-3 rel-L2 of the fp32 CPU path.  Random-init weights (torch default
initialisers, a few power iterations so that sigma is meaningful), B = 1 to keep the CPU oracle at a few seconds.

Gradients are gated TIE-MASKED: a pre-activation within rounding distance of 0 flips its ReLU between two correct implementations
(a forward error eps flips a fraction ~0.8*eps of the units of a layer), and under the white-noise loss used here a fraction f of
flipped units moves a weight gradient by ~sqrt(f) -- 5e-3 already for fp32-class arithmetic (eps 2.5e-6), which says nothing about
the kernels.  So the oracle is evaluated a second time on the HIP path's OWN activation pattern (read back from the 16-bit operand
planes the decoder saved) and the gradients are compared on that common piecewise-linear branch; the untied figure is printed
beside it.  Outputs are always compared against the true-ReLU oracle."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# (mode, gate on outputs, gate on tie-masked gradients): bf16 operands miss the 1e-3 output gate (~1.3e-3) and are only bounded
@pytest.mark.parametrize('prec,tol_out,tol_grad', [(1, 1e-4, 1e-4), (2, 1e-3, 2e-3), (0, 3e-3, 2e-2)])
def test_generator_256_vs_oracle(prec, tol_out, tol_grad):
    from latent_pose_reenactment_amd.nn import Generator
    from oracle import lp_oracle as O
    torch.manual_seed(0)
    G = Generator('zero', 3, 4, 64, 512, 512, 256, 'in', 4, 2, 256, prec=prec)
    with torch.no_grad():
        G.constant.constant.normal_()
    G = G.cuda().train()
    e = torch.randn(1, 512); p = torch.randn(1, 256)
    with torch.no_grad():                                   # settle the power iteration (random u, v give a poor sigma)
        for _ in range(5):
            G({'embeds': e.cuda(), 'pose_embedding': p.cuda()})
    sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    for k, v in sd.items():
        if k.endswith('weight_orig') or k.endswith('.bias') or k.endswith('.constant'):
            v.requires_grad_(True)
    ec, pc = e.cuda().requires_grad_(True), p.cuda().requires_grad_(True)
    dd = {'embeds': ec, 'pose_embedding': pc}
    G._debug = {}
    G(dd)
    masks = [(pl > 0).permute(0, 3, 1, 2).cpu() for pl in G._debug['relu_planes']]      # int16 > 0 <=> value > 0
    g = torch.Generator().manual_seed(1)
    r1, r2 = torch.randn(1, 3, 256, 256, generator=g), torch.randn(1, 1, 256, 256, generator=g)
    ((dd['fake_rgbs'] * r1.cuda()).sum() + (dd['fake_segm'] * r2.cuda()).sum()).backward()
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd_u = {k: v.detach().clone() for k, v in sd.items()}         # (the power iteration updates u, v in place: one copy per oracle run)

    def oracle_grads(state, relu_masks):
        for k, v in state.items():
            if v.dtype.is_floating_point:
                v.grad = None
        eo, po = e.clone().requires_grad_(True), p.clone().requires_grad_(True)
        rgb, segm = O.generator_forward(state, eo, po, num_channels=64, max_num_channels=512, image_size=256, train=True, relu_masks=relu_masks)
        ((rgb * r1).sum() + (segm * r2).sum()).backward()
        g = {'d_embeds': eo.grad, 'd_pose': po.grad}
        for k, prm in G.named_parameters():
            if k.endswith('weight_orig') and state[k].grad is not None:
                g[k] = state[k].grad
        return rgb, segm, g
    for k, v in sd_u.items():
        if k.endswith('weight_orig') or k.endswith('.bias') or k.endswith('.constant'):
            v.requires_grad_(True)
    rgb, segm, g_true = oracle_grads(sd, None)
    _, _, g_tied = oracle_grads(sd_u, masks)
    mine = {'d_embeds': ec.grad, 'd_pose': pc.grad}
    mine.update({k: prm.grad for k, prm in G.named_parameters() if k in g_true})
    errs = {'fake_rgbs': rel(dd['fake_rgbs'], rgb), 'fake_segm': rel(dd['fake_segm'], segm)}
    gerr = {k: rel(mine[k], g_tied[k]) for k in g_tied}
    gerr_untied = {k: rel(mine[k], g_true[k]) for k in g_true}
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:4]
    print(f'[parity-256] prec={prec}: outputs {errs}; tie-masked grads worst {[(k, round(v, 6)) for k, v in worst]}; '
          f'untied worst {max(gerr_untied.values()):.3e}')
    assert all(v < tol_out for v in errs.values()), errs
    assert all(v < tol_grad for v in gerr.values()), worst
