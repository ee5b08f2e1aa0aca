import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_generator_module import load, make_gen, rel
z = load('generator_small.npz')
for warm in (0, 1):
    G = make_gen(z, prec=1)
    G.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd.')}, strict=True)
    G = G.cuda()
    if warm:
        G.eval()
        with torch.no_grad():
            G(dict(embeds=torch.from_numpy(z['embeds']).cuda(), pose_embedding=torch.from_numpy(z['pose']).cuda()))
    G.train()
    e = torch.from_numpy(z['embeds']).cuda().requires_grad_(True)
    p = torch.from_numpy(z['pose']).cuda().requires_grad_(True)
    dd = dict(embeds=e, pose_embedding=p)
    G(dd)
    errs = {'fake_rgbs': rel(dd['fake_rgbs'], z['train_fake_rgbs']), 'fake_segm': rel(dd['fake_segm'], z['train_fake_segm'])}
    loss = (dd['fake_rgbs'] * torch.from_numpy(z['r1']).cuda()).sum() + (dd['fake_segm'] * torch.from_numpy(z['r2']).cuda()).sum()
    loss.backward()
    errs['grad_embeds'] = rel(e.grad, z['grad_embeds']); errs['grad_pose'] = rel(p.grad, z['grad_pose'])
    for k, prm in G.named_parameters():
        errs['grad.' + k] = rel(prm.grad, z['grad.' + k])
    for k, v in G.state_dict().items():
        if k.endswith('_u') or k.endswith('_v'):
            errs['buf.' + k] = rel(v, z['sd_after.' + k])
    print('==== warm', warm)
    for k, v in errs.items():
        if not k.startswith('buf.') or v > 1e-6:
            print(f'{k:60s} {v:.3e}')
