"""Convergence comparison of the MFMA operand modes ("keep bf16x3 the default until a convergence comparison exists"):
the SAME training run -- same seed, same initial weights, same cycle of K synthetic batches -- in f16 (default) and bf16x3 (strict,
fp32-class), each in its own process; losses are logged per step and compared on windows.  A run on random-content synthetic frames
cannot say anything about sample quality; it can say whether the two arithmetic modes follow the same optimisation trajectory (they
must, if fp16 operand rounding is benign: same losses within the step-to-step noise, no divergence, no saturation-driven drift).
usage: convergence_compare.py finetune|metatrain [steps] [batches]            (child: ... --child MODE)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(workload, mode, steps, nb):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'latent_pose_reenactment_amd'))
    import torch
    import bench
    perturb = mode.endswith('+eps')          # control run: the strict mode again, initial generator weights perturbed by a relative 1e-6
    mode = mode.replace('+eps', '')
    args = bench.make_args(256, 8, 'cuda:0', 1, 0, mode, finetune=(workload == 'finetune'))
    if workload == 'finetune':
        args.generator = 'vector_pose_unsupervised_segmentation_noBottleneck'
    tm, opt_G, opt_D, holycow = bench.build(args)
    if perturb:
        g = torch.Generator(device='cuda').manual_seed(99)
        with torch.no_grad():
            for p_ in tm.generator.parameters():
                p_.mul_(1 + 1e-6 * torch.randn(p_.shape, generator=g, device=p_.device))
    batches = [bench.synthetic_batch(args, 8, seed=1000 + i) for i in range(nb)]
    step = holycow.GraphedTrainStep(tm, opt_G, opt_D, args, *batches[0], warmup_steps=3)      # (3 eager optimizer steps on batch 0 precede the capture, in both modes)
    log = []
    for it in range(steps):
        step.load_batch(*batches[it % nb])
        step()
        rec = {k: float(v) for k, v in step.losses_G.items() if torch.is_tensor(v)}
        rec.update({'D.' + k: float(v) for k, v in step.losses_D.items() if torch.is_tensor(v)})
        log.append(rec)
    finite = all(all(x == x and abs(x) < 1e30 for x in r.values()) for r in log)
    print('CHILD ' + json.dumps({'mode': mode + ('+eps' if perturb else ''), 'finite': finite, 'log': log}), flush=True)


def main():
    workload = sys.argv[1]
    if '--child' in sys.argv:
        i = sys.argv.index('--child')
        return child(workload, sys.argv[i + 1], int(sys.argv[2]), int(sys.argv[3]))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    runs = {}
    modes = ('bf16x3', 'bf16x3+eps', 'f16')
    for mode in modes:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), workload, str(steps), str(nb), '--child', mode], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('CHILD ')]
        if not line:
            print(r.stdout[-2000:], r.stderr[-3000:]); raise SystemExit(1)
        runs[mode] = json.loads(line[-1][6:])
    keys = sorted(runs['f16']['log'][0])
    print(f'# {workload}: {steps} steps over a cycle of {nb} synthetic batches of 8; same seed and initial weights; per window: mean (max) of every loss term')
    print('# runs: bf16x3 (strict) | bf16x3+eps = the strict mode with the initial generator weights perturbed by a relative 1e-6 (how far two runs of the')
    print('#       SAME arithmetic drift apart: the chaos floor of this adversarial optimisation) | f16 (default)')
    w = max(steps // 6, 1)
    for k in keys:
        print(f'## {k}')
        for s0 in range(0, steps, w):
            cells = []
            for m in modes:
                seg = [r[k] for r in runs[m]['log'][s0:s0 + w]]
                cells.append(f'{m} {sum(seg) / len(seg):9.4g} ({max(seg, key=abs):9.4g})')
            print(f'  {s0:4d}-{min(s0 + w, steps) - 1:4d} | ' + ' | '.join(cells))
    tot = lambda log, s, e: sum(sum(v for k, v in r.items() if not k.startswith('D.')) for r in log[s:e]) / (e - s)
    for m in modes:
        lg = runs[m]['log']
        peak = max(max(abs(v) for v in r.values()) for r in lg)
        print(f'# {m}: generator-side loss first -> last window {tot(lg, 0, w):.4g} -> {tot(lg, steps - w, steps):.4g}; all finite: {runs[m]["finite"]}; largest |loss term| of the run {peak:.4g}')


if __name__ == '__main__':
    main()
