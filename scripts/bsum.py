import json, subprocess, sys, os
# usage: python scripts/bsum.py TAG [ENV=VAL ...] -- bench args
args = sys.argv[1:]
tag = args[0]; rest = args[1:]
sep = rest.index('--') if '--' in rest else len(rest)
env = dict(os.environ); env.update(dict(a.split('=', 1) for a in rest[:sep]))
out = subprocess.run([sys.executable, 'bench.py'] + rest[sep + 1:], env=env, capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith('{')]
if not line:
    print(tag, 'FAILED', out.stderr[-800:]); sys.exit(0)
d = json.loads(line[-1])
r = d.get('roofline') or {}
w = d.get('roofline_conv_wgrad') or {}
print(f"{tag}: {d['value']} img/s  {d['ms_per_step']} ms/step  mode={d['config'].get('launch_mode')} | conv_igemm {r.get('achieved')} TF/s avg {r.get('avg_launch_us')} us x{r.get('launches')} | wgrad {w.get('achieved')} TF/s avg {w.get('avg_launch_us')} us")
