#!/bin/bash
# round 5, call 34: with the power iteration exact beside the convs, does the fake->G pass still need strict operands for the 1e-3 gate?
O=$GRAFT_REPO_ROOT/gpurun_out/r05ac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
e = d['errors']
print(sys.argv[1], d['critic_fake_to_G_pass'], {k: float(f'{e[k]:.3g}') for k in ('loss.adversarial_G', 'fake_score_G', 'loss.feature_matching', 'loss.adversarial_D', 'fake_rgbs')})
PY
}
for rep in 1 2 3; do
  LP_D_GPASS_PREC=f16 timeout 600 python tests/test_metatrain_full_gpu.py $O/f16_$rep.json > $O/f16_$rep.log 2>&1; show "gpass=f16 run $rep" $O/f16_$rep.json | tee -a $O/gpass.txt
done
for f in 3 6; do
  LP_D_GPASS_FROM=$f timeout 600 python tests/test_metatrain_full_gpu.py $O/from$f.json > $O/from$f.log 2>&1; show "from=$f" $O/from$f.json | tee -a $O/gpass.txt
done
