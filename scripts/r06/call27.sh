#!/bin/bash
# round 6, call 27: lp_proj_score_* / lp_image_prep_* (ABI 11): kernel + module parity, launch inventory, step time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_discriminator_criterions.py tests/test_vgg_planes.py tests/test_full_size_parity.py tests/test_streams_gpu.py tests/test_train_entry_gpu.py tests/test_train_step.py tests/test_metatrain_step.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
WORKLOAD=metatrain python scripts/aten_sites.py > $O/aten_sites.txt 2> $O/aten_sites.err; head -1 $O/aten_sites.txt
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-drive 2> $O/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', d['ms_per_step'])" | tee -a $O/ab.txt
done
