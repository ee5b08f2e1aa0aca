"""Diagnosis (round 5): what exactly is disturbed in a small kernel that runs beside the LDS-DMA conv kernels on another stream?
scripts/diag/canary.hip holds four canaries: 16-byte global loads in the power iteration's pattern over a buffer whose dword i holds i;
a static LDS array; registers held across a sleep; the power iteration's arithmetic from registers.  Each runs quiet and beside
lp_conv16_fwd (8 x 64 x 64 x 256 -> 256) and reports mismatch counts by quarter-wave and by dword of the 16-byte load.
usage: python scripts/canary_probe.py [runs=6]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from latent_pose_reenactment_amd import hipops as ops  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
_so, _src = os.path.join(ROOT, 'scripts', 'diag', '_canary.so'), os.path.join(ROOT, 'scripts', 'diag', 'canary.hip')
if not os.path.exists(_so) or os.path.getmtime(_so) < os.path.getmtime(_src):          # (diagnosis kernels: built on demand, not part of liblp_hip.so)
    import subprocess
    subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', _src, '-o', _so], check=True)
lib = ctypes.CDLL(_so)
lib.canary_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
torch.manual_seed(0)
x = torch.randn(8, 64, 64, 256, device='cuda')
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
prec = {'f16': 2, 'bf16': 0, 'bf16x3': 1}[os.environ.get('LP_PREC', 'f16')]
pk = ops.pack_weights(w, 0, prec)
a = ops.act_pack(x, pro=0, prec=prec)
side = torch.cuda.Stream()
rows, C = 2048, 4608
buf = torch.arange(rows * C, device='cuda', dtype=torch.int32)
out = torch.zeros(32, device='cuda', dtype=torch.int32)
torch.cuda.synchronize()
NOISE = os.environ.get('CANARY_NOISE', 'conv16')
junk = torch.empty(64 * 1024 * 1024, device='cuda')

import struct  # noqa: E402
table = torch.frombuffer(bytearray(struct.pack('<QQii', buf.data_ptr(), out.data_ptr(), rows, C)), dtype=torch.uint8).cuda()
lib.canary_table_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

lib.canary_pk_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for which, name, blocks, reps in ((20, 'v_pk_fma_f32', 1024, 300), (21, 'v_pk_fma_f32 op_sel:[0,1,0]', 1024, 300), (22, 'v_pk_fma_f32 op_sel_hi:[1,0,1]', 1024, 300), (10, 'flat loads', 0, 6), (11, 'global loads via table', 0, 6), (0, 'loads', 0, 6), (1, 'lds', 2048, 400), (2, 'registers', 2048, 200), (3, 'arithmetic', 2048, 200)):
    for noisy in (False, True):
        tot = torch.zeros(32, dtype=torch.int64)
        first = None
        for _ in range(runs):
            out.zero_()
            torch.cuda.synchronize()
            if noisy:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        if NOISE == 'conv16':
                            ops.conv16(a, pk, ksize=3, prec=prec)
                        else:
                            junk.mul_(1.0001)
            if which >= 20:
                rc = lib.canary_pk_launch(which - 20, out.data_ptr(), blocks, reps, torch.cuda.current_stream().cuda_stream)
            elif which >= 10:
                rc = lib.canary_table_launch(which - 10, table.data_ptr(), rows, reps, torch.cuda.current_stream().cuda_stream)
            else:
                rc = lib.canary_launch(which, buf.data_ptr(), rows, C, out.data_ptr(), blocks, reps, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
            o = out.cpu().to(torch.int64) & 0xFFFFFFFF
            if int(o[0]) and first is None:
                first = [hex(int(v)) for v in o[16:20]]
            tot += o
        print(f'[canary] {name:32s} {"beside " + NOISE if noisy else "alone":14s}: mismatches {int(tot[0])}; by quarter-wave {tot[1:5].tolist()}; by dword {tot[8:12].tolist()}'
              + (f'; first (index, got, want, lane) {first}' if first else ''), flush=True)
