"""Diagnosis (round 5): which operand-selection forms of the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) give wrong
results while another stream runs matrix-core kernels?  Every form is checked against the unpacked instruction on the same registers
(scripts/diag/canary.hip), alone and beside (a) lp_conv16_fwd (LDS-DMA + MFMA), (b) a torch fp16 matmul (the library's MFMA GEMM), (c) an
element-wise kernel.  usage: python scripts/pk_forms_probe.py [runs=4]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from latent_pose_reenactment_amd import hipops as ops  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
_so, _src = os.path.join(ROOT, 'scripts', 'diag', '_canary.so'), os.path.join(ROOT, 'scripts', 'diag', 'canary.hip')
if not os.path.exists(_so) or os.path.getmtime(_so) < os.path.getmtime(_src):          # (diagnosis kernels: built on demand, not part of liblp_hip.so)
    import subprocess
    subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', _src, '-o', _so], check=True)
lib = ctypes.CDLL(_so)
lib.canary_form_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
FORMS = ['v_pk_fma_f32', 'v_pk_fma_f32 op_sel:[1,0,0]', 'v_pk_fma_f32 op_sel:[0,1,0]', 'v_pk_fma_f32 op_sel:[0,0,1]', 'v_pk_fma_f32 op_sel_hi:[0,1,1]',
         'v_pk_fma_f32 op_sel_hi:[1,0,1]', 'v_pk_fma_f32 op_sel_hi:[1,1,0]', 'v_pk_fma_f32 op_sel_hi:[0,1,0]', 'v_pk_mul_f32', 'v_pk_mul_f32 op_sel:[1,0]',
         'v_pk_mul_f32 op_sel:[0,1]', 'v_pk_mul_f32 op_sel_hi:[0,1]', 'v_pk_mul_f32 op_sel_hi:[1,0]', 'v_pk_add_f32', 'v_pk_add_f32 op_sel:[1,0]',
         'v_pk_add_f32 op_sel:[0,1]', 'v_pk_add_f32 op_sel_hi:[0,1]', 'v_pk_add_f32 op_sel_hi:[1,0]', 'v_pk_mov_b32 op_sel:[0,0]', 'v_pk_mov_b32 op_sel:[1,0]',
         'v_pk_mov_b32 op_sel:[0,1]', 'v_pk_mov_b32 op_sel:[1,1]', 'v_fma_mix_f32 op_sel:[1,0,0]', 'v_fma_mix_f32 op_sel:[0,1,0]', 'v_fma_mix_f32']
torch.manual_seed(0)
x = torch.randn(8, 64, 64, 256, device='cuda')
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
pk = ops.pack_weights(w, 0, 2)
a = ops.act_pack(x, pro=0, prec=2)
A = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
B = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
junk = torch.empty(64 * 1024 * 1024, device='cuda')
side = torch.cuda.Stream()
out = torch.zeros(32, device='cuda', dtype=torch.int32)
torch.matmul(A, B)
torch.cuda.synchronize()


def noise(kind):
    with torch.cuda.stream(side):
        for _ in range(6):
            if kind == 'conv16':
                ops.conv16(a, pk, ksize=3, prec=2)
            elif kind == 'matmul':
                torch.matmul(A, B)
            elif kind == 'elementwise':
                junk.mul_(1.0001)


for kind in ('alone', 'conv16', 'matmul', 'elementwise'):
    bad = []
    for form, name in enumerate(FORMS):
        tot = torch.zeros(32, dtype=torch.int64)
        for _ in range(runs):
            out.zero_()
            torch.cuda.synchronize()
            if kind != 'alone':
                noise(kind)
            rc = lib.canary_form_launch(form, out.data_ptr(), 1024, 300, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
            tot += out.cpu().to(torch.int64) & 0xFFFFFFFF
        if int(tot[0]):
            bad.append(f'{name}: {int(tot[0])} (quarter-waves {tot[1:5].tolist()}, [low, high] half {tot[8:10].tolist()})')
    print(f'[pk-forms] {"alone" if kind == "alone" else "beside " + kind}: ' + (f'all {len(FORMS)} forms agree with the unpacked instructions' if not bad else 'WRONG: ' + ' | '.join(bad)), flush=True)
