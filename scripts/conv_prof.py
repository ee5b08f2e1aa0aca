"""Phase timing of the single-group conv_igemm pipeline (s_memtime counters of wave 0 / block 0: barrier wait, weight-DMA issue,
halo load issue, MFMA phase, halo conversion + LDS write, total).  Needs an instrumented library:
    scripts/build_variant.sh latent_pose_reenactment_amd/csrc/conv_igemm.hip probes/liblp_hip_dbg.so -DLP_DBG -DLP_PROF
    LP_LIB_OVERRIDE=probes/liblp_hip_dbg.so LP_CONV_PP=0 PREC=1 python scripts/conv_prof.py
(the counters live in the non-ping-pong FAST loop; LP_CONV_DBG=<bits> additionally ablates parts of the pipeline, see ConvParams)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latent_pose_reenactment_amd import hipops as ops, _lib
lib = _lib.lib()
prof = torch.zeros(6, dtype=torch.int64, device='cuda')
lib.lp_dbg_set_prof.argtypes = [ctypes.c_void_p]; lib.lp_dbg_set_prof(prof.data_ptr())
prec = int(os.environ.get('PREC', '0'))
for (n, h, w, cin, cout) in [(8, 32, 32, 512, 512), (8, 64, 64, 256, 256), (8, 256, 256, 64, 64)]:
    x = torch.randn(n, h, w, cin, device='cuda'); wgt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.02
    sc = torch.randn(n, cin, device='cuda'); sh = torch.randn(n, cin, device='cuda')
    pack = ops.pack_weights(wgt, 0, prec)
    for _ in range(3):
        ops.conv(x, pack, ksize=3, pro=1, scale=sc, shift=sh, prec=prec)
    torch.cuda.synchronize()
    v = prof.cpu().tolist()
    names = ['sync', 'dma_issue', 'halo_ld_issue', 'mfma', 'halo_write', 'total']
    print(f'prec={prec} {(n,h,w,cin,cout)}: ' + ' '.join(f'{nm}={x_}' for nm, x_ in zip(names, v)) + '  (s_memtime ticks of wave 0, block 0)')
