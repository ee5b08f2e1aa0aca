"""latent_pose_reenactment_amd -- MI355X (gfx950) native hot path of shrubb/latent-pose-reenactment.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (include/lp_hip.h) -> liblp_hip.so (built in-tree by build.py)
  _lib.py    ctypes binding of the C ABI (fails loudly when the library is missing)
  hipops.py  thin tensor-level wrappers (NHWC fp32 torch tensors in, raw pointers out)
  nn.py      autograd Functions + nn.Modules mirroring the reference plugin API and state_dict keys
"""
__all__ = ['build']
