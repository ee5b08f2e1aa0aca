"""roctx ranges around the phases of a training step (tracing row; VERDICT r05 "missing 6"): with ``LP_ROCTX=1`` the host-side issue of
the embedder / generator / discriminator forward, the criterions, the two backward passes, the gradient exchanges, the optimizer steps and the
EMA update are bracketed by ``roctxRangePushA`` / ``roctxRangePop`` (libroctx64 of the ROCm installation, bound through ctypes: no torch
extension), so that ``rocprofv3 --marker-trace --kernel-trace -- python train.py ...`` groups the kernels of an EAGER step by phase.  Off by default:
a captured step (hipGraph) is issued once, at capture time, and its replays carry no host ranges -- there the kernel names (profiles/*_step_breakdown_*)
are the map.  ``rng(name)`` is a no-op context manager when tracing is off or the library is missing."""
import contextlib
import ctypes
import os

_ON = os.environ.get('LP_ROCTX', '0') != '0'
_LIB = None


def _lib():
    global _LIB, _ON
    if _LIB is None and _ON:
        for name in ('libroctx64.so', '/opt/rocm/lib/libroctx64.so', 'librocprofiler-sdk-roctx.so'):
            try:
                _LIB = ctypes.CDLL(name)
                _LIB.roctxRangePushA.argtypes = [ctypes.c_char_p]
                _LIB.roctxRangePushA.restype = ctypes.c_int
                _LIB.roctxRangePop.restype = ctypes.c_int
                break
            except (OSError, AttributeError):
                _LIB = None
        if _LIB is None:
            _ON = False
    return _LIB


def enabled() -> bool:
    return _ON and _lib() is not None


@contextlib.contextmanager
def rng(name: str):
    lib = _lib() if _ON else None
    if lib is None:
        yield
        return
    lib.roctxRangePushA(('lp:' + name).encode())
    try:
        yield
    finally:
        lib.roctxRangePop()
