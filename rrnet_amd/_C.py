"""ctypes binding of librrnet_hip.so (include/rrnet_hip.h).

The product path has no CPU fallback: if the HIP library is missing or an entry point is
absent, importing/using an op raises — loudly — instead of computing something else."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librrnet_hip.so")
_lib = None

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
c_long, c_double = ctypes.c_long, ctypes.c_double


class RRNetHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RRNetHipError(
                "librrnet_hip.so not found at %s — build it with `python rrnet_amd/csrc/build.py` "
                "(there is no CPU fallback for the HIP path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.rr_last_error.restype = ctypes.c_char_p
        L.rr_abi_version.restype = c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RRNetHipError("%s failed (%d): %s" % (what, rc, lib().rr_last_error().decode()))


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def fn(name, argtypes, restype=c_int):
    f = getattr(lib(), name, None)
    if f is None:
        raise RRNetHipError("librrnet_hip.so does not export %s (stale build?)" % name)
    if f.argtypes is None:
        f.argtypes = argtypes
        f.restype = restype
    return f


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RRNetHipError("rrnet_amd ops run on the MI355X only: got a %s tensor (no CPU fallback)" % t.device)
