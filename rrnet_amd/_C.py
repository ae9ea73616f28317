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
        global LIB_PATH
        LIB_PATH = os.environ.get("RRNET_HIP_LIB", LIB_PATH)     # kernel experiments: an alternative build of the library
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


_CTYPE = {"int": c_int, "float": c_float, "double": c_double, "size_t": c_size_t, "long": c_long,
          "hipStream_t": c_void_p, "void": None}
_SIGS = None


def header_signatures():
    """Parses include/rrnet_hip.h (shipped next to the library as rrnet_amd/rrnet_hip.h when
    installed, otherwise <repo>/include) -> {name: (restype, [argtypes])}.  The header is the
    single source of truth for the ABI; pointers map to void*."""
    import re
    global _SIGS
    if _SIGS is not None:
        return _SIGS
    cands = [os.path.join(_HERE, "rrnet_hip.h"), os.path.join(os.path.dirname(_HERE), "include", "rrnet_hip.h")]
    path = next(p for p in cands if os.path.exists(p))
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    sigs = {}
    for m in re.finditer(r"(?m)^\s*(const\s+char\s*\*|int\s|size_t\s|void\s|double\s)\s*(rr_\w+|_nms)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else _CTYPE[ret.strip()]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(c_void_p)
                else:
                    toks = a.replace("const", "").split()
                    argtypes.append(_CTYPE[toks[0]])
        sigs[name] = (restype, argtypes)
    _SIGS = sigs
    return sigs


def fn(name, argtypes=None, restype=None):
    """Entry point `name` with the prototype declared in include/rrnet_hip.h."""
    f = getattr(lib(), name, None)
    if f is None:
        raise RRNetHipError("librrnet_hip.so does not export %s (stale build?)" % name)
    if f.argtypes is None:
        sig = header_signatures().get(name)
        if sig is None:
            raise RRNetHipError("%s is not declared in include/rrnet_hip.h" % name)
        f.restype, f.argtypes = sig[0], sig[1]
    return f


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RRNetHipError("rrnet_amd ops run on the MI355X only: got a %s tensor (no CPU fallback)" % t.device)
