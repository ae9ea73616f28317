"""Thin, autograd-free wrappers over the C ABI (include/rrnet_hip.h).  Tensors are torch CUDA
tensors used purely as device-memory handles: logical shape NCHW with channels_last strides,
i.e. NHWC in memory; conv weights logical [K,C,R,S] with channels_last strides = OHWI."""
import torch

from rrnet_amd import _C

_P, _I, _F = _C.c_void_p, _C.c_int, _C.c_float
CL = torch.channels_last


def is_nhwc(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def to_nhwc(t):
    """Returns a tensor with the same logical NCHW shape whose memory is NHWC."""
    if is_nhwc(t):
        return t
    return t.contiguous(memory_format=CL)


def empty_nhwc(n, c, h, w, device, dtype=torch.float32):
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def zeros_nhwc(n, c, h, w, device, dtype=torch.float32):
    return torch.zeros((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def out_hw(h, w, r, s, stride, ph, pw):
    return (h + 2 * ph - r) // stride + 1, (w + 2 * pw - s) // stride + 1


def conv_fprop(x, w, bias=None, stride=1, pad=(0, 0), relu=False, want_stats=False):
    """x [N,C,H,W] (NHWC memory), w [K,C,R,S] (OHWI memory) -> y [N,K,P,Q] (NHWC memory)
    and, if want_stats, the per-block BatchNorm partial-sum slab (see rr_conv_fprop)."""
    _C.require_cuda(x, w, bias)
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    assert is_nhwc(x) and is_nhwc(w), "conv_fprop wants NHWC activations / OHWI weights"
    n, c, h, wd = x.shape
    k, c2, r, s = w.shape
    assert c == c2
    p, q = out_hw(h, wd, r, s, stride, pad[0], pad[1])
    y = empty_nhwc(n, k, p, q, x.device)
    slab = None
    if want_stats:
        nbytes = _C.fn("rr_conv_stat_slab_bytes")(n, p, q, k)
        slab = torch.empty(nbytes // 8, dtype=torch.float64, device=x.device)
    f = _C.fn("rr_conv_fprop")
    _C.check(f(_C.ptr(x), _C.ptr(w), _C.ptr(bias), _C.ptr(y), _C.ptr(slab), n, h, wd, c, k, r, s, stride,
               pad[0], pad[1], int(relu), _C.stream()), "rr_conv_fprop")
    return (y, slab) if want_stats else y


def conv_dgrad(dy, w, x_shape, stride=1, pad=(0, 0), out=None, accumulate=False):
    """dy [N,K,P,Q], w [K,C,R,S] -> dx [N,C,H,W]; with `out` and accumulate adds into it."""
    _C.require_cuda(dy, w)
    assert is_nhwc(dy) and is_nhwc(w)
    n, c, h, wd = x_shape
    k, c2, r, s = w.shape
    assert c == c2 and dy.shape[1] == k
    if out is None:
        out = empty_nhwc(n, c, h, wd, dy.device)
        accumulate = False
    assert is_nhwc(out)
    f = _C.fn("rr_conv_dgrad")
    _C.check(f(_C.ptr(dy), _C.ptr(w), _C.ptr(out), n, h, wd, c, k, r, s, stride, pad[0], pad[1],
               int(accumulate), _C.stream()), "rr_conv_dgrad")
    return out


def conv_wgrad(x, dy, dw, stride=1, pad=(0, 0)):
    """dw [K,C,R,S] (OHWI memory) += x (*) dy.  dw must be pre-zeroed / hold the running gradient."""
    _C.require_cuda(x, dy, dw)
    assert is_nhwc(x) and is_nhwc(dy) and is_nhwc(dw)
    n, c, h, wd = x.shape
    k, c2, r, s = dw.shape
    assert c == c2 and dy.shape[1] == k
    f = _C.fn("rr_conv_wgrad")
    _C.check(f(_C.ptr(x), _C.ptr(dy), _C.ptr(dw), n, h, wd, c, k, r, s, stride, pad[0], pad[1], _C.stream()),
             "rr_conv_wgrad")
    return dw
