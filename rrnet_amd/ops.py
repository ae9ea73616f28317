"""Thin, autograd-free wrappers over the C ABI (include/rrnet_hip.h).  Tensors are torch CUDA
tensors used purely as device-memory handles: logical shape NCHW with channels_last strides,
i.e. NHWC in memory; conv weights logical [K,C,R,S] with channels_last strides = OHWI."""
import os
import sys
import threading
import types

import torch

from rrnet_amd import _C

_P, _I, _F = _C.c_void_p, _C.c_int, _C.c_float
CL = torch.channels_last


class KernelTimer:
    """Optional live timing of individual conv launches with HIP events recorded on the launch
    stream (bench.py's roofline leg).  Off (None) in normal operation: no events, no overhead."""

    def __init__(self, only=None, every=1):
        self.records = []      # (kernel name, flops, start event, end event)
        self.bytes = {}        # kernel name -> algorithmic bytes (operands read once + result written once)
        self.only = only       # optional set of kernel names: the others run without events (each pair costs ~10 us)
        # time every `every`-th launch of a kernel name: an event pair is two tiny blit kernels on the stream (rocprofv3
        # shows them as __amd_rocclr_copyBuffer, 346 per step = 1.45 ms when every launch of the dominant kernel is
        # timed); a 1-in-4 systematic sample over the timed region (173 launches per step: the phase shifts every step)
        # keeps the average and costs a quarter
        self.every = max(int(every), 1)
        self.seen = {}

    def launch(self, name, flops, fn, detail=None, nbytes=0.0):
        if self.only is not None and name not in self.only:
            return fn()
        k = self.seen.get(name, 0)
        self.seen[name] = k + 1
        if k % self.every:
            return fn()
        self.bytes[name] = self.bytes.get(name, 0.0) + nbytes
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn()
        e.record()
        self.records.append((name, flops, s, e, detail))
        return rc

    def by_shape(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, s, e, detail in self.records:
            d = out.setdefault((name, detail), dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, s, e, _ in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
        return out


AUX_STREAMS = {}       # (device type, index) -> auxiliary HIP streams that may hold kernels writing gradient buffers


def join_aux_streams(device):
    """The current stream waits for everything enqueued on the auxiliary streams (weight gradients on their side
    stream): called before a collective is launched on, or a consumer reads, the flat gradient buffer."""
    if device.type != "cuda":
        return
    cur = torch.cuda.current_stream(device)
    for st in AUX_STREAMS.get((device.type, device.index), ()):
        if st != cur:
            cur.wait_stream(st)


TIMER = None
SYNC_WAIT_S = 0.0      # host time spent blocked in the RoI-count read (bench.py separates it from the enqueue time)


def _timed(name, flops, fn, detail=None, nbytes=0.0):
    if TIMER is None:
        return fn()
    return TIMER.launch(name, flops, fn, detail, nbytes)


_SMALL_TILES = 16
_MID_TILES = 48


def _igemm_name(kind, n_gemm, scalar, m_rows=1 << 30, stride=1):
    """Timer key = the HIP kernel instance that runs (tile width as picked in csrc/conv.hip: 128 columns, 64 for
    33..64-column layers, 32 for narrow ones and for layers with at most RR_CONV_SMALL_TILES 128x128 tiles)."""
    bn = 128 if n_gemm > 64 or (scalar and n_gemm > 32) else (64 if n_gemm > 32 else 32)
    if bn == 128 and not scalar and (kind == "fprop" or stride == 1):
        tiles = -(-m_rows // 128) * -(-n_gemm // 128)
        if tiles <= _SMALL_TILES:
            bn = 32
        elif tiles <= _MID_TILES:
            bn = 64
    return "conv_%s<BN=%d,%s>" % (kind, bn, "scalar" if scalar else "vec4")


def is_nhwc(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def is_phantom(t):
    """A bf16-only activation / gradient (phantom_f32): an fp32 tensor OBJECT without memory whose values live in its bf16 image.
    Recognised by IDENTITY: every such handle is a zero-stride view of the one per-device NaN stub (_PHANTOM_STUB), so the test is
    "does t view that storage" — views and tensors unpacked from an autograd node keep it, nothing else can have it (round 5 went
    by a signature: strides 0, offset 1, an 8-byte storage).  A view that lost the image link (the Python attribute) fails loudly
    in image_of(); a torch-native read of the handle returns NaN."""
    if t is None or t.dim() != 4 or t.stride() != (0, 0, 0, 0):
        return False
    stub = _PHANTOM_STUB.get(t.device)
    return stub is not None and t.untyped_storage().data_ptr() == stub.untyped_storage().data_ptr()


def image_of(t):
    hit = getattr(t, "_rr_b16", None)
    if hit is None:
        raise RuntimeError("a bf16-only tensor of shape %s reached a consumer without its bf16 image (a view / saved tensor whose "
                           "producer did not hand the image on: ops.b16_carry / b16_restore)" % (tuple(t.shape),))
    return hit[2]


def f32_of(t):
    """The fp32 tensor a consumer without a bf16 form needs: t itself, or — for a bf16-only tensor — its image widened into a FRESH
    tensor of the caller's (allocated on the current stream, never remembered on t: a copy cached on the tensor object in the forward
    outlived it until the backward, where the weight-gradient side stream read it while the main stream's allocator had already
    reused the block — found by the stress arm of tests/test_streams_gpu.py)."""
    if not is_phantom(t):
        return t
    img = image_of(t)
    if _PHANTOM_TRACE:
        import traceback
        key = (tuple(t.shape), traceback.format_stack(limit=4)[0].strip().split("\n")[0])
        PHANTOM_WIDENED[key] = PHANTOM_WIDENED.get(key, 0) + 1
    out = torch.empty_like(img, dtype=torch.float32)
    assert out.stride() == img.stride()
    _C.check(_C.fn("rr_from_bf16")(_C.ptr(img), _C.ptr(out), img.numel(), _C.stream()), "rr_from_bf16")
    return out


def to_nhwc(t, keep_phantom=False):
    """Returns a tensor with the same logical NCHW shape whose memory is NHWC.  A bf16-only tensor is handed through only to
    callers that say they can read its image (keep_phantom); everybody else receives the widened fp32 copy."""
    if is_phantom(t):
        return t if keep_phantom else f32_of(t)
    if is_nhwc(t):
        return t
    return t.contiguous(memory_format=CL)


def empty_nhwc(n, c, h, w, device, dtype=torch.float32):
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def zeros_nhwc(n, c, h, w, device, dtype=torch.float32):
    return torch.zeros((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def out_hw(h, w, r, s, stride, ph, pw):
    return (h + 2 * ph - r) // stride + 1, (w + 2 * pw - s) // stride + 1


# bf16 matrix operands for the convolutions (BASELINE config 4; csrc/conv_bf16.hip).  A process-wide switch set by the
# model (cfg.Model.bf16 -> rrnet_amd.models.rrnet) for the duration of its forward / backward: same tensors (fp32 in
# HBM), same autograd graph, only the kernels the launches go to differ.  The headline configuration never sets it.
#
# The switch has three positions (cfg.Model.conv_math, math_mode()):
#   0 / False  "f32"    v_mfma_f32_32x32x2_f32 (csrc/conv.hip) — the reference's arithmetic
#   1 / True   "bf16"   operands rounded to bf16 (config 4)
#   2          "f16x3"  operands split into two fp16 parts, three products: fp32-level accuracy at the 16-bit matrix rate
MATH_F32, MATH_BF16, MATH_F16X3 = 0, 1, 2
_MATH_NAMES = {"f32": MATH_F32, "fp32": MATH_F32, "bf16": MATH_BF16, "f16x3": MATH_F16X3}


def _env_mode():
    name = os.environ.get("RR_CONV_MATH", "f32")
    if name not in _MATH_NAMES:
        raise ValueError("RR_CONV_MATH must be one of %s, got %r" % (sorted(_MATH_NAMES), name))
    return _MATH_NAMES[name]


_BF16_ENV = _env_mode()


# The switch is per THREAD (two models of different conv_math driven from two threads, or a backward on an autograd worker
# while another thread runs a forward, must not see each other's setting); a thread that never set it reads the
# environment's default.  `ops.BF16` stays readable / assignable as a module attribute (the module class below).
class _Mode(threading.local):
    def __init__(self):
        self.value = _BF16_ENV


_MODE = _Mode()


def _mode():
    return _MODE.value


def math_mode(model_cfg):
    """cfg.Model -> the switch position: conv_math ("f32" | "bf16" | "f16x3") if present, else the older bf16 flag."""
    name = getattr(model_cfg, "conv_math", None)
    if name:
        if name not in _MATH_NAMES:
            raise ValueError("cfg.Model.conv_math must be one of %s, got %r" % (sorted(_MATH_NAMES), name))
        return _MATH_NAMES[name]
    return MATH_BF16 if getattr(model_cfg, "bf16", False) else MATH_F32


class bf16_scope:
    """`with ops.bf16_scope(mode):` — the convolutions launched inside (by this thread) take the fp32 (0 / None: the
    environment's default, RR_CONV_MATH), bf16-operand (1 / True) or split-operand (2) kernels; `force=True` takes `mode`
    literally, so that 0 means the fp32 kernels whatever the environment says.  RRNet.forward opens it for a model built
    with cfg.Model.bf16 / conv_math; every convolution node remembers the setting of its forward and re-opens it around
    its backward (rrnet_amd/functional.py)."""

    def __init__(self, on, force=False):
        self.on = int(on or 0) if force else (int(on or 0) or _BF16_ENV)

    def __enter__(self):
        self.prev, _MODE.value = _MODE.value, self.on

    def __exit__(self, *exc):
        _MODE.value = self.prev


def _bf16_ok(c, k, r, s, *tensors, pixels=None):
    """Shapes csrc/conv_bf16.hip takes: vector path (C and K multiples of 4, <= 64 taps), tensors below 2 GiB.
    -> 0 (fp32 kernels) or the switch position (MATH_BF16 / MATH_F16X3)."""
    ok = _mode() and c % 4 == 0 and k % 4 == 0 and r * s <= 64 and all(t.numel() * 4 < (1 << 31) for t in tensors)
    if ok and _mode() == MATH_F16X3:
        # the split kernels pay two small reductions (the operands' maxima) per launch and three matrix instructions per tile:
        # they beat the fp32-MFMA kernels on the large 3x3 layers only; the rest stays on csrc/conv.hip — same accuracy class
        if pixels is None or pixels < _SPLIT_MIN_PIXELS or c * r * s < _SPLIT_MIN_K or k < _SPLIT_MIN_CH or c < _SPLIT_MIN_CH:
            return 0
    return int(_mode()) if ok else 0


_SPLIT_MIN_PIXELS = 2048    # N*P*Q below which a layer stays on the fp32 kernels
_FUSED_AMAX_FWD = True    # bn_apply leaves max |out| for the next convolution
_FUSED_AMAX_BWD = True    # bn_bwd_apply leaves max |dx| for the data / weight gradient
_SPLIT_PRESPLIT_PIXELS = 65536   # from here on the filter is split once per launch


def split_filter(w, amax_word, pixels, flat=None):
    """The two fp16 images of a filter for the f16x3 kernels (rr_weight_split_f16: hi, then lo; scaled by the power of two
    the kernels derive from `amax_word`), or None below the layer size where a ready-made split pays.  w: the OHWI filter
    (forward) — or, with `flat`, the flat flipped / transposed copy of it (data gradients), whose K and C trade places."""
    k, c = (w.shape[0], w.shape[1])
    if pixels < _SPLIT_PRESPLIT_PIXELS or c % 8 != 0 or k % 8 != 0 or min(k, c) <= 64:
        return None
    src = w if flat is None else flat
    out = torch.empty(2 * src.numel(), dtype=torch.float16, device=src.device)
    _C.check(_C.fn("rr_weight_split_f16")(_C.ptr(src), src.numel(), _C.ptr(amax_word), _C.ptr(out), _C.stream()), "rr_weight_split_f16")
    return out


_SPLIT_PER_LAUNCH = True     # the filter split at every large launch into a fresh temporary (0: in-tile split)
_SPLIT_MIN_CH = 64                                                         # narrower layers (either side) likewise
_SPLIT_MIN_K = 1024               # C*R*S (reduction length) likewise


def amax_carry(t):
    """The remembered maximum of t (or None), to be kept by an autograd node next to the tensor it saves: saved tensors
    come back as new Python objects.  amax_restore() hands it to the unpacked tensor in backward."""
    hit = getattr(t, "_rr_amax", None)
    return hit[2] if (hit is not None and hit[0] == t._version) else None


def amax_restore(t, word):
    """Backward: the maximum computed in forward (complete long before: valid on every stream)."""
    if word is not None and t is not None:
        t._rr_amax = (t._version, None, word)


def amax_publish(t):
    """The caller has just ordered another stream behind the current one (wait_stream): a maximum remembered on t from the
    current stream may be read there."""
    hit = getattr(t, "_rr_amax", None) if t is not None else None
    if hit is not None and hit[1] == torch.cuda.current_stream(t.device).cuda_stream:
        t._rr_amax = (hit[0], None, hit[2])


_AMAX_CHECK = os.environ.get("RR_AMAX_CHECK", "0") == "1"     # recompute every remembered maximum when it is used; raise on mismatch


def amax_drop(t):
    """`t` is about to be (or has just been) written through its raw pointer by a kernel — an accumulate-form data gradient,
    bn_bwd_apply's g_into, any `out=` argument: such writes never move `t._version`, so a maximum remembered on the tensor
    object would look valid and be stale (too low: the fp16 parts overflow; too high: bits are dropped silently)."""
    if t is not None and getattr(t, "_rr_amax", None) is not None:
        t._rr_amax = None
    if t is not None and getattr(t, "_rr_b16", None) is not None:
        t._rr_b16 = None                       # (the bf16 image is as stale as the maximum)


# ---- bf16 images of fp32 tensors (csrc/conv16.hip reads both operands of a convolution as bf16 tensors) -----------------
# Under cfg.Model.bf16 the producers of a convolution's operands (bn_apply, bn_bwd_apply) write a bf16 image next to the fp32
# tensor in the same pass; it rides on the tensor OBJECT as `_rr_b16 = (version, stream id | None, image)`.  An operand
# without one (a fan-in sum, the up-sample add, a head's masked gradient) is converted on first use (rr_to_bf16) and
# remembered the same way.  Like the remembered maxima it must be dropped when a kernel rewrites the tensor through its
# raw pointer (amax_drop does both), is valid on the stream that made it, and on every stream once published.
_CONV16 = os.environ.get("RR_CONV16", "1") != "0"               # 0: the round-4 kernels (fp32 tensors, converted inside every launch)
_CONV16_MIN_PIXELS = 8192     # output pixels below which the 256-pixel tiles lose to the round-4 kernels (8 x 32 x 32 x 384: 351 vs 229 TFLOP/s; 16 x 16: 24 workgroups)


class _PhantomScope(threading.local):
    def __init__(self):
        self.on = False


_PHANTOM = _PhantomScope()
_PHANTOM_TRACE = False       # count, by shape and caller, the bf16-only tensors that had to be widened
PHANTOM_WIDENED = {}
_PHANTOM_ENABLED = os.environ.get("RR_BF16_ONLY_ACT", "1") != "0"      # 0: every activation keeps its fp32 tensor next to the image


class phantom_scope:
    """Inside (HourglassNet.forward under cfg.Model.bf16): conv -> bn [-> +res] -> relu layers and the up-path add may return
    bf16-ONLY activations (phantom_f32) where the shape qualifies — everything that consumes them there reads the image.  Tensors
    that leave the backbone are produced outside the scope and are ordinary fp32 tensors."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _PHANTOM.on = _PHANTOM.on, self.on

    def __exit__(self, *exc):
        _PHANTOM.on = self.prev


def phantom_out_ok(c, pixels, device):
    # (no grad-mode test: inside an autograd Function's forward grad mode is always off)
    return bool(_PHANTOM.on and _PHANTOM_ENABLED and _CONV16 and _mode() == MATH_BF16 and device.type == "cuda"
                and c % 256 == 0 and pixels >= _CONV16_MIN_PIXELS)


_PHANTOM_Y = True      # 0: pre-BN convolution outputs keep their fp32 tensor


def phantom_y_ok(k, x, w, stride, pad):
    """conv -> bn layers inside the backbone (phantom_scope): may the convolution's pre-BN output exist as a bf16 image only?"""
    if not (_PHANTOM_Y and x.is_cuda):
        return False
    n, c, h, wd = x.shape
    r, s = w.shape[2], w.shape[3]
    p, q = out_hw(h, wd, r, s, stride, pad[0], pad[1])
    return phantom_out_ok(k, n * p * q, x.device) and conv16_ok(c, k, r, s, stride, n * p * q, x)


def b16_attach(t, image):
    t._rr_b16 = (t._version, torch.cuda.current_stream(t.device).cuda_stream, image)


def b16_carry(t):
    """The bf16 image riding on t (or None) — for an autograd node to keep next to a tensor it saves (saved tensors come
    back as new Python objects); b16_restore hands it to the unpacked tensor."""
    hit = getattr(t, "_rr_b16", None)
    return hit[2] if (hit is not None and hit[0] == t._version) else None


def b16_restore(t, image):
    if image is not None and t is not None:
        t._rr_b16 = (t._version, None, image)          # made in the forward: complete on every stream by now


def b16_publish(t):
    hit = getattr(t, "_rr_b16", None) if t is not None else None
    if hit is not None and hit[1] == torch.cuda.current_stream(t.device).cuda_stream:
        t._rr_b16 = (hit[0], None, hit[2])


def bf16_of(t):
    """bf16 image of the fp32 NHWC tensor t (same logical shape, same memory order)."""
    if is_phantom(t):
        return image_of(t)
    sid = torch.cuda.current_stream(t.device).cuda_stream
    hit = getattr(t, "_rr_b16", None)
    if hit is not None and hit[0] == t._version and (hit[1] is None or hit[1] == sid):
        return hit[2]
    assert t.dtype == torch.float32 and t.numel() % 4 == 0
    img = torch.empty_like(t, dtype=torch.bfloat16)
    assert img.stride() == t.stride()
    _C.check(_C.fn("rr_to_bf16")(_C.ptr(t), _C.ptr(img), t.numel(), _C.stream()), "rr_to_bf16")
    try:
        t._rr_b16 = (t._version, sid, img)
    except AttributeError:
        pass
    return img


_PHANTOM_STUB = {}       # device -> the two-NaN storage every handle on that device is a view of


def phantom_f32(shape, device, image):
    """An fp32 tensor OBJECT of the given logical shape that owns no memory (one element, stride 0) and carries `image` as its
    bf16 image: the handle of a gradient that exists only in bf16 — every consumer is a conv16 kernel.  A kernel wrapper that
    would read its fp32 data trips over `is_nhwc` (stride 0); a torch-native consumer (a hook, `+`, `.sum()`, a print) reads
    NaN — the stub's one element — so an accidental read poisons the loss loudly instead of training on one garbage value."""
    device = torch.device(device)
    stub = _PHANTOM_STUB.get(device)
    if stub is None:
        stub = _PHANTOM_STUB[device] = torch.full((2,), float("nan"), dtype=torch.float32, device=device)
    t = stub[1:].expand(shape)                            # (element 1 of 2: is_phantom's signature)
    t._rr_b16 = (t._version, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None, image)
    return t


def wgrad_16bit_shape(c, k, r, s):
    """Layer shapes whose weight gradient leaves the fp32 kernel when a 16-bit arithmetic is on (conv_wgrad's dispatch; the
    kernel audit mirrors it)."""
    return c > 32 and (k > 32 or (k >= 16 and r * s >= 9))


def dgrad16_takes(dy_shape, w_shape, x_shape, stride, pad, relu_bias_link=False):
    """True when conv_dgrad(dy, w, x_shape, ...) will go to rr_conv16_dgrad_s1 (mirrors the dispatch there)."""
    n, c, h, wd = x_shape
    k, _, r, s = w_shape
    if stride == 2:
        return bool(_s2_parity_pads_ok(r, s, pad) and _CONV16 and _mode() == MATH_BF16 and _BF16_S2_DGRAD and k % 64 == 0 and c % 128 == 0
                    and r * s <= 16 and n * h * wd // 4 >= _CONV16_MIN_PIXELS
                    and max(n * h * wd * c, dy_shape[0] * dy_shape[1] * dy_shape[2] * dy_shape[3]) * 2 < (1 << 31))
    return bool(stride == 1 and pad[0] < r and pad[1] < s and not relu_bias_link and _CONV16 and _mode() == MATH_BF16
                and n * h * wd >= _CONV16_MIN_PIXELS and _C.fn("rr_conv16_supported")(k, c, r, s, 1)
                and max(n * h * wd * c, dy_shape[0] * dy_shape[1] * dy_shape[2] * dy_shape[3]) * 2 < (1 << 31))


def wgrad16_takes(x_shape, dy_shape, w_shape, stride):
    """True when conv_wgrad(x, dy, dw, stride, ...) will go to rr_conv16_wgrad."""
    k, c, r, s = w_shape
    npix = dy_shape[0] * dy_shape[2] * dy_shape[3]
    xn = x_shape[0] * x_shape[1] * x_shape[2] * x_shape[3]
    return bool(_CONV16 and _mode() == MATH_BF16 and npix >= _CONV16_MIN_PIXELS and _C.fn("rr_conv16_wgrad_supported")(c, k, r, s, stride)
                and max(xn, npix * k) * 2 < (1 << 31))


def conv16_ok(c, k, r, s, stride, pixels, *tensors):
    """-> True when a forward-kernel launch of this shape goes to csrc/conv16.hip (bf16 mode only)."""
    return bool(_CONV16 and _mode() == MATH_BF16 and pixels >= _CONV16_MIN_PIXELS and all(t is None or t.is_cuda for t in tensors)
                and _C.fn("rr_conv16_supported")(c, k, r, s, stride) and all(t is None or t.numel() * 2 < (1 << 31) for t in tensors))


def _absmax_word(t):
    word = _ZEROS.take(1, t.device)                   # 8 zeroed bytes; the kernels read the first 4
    n = t.numel()
    assert n % 4 == 0 and t.dtype == torch.float32
    _C.check(_C.fn("rr_absmax_bits")(_C.ptr(t), n, _C.ptr(word), _C.stream()), "rr_absmax_bits")
    return word


def amax_of(t):
    """Device word holding the bit pattern of max|t| (rr_absmax_bits) for the split-operand kernels, computed on the
    current stream and remembered on the tensor object (same version, same stream): a gradient serves its data
    gradient and its weight gradient with one reduction.  RR_AMAX_CHECK=1 (tests): every remembered word is checked
    against a fresh reduction at the moment it is used."""
    sid = torch.cuda.current_stream(t.device).cuda_stream
    hit = getattr(t, "_rr_amax", None)
    if hit is not None and hit[0] == t._version and (hit[1] == sid or hit[1] is None):
        if _AMAX_CHECK:
            fresh = int(_absmax_word(t).view(torch.int32)[0].item())
            kept = int(hit[2].view(torch.int32)[0].item())
            if fresh != kept:
                raise RuntimeError("ops.amax_of: stale remembered maximum on a tensor of shape %s: remembered bits 0x%08x, "
                                   "actual 0x%08x (a kernel wrote it through its raw pointer without ops.amax_drop)"
                                   % (tuple(t.shape), kept & 0xffffffff, fresh & 0xffffffff))
            AMAX_CHECKED[0] += 1
        return hit[2]
    word = _absmax_word(t)
    try:
        t._rr_amax = (t._version, sid, word)
    except AttributeError:
        pass
    return word


AMAX_CHECKED = [0]       # remembered maxima verified under RR_AMAX_CHECK=1 (the audit asserts that the check really ran)


def _math_tail(mode, filter16, src, flt, filter_split):
    """The trailing arguments of a forward-kernel launch under the three arithmetics: fp32 (stream), bf16 (the filter's cached
    bf16 copy or NULL, stream), f16x3 (the two operands' maxima, the filter's ready-made fp16 split or NULL, stream)."""
    if mode == MATH_F16X3:
        return (_C.ptr(amax_of(src)), _C.ptr(amax_of(flt)), _C.ptr(filter_split), _C.stream())
    if mode == MATH_BF16:
        return (_C.ptr(filter16), _C.stream())
    return (_C.stream(),)


def conv_fprop(x, w, bias=None, stride=1, pad=(0, 0), relu=False, want_stats=False, algo_kg=None, w16=None, w_split=None,
               y_bf16_only=False):
    """x [N,C,H,W] (NHWC memory), w [K,C,R,S] (OHWI memory) -> y [N,K,P,Q] (NHWC memory)
    and, if want_stats, the per-block BatchNorm partial-sum slab (see rr_conv_fprop).
    algo_kg: the useful reduction length when the operands carry zero padding (timer FLOPs stay algorithmic)."""
    _C.require_cuda(x, w, bias)
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    n, c, h, wd = x.shape
    k, c2, r, s = w.shape
    assert c == c2
    p, q = out_hw(h, wd, r, s, stride, pad[0], pad[1])
    use16 = conv16_ok(c, k, r, s, stride, n * p * q, x) and n * p * q * k * 4 < (1 << 31)
    y16 = None
    if use16 and y_bf16_only:
        # the output as a bf16 image only (the caller's BatchNorm reads it; the statistics below come from the fp32 accumulators)
        y16 = torch.empty((n, p, q, k), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
        y = phantom_f32((n, k, p, q), x.device, y16)
    else:
        y = empty_nhwc(n, k, p, q, x.device)
    slab = None
    if is_phantom(x) and not use16:
        x = f32_of(x)                       # a bf16-only input in front of a layer the conv16 kernels do not take
    assert (is_nhwc(x) or is_phantom(x)) and is_nhwc(w), "conv_fprop wants NHWC activations / OHWI weights"
    if use16:
        # both operands as bf16 tensors (csrc/conv16.hip): x's image from its producer (or converted once), the filter's from
        # the flat cache
        x16 = bf16_of(x)
        if w16 is None:
            w16 = bf16_of(w)
        if want_stats:
            slab = torch.empty(_C.fn("rr_conv16_stat_slab_bytes")(n, p, q, k) // 8, dtype=torch.float64, device=x.device)
        flops = 2.0 * n * p * q * k * (c * r * s if algo_kg is None else algo_kg)
        _C.check(_timed("conv16_fprop", flops,
                        lambda: _C.fn("rr_conv16_fprop")(_C.ptr(x16), _C.ptr(w16), _C.ptr(bias), _C.ptr(None if y16 is not None else y),
                                                         _C.ptr(y16), _C.ptr(slab), n, h, wd, c, k, r, s, stride, pad[0], pad[1],
                                                         int(relu), _C.stream()),
                        (n, h, wd, c, k, r, s, stride), 2.0 * (x.numel() + w.numel()) + (2.0 if y16 is not None else 4.0) * y.numel()),
                 "rr_conv16_fprop")
        return (y, slab) if want_stats else y
    if want_stats:
        nbytes = _C.fn("rr_conv_stat_slab_bytes")(n, p, q, k)
        slab = torch.empty(nbytes // 8, dtype=torch.float64, device=x.device)
    bf = _bf16_ok(c, 4 if _mode() != MATH_F16X3 else k, r, s, x, w, y, pixels=n * p * q)
    f = _C.fn(("rr_conv_fprop", "rr_conv_fprop_bf16", "rr_conv_fprop_f16x3")[bf])
    flops = 2.0 * n * p * q * k * (c * r * s if algo_kg is None else algo_kg)
    # w16: the filter already rounded to bf16 (optional); split operands: the two tensors' maxima
    if bf == MATH_F16X3 and w_split is None and _SPLIT_PER_LAUNCH:
        # a LOCAL that lives until the launch below has been enqueued: built inside the argument tuple the temporary was
        # released before the launch, the caching allocator handed its block to the next zero-filled scratch, and the
        # data gradients read a filter of zeros / garbage — non-finite gradients, and a step that ran 10 % FASTER
        w_split = split_filter(w, amax_of(w), n * p * q)
    tail = _math_tail(bf, w16, x, w, w_split)
    _C.check(_timed(_igemm_name("fprop", k, c % 4 != 0, n * p * q) + ("", "+bf16", "+f16x3")[bf], flops,
                    lambda: f(_C.ptr(x), _C.ptr(w), _C.ptr(bias), _C.ptr(y), _C.ptr(slab), n, h, wd, c, k, r, s,
                              stride, pad[0], pad[1], int(relu), *tail),
                    (n, h, wd, c, k, r, s, stride), 4.0 * (x.numel() + y.numel() + w.numel())), "rr_conv_fprop")
    return (y, slab) if want_stats else y


_STEM_PACK = True


def conv_packable(x, w, stride):
    """A convolution on very few channels (the 7x7 stride-2 stem on an RGB image): its taps are packed into one row
    per output pixel and it runs as a 1x1 convolution on the vector kernels (conv_fprop_packed / conv_wgrad_packed)."""
    k, c, r, s = w.shape
    return _STEM_PACK and c % 4 != 0 and 32 < r * s * c <= 512 and k >= 32 and not x.requires_grad


def conv_fprop_packed(x, w, stride, pad, want_stats=False):
    """-> (y, slab | None, xp): xp [N,KP,P,Q] = the packed taps (rr_conv_pack_taps), kept for the weight gradient."""
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    kg = r * s * c
    kp = (kg + 31) // 32 * 32
    p, q = out_hw(h, wd, r, s, stride, pad[0], pad[1])
    xp = empty_nhwc(n, kp, p, q, x.device)
    _C.check(_C.fn("rr_conv_pack_taps")(_C.ptr(x), _C.ptr(xp), n, h, wd, c, r, s, stride, pad[0], pad[1], kp, _C.stream()),
             "rr_conv_pack_taps")
    wp = torch.zeros((k, kp), dtype=torch.float32, device=x.device)
    wp[:, :kg] = w.permute(0, 2, 3, 1).reshape(k, kg)                       # OHWI memory = [k][(r,s,c)]
    out = conv_fprop(xp, wp.view(k, kp, 1, 1), None, 1, (0, 0), False,
                     want_stats, algo_kg=kg)
    return (out[0], out[1], xp) if want_stats else (out, None, xp)


def conv_wgrad_packed(xp, dy, dw):
    """dw [K,C,R,S] (OHWI memory) += the weight gradient of the convolution whose packed taps are xp."""
    k, c, r, s = dw.shape
    kg, kp = r * s * c, xp.shape[1]
    dwp = zeros_nhwc(k, kp, 1, 1, xp.device)
    conv_wgrad(xp, dy, dwp, 1, (0, 0), algo_c=kg)
    dw.permute(0, 2, 3, 1).add_(dwp.reshape(k, kp)[:, :kg].view(k, r, s, c))
    return dw


def stem_wgrad_s2d(x, dy, dw):
    """wgrad of the 7x7 stride-2 pad-3 stem on a 3-channel image, computed on the space-to-depth image:
    y[p] = sum_r x[2p-3+r] w[r]  ==  sum_{a<4,i<2} x2[p-2+a, i] w'[2a+i],  x2[u,i] = x[2u+i], w'[r+1] = w[r]
    i.e. a 4x4 stride-1 convolution over 12 channels with leading pad 2 (trailing 1).  Per tap the MFMA
    tile then carries 12 useful channels instead of 3 (49 taps x 32 padded -> 16 taps x 32 padded)."""
    n, c, h, w = x.shape
    k = dw.shape[0]
    assert c == 3 and h % 2 == 0 and w % 2 == 0 and tuple(dw.shape[1:]) == (3, 7, 7)
    xm = x.permute(0, 2, 3, 1)                                   # NHWC view of the memory
    x2 = xm.reshape(n, h // 2, 2, w // 2, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 12)
    x2 = x2.contiguous().permute(0, 3, 1, 2)                     # logical [n,12,h/2,w/2], NHWC memory
    dw2 = zeros_nhwc(k, 12, 4, 4, x.device)                      # OHWI [k][a][b][(i,j,c)]
    conv_wgrad(x2, dy, dw2, 1, (2, 2), explicit_out=True)
    g = dw2.permute(0, 2, 3, 1).reshape(k, 4, 4, 2, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(k, 8, 8, 3)
    dw.permute(0, 2, 3, 1).add_(g[:, 1:, 1:, :])                 # drop the phantom taps r' = 0 / s' = 0
    return dw


_DGRAD_VIA_FPROP = True
_DGRAD_VIA_FPROP_MIN_PIXELS = 4096  # below: the dgrad kernel's split-K wins


class BnLink:
    """What the data gradient of a convolution needs to ALSO produce the BatchNorm-backward sums of the conv -> bn
    [-> +residual] -> relu layer that produced its input (rr_conv_dgrad_s1_bnsum): the producer's pre-BN output y, its
    statistics, and the source of its ReLU mask (z for layers with a residual, scale / shift otherwise).  The consumer's
    backward leaves `sums` (and the gradient tensor they were taken of) here; the producer's backward picks them up
    instead of running rr_bn_bwd_reduce when the gradient it receives is that very tensor."""
    __slots__ = ("y", "use_z", "mean", "invstd", "msc", "msh", "sums", "dz", "consumers", "relu_bias", "mask_only")

    def __init__(self):
        self.y = self.mean = self.invstd = self.msc = self.msh = self.sums = self.dz = None
        # relu_bias: the producer is conv + bias + ReLU (a head's 3x3 layer): the consumer's data gradient stores the
        # ReLU-masked gradient and sums[:c] is the producer's bias gradient (rr_conv_dgrad_s1_relubias)
        self.relu_bias = False
        self.mask_only = False      # relu_bias, and the producer is a bare ReLU (functional._ReLU): only the mask is wanted, no column sums
        # the mask comes from the producer's OUTPUT z (layers with a residual).  The link must not hold z itself — z
        # carries the link as an attribute, and such a cycle keeps a step's activations alive until the cyclic GC runs
        # (the caching allocator then thrashes); the consumer passes its own saved input, which IS z.
        self.use_z = False
        self.consumers = 0


_DGRAD_BNSUM = True
# measured at 8 x 256 x 256 x 256 (tools/bench_head_dgrad.py): K = 10: 0.23 ms against 0.40, K = 2: 0.22 against 0.39; K = 34 (the WH head: 144 filter
# registers per lane, two waves per SIMD): 0.58 against 0.48 — that layer stays on the implicit-GEMM kernel
_HEAD_DGRAD_MAX_K = 12
_HEAD_DGRAD = True      # 0: the heads' narrow 1x1 data gradients on the implicit-GEMM kernel (round 3)
_BF16_S2_DGRAD = True     # A/B: stride-2 data gradients stay on the fp32 kernel


def _s2_parity_pads_ok(r, s, pad):
    """The parity-class launches of rr_conv_dgrad_s2_bf16 / _f16x3 need a non-negative leading pad in every class that has
    taps (csrc/conv_bf16.hip, dgrad_s2_impl: lead = taps - 1 - (parity + pad - first tap) // 2); e.g. 3x3 stride 2 pad 2 has
    none — such a layer takes rr_conv_dgrad like every other unsupported shape instead of raising."""
    for size, pd in ((r, pad[0]), (s, pad[1])):
        for par in (0, 1):
            t0 = (par + pd) & 1
            taps = (size - t0 + 1) // 2 if t0 < size else 0
            if taps and (taps - 1) - (par + pd - t0) // 2 < 0:
                return False
    return True


def conv_dgrad(dy, w, x_shape, stride=1, pad=(0, 0), out=None, accumulate=False, bnsum=None, bnsum_z=None, wt=None, wt16=None,
               wt_split=None):
    """dy [N,K,P,Q], w [K,C,R,S] -> dx [N,C,H,W]; with `out` and accumulate adds into it.
    wt: the flipped / transposed filter (rr_weight_flip_transpose of w) when the caller keeps one (FlatParams.wt_view).
    bnsum (BnLink of the layer that produced the convolution's input): when the launch can carry them, the producer's
    BatchNorm-backward sums are computed in the epilogue and left in bnsum.sums / bnsum.dz.  bnsum_z: the
    convolution's input itself (= the producer's output), needed when bnsum.use_z."""
    _C.require_cuda(dy, w)
    assert (is_nhwc(dy) or is_phantom(dy)) and is_nhwc(w)     # (phantom: a bf16-only gradient, see phantom_f32)
    n, c, h, wd = x_shape
    k, c2, r, s = w.shape
    assert c == c2 and dy.shape[1] == k
    if out is None:
        out = empty_nhwc(n, c, h, wd, dy.device)
        accumulate = False
    else:
        amax_drop(out)                # an existing tensor rewritten / added into through its pointer
    assert is_nhwc(out)
    if bnsum is not None and not bnsum.relu_bias and bnsum.y is not None and is_phantom(bnsum.y):
        bnsum = None                  # the producer's pre-BN output exists only as a bf16 image: it runs its own reduce pass
    if bnsum is not None and bnsum_z is not None and is_phantom(bnsum_z):
        # the producer's output exists only as a bf16 image: the fp32-reading epilogues cannot take their mask from it — a ReLU /
        # bias producer gets the widened copy, a BatchNorm producer runs its own reduce pass (rr_bn_bwd_reduce_b16)
        if bnsum.relu_bias:
            bnsum_z = f32_of(bnsum_z)
        else:
            bnsum = None
    s2_16 = (stride == 2 and _s2_parity_pads_ok(r, s, pad) and _CONV16 and _mode() == MATH_BF16 and _BF16_S2_DGRAD and dy.is_cuda
             and k % 64 == 0 and c % 128 == 0 and r * s <= 16 and n * h * wd // 4 >= _CONV16_MIN_PIXELS)
    if is_phantom(dy) and not s2_16 and not (stride == 1 and pad[0] < r and pad[1] < s and conv16_ok(k, c, r, s, 1, n * h * wd, dy, out)
                                             and not (bnsum is not None and bnsum.relu_bias)):
        dy = f32_of(dy)
    if (bnsum is not None and bnsum.relu_bias and _DGRAD_BNSUM and stride == 1 and c % 4 == 0
            and bnsum_z is not None and is_nhwc(bnsum_z) and bnsum_z.shape == out.shape and out.numel() * 4 < (1 << 31)
            and r * s <= 64 and pad[0] < r and pad[1] < s and dy.shape[2] * dy.shape[3] >= _DGRAD_VIA_FPROP_MIN_PIXELS
            and (n * h * wd) % 128 == 0):      # whole 128-row tiles only (see fprop_impl in csrc/conv.hip)
        if bnsum.mask_only and pad[0] < r and pad[1] < s and conv16_ok(k, c, r, s, 1, n * h * wd, dy, out) and not is_phantom(bnsum_z):
            # the producer is a bare ReLU (its mask is all that is wanted): conv16's data gradient with the mask in its epilogue
            dy16 = bf16_of(dy)
            if wt16 is None:
                if wt is None:
                    wt = torch.empty(k * c * r * s, dtype=torch.float32, device=dy.device)
                    _C.check(_C.fn("rr_weight_flip_transpose")(_C.ptr(w), _C.ptr(wt), k, c, r, s, _C.stream()), "rr_weight_flip_transpose")
                wt16 = torch.empty(wt.numel(), dtype=torch.bfloat16, device=dy.device)
                _C.check(_C.fn("rr_to_bf16")(_C.ptr(wt), _C.ptr(wt16), wt.numel(), _C.stream()), "rr_to_bf16")
            flops_m = 2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * k * c * r * s
            _C.check(_timed("conv16_dgrad_s1+relumask", flops_m,
                            lambda: _C.fn("rr_conv16_dgrad_s1_relumask")(_C.ptr(dy16), _C.ptr(wt16), _C.ptr(out), n, h, wd, c, k, r, s, pad[0],
                                                                         pad[1], int(accumulate), _C.ptr(bnsum_z), _C.stream()),
                            (n, h, wd, c, k, r, s, stride)), "rr_conv16_dgrad_s1_relumask")
            bnsum.sums, bnsum.dz = _ZEROS.take(2, dy.device), out          # (sums: the "done" marker _ReLU.backward looks for)
            return out
        if r == 1 and s == 1 and k <= _HEAD_DGRAD_MAX_K and 1024 % c == 0 and _HEAD_DGRAD and dy.is_cuda:
            # a head's narrow 1x1 layer (K = 10 / 2 / 34 of 36): not a GEMM worth a matrix kernel — one HBM-bound pass
            # (rr_head_dgrad_relubias), no channel padding; under conv16 the bf16 image of dx comes out of the same pass
            sums = _ZEROS.take(2 * c, dy.device)
            want16 = _CONV16 and _mode() == MATH_BF16 and c % 256 == 0 and n * h * wd >= _CONV16_MIN_PIXELS
            out16 = torch.empty_like(out, dtype=torch.bfloat16) if want16 else None
            _C.check(_C.fn("rr_head_dgrad_relubias")(_C.ptr(dy), _C.ptr(w), _C.ptr(out), _C.ptr(out16), _C.ptr(bnsum_z), _C.ptr(sums),
                                                     n * h * wd, c, k, int(accumulate), _C.stream()), "rr_head_dgrad_relubias")
            if want16:
                b16_attach(out, out16)
            bnsum.sums, bnsum.dz = sums, out
            return out
        # producer = conv + bias + ReLU: masked gradient + bias column sums in this launch's epilogue.  A 10- or
        # 2-channel dy (hm / offset heads) is zero-padded to a multiple of 4 for the vector kernel (25 MB at 8x256x256)
        kp = (k + 3) // 4 * 4
        dyp, wp = dy, w
        if kp != k:
            dyp = empty_nhwc(dy.shape[0], kp, dy.shape[2], dy.shape[3], dy.device)
            _C.check(_C.fn("rr_pad_channels")(_C.ptr(dy), _C.ptr(dyp), dy.shape[0] * dy.shape[2] * dy.shape[3], k, kp, _C.stream()),
                     "rr_pad_channels")
            wp = zeros_nhwc(kp, c, r, s, dy.device)
            wp[:k] = w
        if wt is None or kp != k:
            wt = torch.empty(kp * c * r * s, dtype=torch.float32, device=dy.device)
            _C.check(_C.fn("rr_weight_flip_transpose")(_C.ptr(wp), _C.ptr(wt), kp, c, r, s, _C.stream()), "rr_weight_flip_transpose")
        slab = torch.empty(_C.fn("rr_conv_stat_slab_bytes")(n, h, wd, c) // 8, dtype=torch.float64, device=dy.device)
        sums = _ZEROS.take(2 * c, dy.device)
        flops_m = 2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * k * c * r * s
        bf = _bf16_ok(kp, c, r, s, dyp, out, pixels=n * h * wd)
        fr = _C.fn("rr_conv_dgrad_s1_relubias" + ("", "_bf16", "_f16x3")[bf])
        rtail = (_C.ptr(amax_of(dyp)), _C.ptr(amax_of(wp)), _C.stream()) if bf == MATH_F16X3 else (_C.stream(),)
        _C.check(_timed(_igemm_name("fprop", c, False, n * h * wd) + "+relubias" + ("", "+bf16", "+f16x3")[bf], flops_m,
                        lambda: fr(_C.ptr(dyp), _C.ptr(wt), _C.ptr(out), n, h, wd, c, kp, r, s, pad[0], pad[1], int(accumulate),
                                   _C.ptr(bnsum_z), _C.ptr(slab), _C.ptr(sums), *rtail), (n, h, wd, c, k, r, s, stride)),
                 "rr_conv_dgrad_s1_relubias")
        bnsum.sums, bnsum.dz = sums, out
        return out
    flops = 2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * k * c * r * s
    if (stride == 1 and pad[0] < r and pad[1] < s and conv16_ok(k, c, r, s, 1, n * h * wd, dy, out)
            and not (bnsum is not None and bnsum.relu_bias)):
        # csrc/conv16.hip: the forward kernel on dY's bf16 image and the flipped filter's bf16 copy.  (The producer's
        # BatchNorm-backward sums are NOT carried by this kernel: `bnsum` stays untouched and the producer runs its reduce pass.
        # An epilogue that did — round 5 — lost: with one workgroup per CU its image reads are in the open, config-4 step
        # 111.8-113.3 ms against 108.9-110.8 ms; removed in round 6, the numbers live in DESIGN §13.1.)
        dy16 = bf16_of(dy)
        if wt16 is None:
            if wt is None:
                wt = torch.empty(k * c * r * s, dtype=torch.float32, device=dy.device)
                _C.check(_C.fn("rr_weight_flip_transpose")(_C.ptr(w), _C.ptr(wt), k, c, r, s, _C.stream()), "rr_weight_flip_transpose")
            wt16 = torch.empty(wt.numel(), dtype=torch.bfloat16, device=dy.device)
            _C.check(_C.fn("rr_to_bf16")(_C.ptr(wt), _C.ptr(wt16), wt.numel(), _C.stream()), "rr_to_bf16")
        _C.check(_timed("conv16_dgrad_s1", flops,
                        lambda: _C.fn("rr_conv16_dgrad_s1")(_C.ptr(dy16), _C.ptr(wt16), _C.ptr(out), None, n, h, wd, c, k, r, s, pad[0], pad[1],
                                                            int(accumulate), _C.stream()), (n, h, wd, c, k, r, s, stride),
                        2.0 * (dy.numel() + w.numel()) + 4.0 * out.numel() * (2 if accumulate else 1)), "rr_conv16_dgrad_s1")
        return out
    # (bf16 operands: the forward kernel at every size — its split-K covers the small maps, and the dgrad kernel is fp32-only)
    if (stride == 1 and k % 4 == 0 and c % 4 == 0 and r * s <= 64 and pad[0] < r and pad[1] < s and _DGRAD_VIA_FPROP
            and (dy.shape[2] * dy.shape[3] >= _DGRAD_VIA_FPROP_MIN_PIXELS or _bf16_ok(k, c, r, s, dy, out, pixels=n * h * wd))):
        # the forward kernel on dy with the flipped / transposed filter (one tiny transpose per layer and step)
        if wt is None:
            wt = torch.empty(k * c * r * s, dtype=torch.float32, device=dy.device)
            _C.check(_C.fn("rr_weight_flip_transpose")(_C.ptr(w), _C.ptr(wt), k, c, r, s, _C.stream()),
                     "rr_weight_flip_transpose")
        bf = _bf16_ok(k, c, r, s, dy, out, pixels=n * h * wd)
        sfx, tsfx = (("", ""), ("_bf16", "+bf16"), ("_f16x3", "+f16x3"))[bf]
        # wt16: the flipped filter already in bf16 (optional); split operands: the maxima of dy and of the filter
        if bf == MATH_F16X3 and wt_split is None and _SPLIT_PER_LAUNCH:
            wt_split = split_filter(w, amax_of(w), n * h * wd, flat=wt)
        tail = _math_tail(bf, wt16, dy, w, wt_split)
        if (bnsum is not None and not bnsum.relu_bias and _DGRAD_BNSUM and bnsum.y is not None and c <= 1024 and out.numel() * 4 < (1 << 31)
                and tuple(bnsum.y.shape) == tuple(out.shape)
                and (not bnsum.use_z or (bnsum_z is not None and is_nhwc(bnsum_z) and bnsum_z.shape == out.shape))):
            zt = bnsum_z if bnsum.use_z else None
            nb = _C.fn("rr_conv_stat_slab_bytes")(n, h, wd, c)
            slab = torch.empty(nb // 8, dtype=torch.float64, device=dy.device)
            sums = _ZEROS.take(2 * c, dy.device)
            fb = _C.fn("rr_conv_dgrad_s1_bnsum" + sfx)
            _C.check(_timed(_igemm_name("fprop", c, False, n * h * wd) + "+bnsum" + tsfx, flops,
                            lambda: fb(_C.ptr(dy), _C.ptr(wt), _C.ptr(out), n, h, wd, c, k, r, s, pad[0], pad[1],
                                       int(accumulate), _C.ptr(bnsum.y), _C.ptr(zt), _C.ptr(bnsum.mean),
                                       _C.ptr(bnsum.invstd), _C.ptr(bnsum.msc), _C.ptr(bnsum.msh), _C.ptr(slab),
                                       _C.ptr(sums), *tail), (n, h, wd, c, k, r, s, stride),
                            4.0 * (dy.numel() + out.numel() * (3 if accumulate else 2) + w.numel())),
                     "rr_conv_dgrad_s1_bnsum")
            bnsum.sums, bnsum.dz = sums, out
            return out
        f1 = _C.fn("rr_conv_dgrad_s1" + sfx)
        # same HIP kernel instance as a forward convolution: timed under its name
        _C.check(_timed(_igemm_name("fprop", c, False, n * h * wd) + tsfx, flops,
                        lambda: f1(_C.ptr(dy), _C.ptr(wt), _C.ptr(out), n, h, wd, c, k, r, s, pad[0], pad[1],
                                   int(accumulate), *tail), (n, h, wd, c, k, r, s, stride),
                        4.0 * (dy.numel() + out.numel() * (2 if accumulate else 1) + w.numel())), "rr_conv_dgrad_s1")
        return out
    if (stride == 2 and _s2_parity_pads_ok(r, s, pad) and _CONV16 and _mode() == MATH_BF16 and _BF16_S2_DGRAD and dy.is_cuda
            and k % 64 == 0 and c % 128 == 0 and r * s <= 16 and n * h * wd // 4 >= _CONV16_MIN_PIXELS
            and max(dy.numel(), out.numel()) * 2 < (1 << 31)):
        # csrc/conv16.hip: the four parity-class launches on dY's bf16 image and bf16 sub-filters packed inside the call
        dy16 = bf16_of(dy)
        wsub16 = torch.empty(k * c * r * s, dtype=torch.bfloat16, device=dy.device)
        _C.check(_timed("conv16_dgrad_s2", flops,
                        lambda: _C.fn("rr_conv16_dgrad_s2")(_C.ptr(dy16), _C.ptr(w), _C.ptr(out), n, h, wd, c, k, r, s, pad[0], pad[1],
                                                            int(accumulate), _C.ptr(wsub16), _C.stream()), (n, h, wd, c, k, r, s, stride)),
                 "rr_conv16_dgrad_s2")
        return out
    if (stride == 2 and _s2_parity_pads_ok(r, s, pad) and _bf16_ok(k, c, r, s, dy, out, pixels=n * h * wd // 4)
            and _BF16_S2_DGRAD):
        # bf16 operands: one launch of the forward kernel per output parity class on its packed sub-filter
        wsub = torch.empty(k * c * r * s, dtype=torch.float32, device=dy.device)
        sx = _bf16_ok(k, c, r, s, dy, out, pixels=n * h * wd // 4) == MATH_F16X3
        f2 = _C.fn("rr_conv_dgrad_s2_f16x3" if sx else "rr_conv_dgrad_s2_bf16")
        stail = (_C.ptr(amax_of(dy)), _C.ptr(amax_of(w)), _C.stream()) if sx else (_C.stream(),)
        _C.check(_timed("conv_dgrad_s2" + ("+f16x3" if sx else "+bf16"), flops,
                        lambda: f2(_C.ptr(dy), _C.ptr(w), _C.ptr(out), n, h, wd, c, k, r, s, pad[0], pad[1], int(accumulate),
                                   _C.ptr(wsub), *stail), (n, h, wd, c, k, r, s, stride)), "rr_conv_dgrad_s2_bf16")
        return out
    f = _C.fn("rr_conv_dgrad")
    _C.check(_timed(_igemm_name("dgrad", c, (k % 4 != 0) or (c % 4 != 0), n * h * wd, stride), flops,
                    lambda: f(_C.ptr(dy), _C.ptr(w), _C.ptr(out), n, h, wd, c, k, r, s, stride, pad[0], pad[1],
                              int(accumulate), _C.stream()), (n, h, wd, c, k, r, s, stride)), "rr_conv_dgrad")
    return out


def conv_wgrad(x, dy, dw, stride=1, pad=(0, 0), explicit_out=False, algo_c=None):
    """dw [K,C,R,S] (OHWI memory) += x (*) dy.  dw must be pre-zeroed / hold the running gradient.
    explicit_out: take the output size from dy (asymmetric padding, `pad` = leading pads)."""
    _C.require_cuda(x, dy, dw)
    assert (is_nhwc(x) or is_phantom(x)) and (is_nhwc(dy) or is_phantom(dy)) and is_nhwc(dw)
    n, c, h, wd = x.shape
    k, c2, r, s = dw.shape
    assert c == c2 and dy.shape[1] == k
    flops = 2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * k * (c * r * s if algo_c is None else algo_c)
    npix = dy.shape[0] * dy.shape[2] * dy.shape[3]
    if (_CONV16 and _mode() == MATH_BF16 and not explicit_out and npix >= _CONV16_MIN_PIXELS and x.is_cuda
            and _C.fn("rr_conv16_wgrad_supported")(c, k, r, s, stride) and max(x.numel(), dy.numel()) * 2 < (1 << 31)):
        x16, dy16 = bf16_of(x), bf16_of(dy)
        _C.check(_timed("conv16_wgrad", flops,
                        lambda: _C.fn("rr_conv16_wgrad")(_C.ptr(x16), _C.ptr(dy16), _C.ptr(dw), n, h, wd, c, k, r, s, stride, pad[0], pad[1],
                                                         _C.stream()), (n, h, wd, c, k, r, s, stride)), "rr_conv16_wgrad")
        return dw
    x, dy = f32_of(x), f32_of(dy)            # (bf16-only operands in front of a layer the conv16 weight gradient does not take)
    # (narrow filter banks stay on the fp32 kernel's 32-filter tiles — except 3x3 banks of 16..32 filters, the DCN offset / mask
    #  convolution's 28: measured 1.20 ms fp32 against 0.54 ms on the bf16 kernel's 128-wide tile at the config-4 layer)
    bf = _bf16_ok(c, k, r, s, x, dy, pixels=npix) if wgrad_16bit_shape(c, k, r, s) else 0
    f = _C.fn(("rr_conv_wgrad", "rr_conv_wgrad_bf16", "rr_conv_wgrad_f16x3")[bf])
    wtail = (_C.ptr(amax_of(x)), _C.ptr(amax_of(dy)), _C.stream()) if bf == MATH_F16X3 else (_C.stream(),)
    _C.check(_timed("conv_wgrad<BN=%d>%s" % (128 if c > 32 else 32, ("", "+bf16", "+f16x3")[bf]), flops,
                    lambda: f(_C.ptr(x), _C.ptr(dy), _C.ptr(dw), n, h, wd, c, k, r, s, stride, pad[0], pad[1],
                              dy.shape[2] if explicit_out else 0, dy.shape[3] if explicit_out else 0,
                              *wtail), (n, h, wd, c, k, r, s, stride)), "rr_conv_wgrad")
    return dw


# ---------------------------------------------------------------------------------------------
# BatchNorm / elementwise
# ---------------------------------------------------------------------------------------------
def _f32(n, device):
    return torch.empty(n, dtype=torch.float32, device=device)


class _ZeroPool:
    """Pre-zeroed float64 scratch handed out in slices: the 163 BatchNorm layers of a step take their statistics
    accumulators from one zeroed chunk (one fill per ~1 MB) instead of one `torch.zeros` launch each.  A slice is
    never handed out twice; exhausted chunks stay alive as long as their slices do.
    A chunk belongs to the stream that was current when it was filled: a take() on another stream (the DCN backward's
    side stream) gets its own chunk, so every slice is ordered behind its fill.  Slices are independent tensors over
    the chunk's storage (own version counters): an in-place write into one does not invalidate siblings that
    autograd saved."""

    CHUNK = 128 * 1024      # doubles

    def __init__(self):
        self.buf = {}

    def take(self, n, device):
        n8 = (n + 1) // 2 * 2                      # keep slices 16-byte aligned
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        cur = self.buf.get(key)
        if cur is None or cur[1] + n8 > cur[0].numel():
            cur = [torch.zeros(max(self.CHUNK, n8), dtype=torch.float64, device=device), 0]
            self.buf[key] = cur
        out = torch.empty(0, dtype=torch.float64, device=device).set_(cur[0].untyped_storage(), cur[1], (n,))
        cur[1] += n8
        return out


_ZEROS = _ZeroPool()


def bn_reduce_slab(slab, c, extra=0, count=None):
    """slab [mtiles,2,c] doubles -> sums [2c (+extra)] doubles (extra slots zeroed: room for the
    sample count in the SyncBN exchange).  count: the local sample count, stored into sums[2c] by the same launch."""
    sums = _ZEROS.take(2 * c + extra, slab.device)
    mtiles = slab.numel() // (2 * c)
    if count is not None:
        assert extra >= 1
        _C.check(_C.fn("rr_bn_reduce_slab_count")(_C.ptr(slab), mtiles, c, _C.ptr(sums), float(count), _C.ptr(sums[2 * c:]), _C.stream()),
                 "rr_bn_reduce_slab_count")
        return sums
    _C.check(_C.fn("rr_bn_reduce_slab")(_C.ptr(slab), mtiles, c, _C.ptr(sums), _C.stream()), "rr_bn_reduce_slab")
    return sums


def bn_finalize_sync(sums, count_slot, gamma, beta, running_mean, running_var, momentum, eps, num_batches_tracked=None):
    """bn_finalize for an exchanged statistics buffer: the (global) sample count is read from the device (`count_slot`, a slot of
    the buffer that went through the all-reduce) and handed back in storage of its own for the backward.
    -> mean, invstd, scale, shift, count (float64 [1])"""
    c = gamma.numel()
    buf = _f32(4 * c, gamma.device)
    mean, invstd, scale, shift = buf[0:c], buf[c:2 * c], buf[2 * c:3 * c], buf[3 * c:4 * c]
    cnt = torch.empty(1, dtype=torch.float64, device=gamma.device)
    assert num_batches_tracked is None or num_batches_tracked.dtype == torch.int64
    _C.check(_C.fn("rr_bn_finalize_count")(_C.ptr(sums), _C.ptr(count_slot), _C.ptr(gamma), _C.ptr(beta), _C.ptr(running_mean),
                                           _C.ptr(running_var), float(momentum), float(eps), _C.ptr(mean), _C.ptr(invstd),
                                           _C.ptr(scale), _C.ptr(shift), c, _C.ptr(num_batches_tracked), _C.ptr(cnt), _C.stream()),
             "rr_bn_finalize_count")
    return mean, invstd, scale, shift, cnt


def bn_affine_grad(sums, dgamma, dbeta):
    """dbeta += sums[:c], dgamma += sums[c:2c] (the LOCAL BatchNorm-backward sums, before their SyncBN exchange): one launch."""
    c = dgamma.numel()
    assert dgamma.is_contiguous() and dbeta.is_contiguous() and dgamma.dtype == torch.float32
    _C.check(_C.fn("rr_bn_affine_grad")(_C.ptr(sums), _C.ptr(dgamma), _C.ptr(dbeta), c, _C.stream()), "rr_bn_affine_grad")


def bn_finalize(sums, count, gamma, beta, running_mean, running_var, momentum, eps, count_dev=None,
                num_batches_tracked=None):
    c = gamma.numel()
    dev = gamma.device
    buf = _f32(4 * c, dev)                       # mean | invstd | scale | shift: one allocation
    mean, invstd, scale, shift = buf[0:c], buf[c:2 * c], buf[2 * c:3 * c], buf[3 * c:4 * c]
    assert num_batches_tracked is None or num_batches_tracked.dtype == torch.int64
    _C.check(_C.fn("rr_bn_finalize")(_C.ptr(sums), float(count), _C.ptr(count_dev), _C.ptr(gamma), _C.ptr(beta), _C.ptr(running_mean),
                                     _C.ptr(running_var), float(momentum), float(eps), _C.ptr(mean), _C.ptr(invstd),
                                     _C.ptr(scale), _C.ptr(shift), c, _C.ptr(num_batches_tracked), _C.stream()),
             "rr_bn_finalize")
    return mean, invstd, scale, shift


def bn_stats_finalize(slab, count, gamma, beta, running_mean, running_var, momentum, eps, num_batches_tracked=None):
    """bn_reduce_slab + bn_finalize in one launch (single process: nothing to exchange in between)."""
    c = gamma.numel()
    buf = _f32(4 * c, gamma.device)
    mean, invstd, scale, shift = buf[0:c], buf[c:2 * c], buf[2 * c:3 * c], buf[3 * c:4 * c]
    mtiles = slab.numel() // (2 * c)
    assert num_batches_tracked is None or num_batches_tracked.dtype == torch.int64
    _C.check(_C.fn("rr_bn_stats_finalize")(_C.ptr(slab), mtiles, float(count), _C.ptr(gamma), _C.ptr(beta),
                                           _C.ptr(running_mean), _C.ptr(running_var), float(momentum), float(eps),
                                           _C.ptr(mean), _C.ptr(invstd), _C.ptr(scale), _C.ptr(shift), c,
                                           _C.ptr(num_batches_tracked), _C.stream()), "rr_bn_stats_finalize")
    return mean, invstd, scale, shift


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    c = gamma.numel()
    scale, shift = _f32(c, gamma.device), _f32(c, gamma.device)
    _C.check(_C.fn("rr_bn_eval_coeffs")(_C.ptr(gamma), _C.ptr(beta), _C.ptr(running_mean), _C.ptr(running_var),
                                        float(eps), _C.ptr(scale), _C.ptr(shift), c, _C.stream()), "rr_bn_eval_coeffs")
    return scale, shift


def bn_apply(y, scale, shift, residual=None, relu=False, res_scale=None, res_shift=None, bf16_only=False):
    """bf16_only (conv16, ops.phantom_scope): the output is written as a bf16 image only and returned as a memory-less fp32
    handle (phantom_f32).  A bf16-only residual is read from its image."""
    res_ph, y_ph = is_phantom(residual), is_phantom(y)
    assert (is_nhwc(y) or y_ph) and (residual is None or ((is_nhwc(residual) or res_ph) and residual.shape == y.shape))
    n, c, h, w = y.shape
    want16 = _CONV16 and _mode() == MATH_BF16 and y.is_cuda and c % 128 == 0 and n * h * w >= _CONV16_MIN_PIXELS
    if (bf16_only or res_ph or y_ph) and y.is_cuda and c % 4 == 0:
        out = None if bf16_only else empty_nhwc(n, c, h, w, y.device)
        out16 = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=y.device).permute(0, 3, 1, 2) if (bf16_only or want16) else None
        _C.check(_C.fn("rr_bn_apply_b16")(_C.ptr(None if y_ph else y), _C.ptr(image_of(y) if y_ph else None), _C.ptr(scale), _C.ptr(shift),
                                          _C.ptr(None if res_ph else residual), _C.ptr(image_of(residual) if res_ph else None),
                                          _C.ptr(res_scale), _C.ptr(res_shift), _C.ptr(out), _C.ptr(out16), y.numel(), c, int(relu),
                                          _C.stream()), "rr_bn_apply_b16")
        if bf16_only:
            return phantom_f32((n, c, h, w), y.device, out16)
        if out16 is not None:
            b16_attach(out, out16)
        return out
    out = empty_nhwc(n, c, h, w, y.device)
    if _mode() == MATH_F16X3 and y.is_cuda and n * h * w >= _SPLIT_MIN_PIXELS and _FUSED_AMAX_FWD:
        # split-operand convolutions: the consumer's operand scale comes out of this pass (see amax_of)
        word = _ZEROS.take(1, y.device)
        _C.check(_C.fn("rr_bn_apply_amax")(_C.ptr(y), _C.ptr(scale), _C.ptr(shift), _C.ptr(residual), _C.ptr(res_scale),
                                           _C.ptr(res_shift), _C.ptr(out), y.numel(), c, int(relu), _C.ptr(word), _C.stream()),
                 "rr_bn_apply_amax")
        out._rr_amax = (out._version, torch.cuda.current_stream(y.device).cuda_stream, word)
        return out
    if want16:
        # the bf16 image the consuming convolution (csrc/conv16.hip) reads, written in the same pass
        out16 = torch.empty_like(out, dtype=torch.bfloat16)
        _C.check(_C.fn("rr_bn_apply_b16")(_C.ptr(y), None, _C.ptr(scale), _C.ptr(shift), _C.ptr(residual), None, _C.ptr(res_scale),
                                          _C.ptr(res_shift), _C.ptr(out), _C.ptr(out16), y.numel(), c, int(relu), _C.stream()),
                 "rr_bn_apply_b16")
        b16_attach(out, out16)
        return out
    _C.check(_C.fn("rr_bn_apply")(_C.ptr(y), _C.ptr(scale), _C.ptr(shift), _C.ptr(residual), _C.ptr(res_scale),
                                  _C.ptr(res_shift), _C.ptr(out), y.numel(), c, int(relu), _C.stream()), "rr_bn_apply")
    return out


def bn_bwd_reduce(dz, z, y, mean, invstd, extra=0, mask_scale=None, mask_shift=None):
    n, c, h, w = y.shape
    sums = _ZEROS.take(2 * c + extra, y.device)            # pre-zeroed pool slice: no memset launch per layer
    z_ph, y_ph = is_phantom(z), is_phantom(y)
    if z_ph or y_ph:                                        # the layer's output / pre-BN output exist only as bf16 images
        _C.check(_C.fn("rr_bn_bwd_reduce_b16")(_C.ptr(dz), _C.ptr(None if z_ph else z), _C.ptr(image_of(z) if z_ph else None),
                                               _C.ptr(None if y_ph else y), _C.ptr(image_of(y) if y_ph else None), _C.ptr(mean),
                                               _C.ptr(invstd), _C.ptr(mask_scale), _C.ptr(mask_shift), _C.ptr(sums), n * h * w, c,
                                               _C.stream()), "rr_bn_bwd_reduce_b16")
        return sums
    _C.check(_C.fn("rr_bn_bwd_reduce")(_C.ptr(dz), _C.ptr(z), _C.ptr(y), _C.ptr(mean), _C.ptr(invstd),
                                       _C.ptr(mask_scale), _C.ptr(mask_shift), _C.ptr(sums),
                                       n * h * w, c, 1, _C.stream()), "rr_bn_bwd_reduce")
    return sums


def bn_bwd_apply(dz, z, y, mean, invstd, gamma, sums, count, want_g=False, dgamma=None, dbeta=None, count_dev=None,
                 mask_scale=None, mask_shift=None, g_into=None, bf16_only=False):
    """-> (dx, g).  g_into: an existing gradient buffer of y's shape that the masked gradient is ADDED to (returned as g).
    bf16_only (conv16): dx is written as a bf16 image only — the caller knows that its data and weight gradient both read
    that image; the returned dx is a memory-less fp32 handle (phantom_f32)."""
    n, c, h, w = y.shape
    z_ph, y_ph = is_phantom(z), is_phantom(y)
    only16 = bool(bf16_only and _CONV16 and _mode() == MATH_BF16 and y.is_cuda and c % 4 == 0)
    if only16 or z_ph or y_ph:
        if g_into is not None:
            assert is_nhwc(g_into) and g_into.shape == y.shape
            amax_drop(g_into)
            g = g_into
        else:
            g = empty_nhwc(n, c, h, w, y.device) if want_g else None
        want16 = only16 or (_CONV16 and _mode() == MATH_BF16 and c % 128 == 0 and n * h * w >= _CONV16_MIN_PIXELS)
        dx = None if only16 else empty_nhwc(n, c, h, w, y.device)
        dx16 = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=y.device).permute(0, 3, 1, 2) if want16 else None
        _C.check(_C.fn("rr_bn_bwd_apply_b16")(_C.ptr(dz), _C.ptr(None if z_ph else z), _C.ptr(image_of(z) if z_ph else None),
                                              _C.ptr(None if y_ph else y), _C.ptr(image_of(y) if y_ph else None),
                                              _C.ptr(mean), _C.ptr(invstd), _C.ptr(gamma), _C.ptr(mask_scale), _C.ptr(mask_shift),
                                              _C.ptr(sums), float(count), _C.ptr(count_dev), _C.ptr(dx), _C.ptr(dx16), _C.ptr(g),
                                              int(g_into is not None), _C.ptr(dgamma), _C.ptr(dbeta), y.numel(), c, _C.stream()),
                 "rr_bn_bwd_apply_b16")
        if only16:
            return phantom_f32((n, c, h, w), y.device, dx16), g
        if dx16 is not None:
            b16_attach(dx, dx16)
        return dx, g
    dx = empty_nhwc(n, c, h, w, y.device)
    if g_into is not None:
        assert is_nhwc(g_into) and g_into.shape == y.shape
        amax_drop(g_into)             # added into through its pointer
        g = g_into
    else:
        g = empty_nhwc(n, c, h, w, y.device) if want_g else None
    if _mode() == MATH_F16X3 and y.is_cuda and n * h * w >= _SPLIT_MIN_PIXELS and _FUSED_AMAX_BWD:
        # split-operand convolutions: dx is the operand of the data / weight gradient launched next — its maximum comes out
        # of this pass.  (dx is a fresh tensor that nothing adds into later: the remembered maximum cannot go stale.)
        word = _ZEROS.take(1, y.device)
        _C.check(_C.fn("rr_bn_bwd_apply_amax")(_C.ptr(dz), _C.ptr(z), _C.ptr(y), _C.ptr(mean), _C.ptr(invstd), _C.ptr(gamma),
                                               _C.ptr(mask_scale), _C.ptr(mask_shift), _C.ptr(sums), float(count), _C.ptr(count_dev),
                                               _C.ptr(dx), _C.ptr(g), int(g_into is not None), _C.ptr(dgamma), _C.ptr(dbeta),
                                               y.numel(), c, _C.ptr(word), _C.stream()), "rr_bn_bwd_apply_amax")
        dx._rr_amax = (dx._version, torch.cuda.current_stream(y.device).cuda_stream, word)
        return dx, g
    if _CONV16 and _mode() == MATH_BF16 and y.is_cuda and c % 128 == 0 and n * h * w >= _CONV16_MIN_PIXELS:
        dx16 = torch.empty_like(dx, dtype=torch.bfloat16)
        _C.check(_C.fn("rr_bn_bwd_apply_b16")(_C.ptr(dz), _C.ptr(z), None, _C.ptr(y), None, _C.ptr(mean), _C.ptr(invstd), _C.ptr(gamma),
                                              _C.ptr(mask_scale), _C.ptr(mask_shift), _C.ptr(sums), float(count), _C.ptr(count_dev),
                                              _C.ptr(dx), _C.ptr(dx16), _C.ptr(g), int(g_into is not None), _C.ptr(dgamma), _C.ptr(dbeta),
                                              y.numel(), c, _C.stream()), "rr_bn_bwd_apply_b16")
        b16_attach(dx, dx16)
        return dx, g
    _C.check(_C.fn("rr_bn_bwd_apply_gacc" if g_into is not None else "rr_bn_bwd_apply")(_C.ptr(dz), _C.ptr(z), _C.ptr(y), _C.ptr(mean), _C.ptr(invstd), _C.ptr(gamma),
                                      _C.ptr(mask_scale), _C.ptr(mask_shift), _C.ptr(sums), float(count), _C.ptr(count_dev), _C.ptr(dx), _C.ptr(g), _C.ptr(dgamma), _C.ptr(dbeta),
                                      y.numel(), c, _C.stream()), "rr_bn_bwd_apply")
    return dx, g


def relu_fwd(x):
    out = torch.empty_like(x)
    assert out.stride() == x.stride()
    _C.check(_C.fn("rr_relu_fwd")(_C.ptr(x), _C.ptr(out), x.numel(), _C.stream()), "rr_relu_fwd")
    return out


def sum_n(grads, z=None):
    """(sum of same-layout tensors) * (z > 0 if z is given) in one pass."""
    import ctypes
    g0 = grads[0]
    for g in grads:
        assert g.shape == g0.shape and g.stride() == g0.stride() and g.is_cuda
    out = torch.empty_like(g0)
    assert out.stride() == g0.stride()
    arr = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
    _C.check(_C.fn("rr_sum_n")(ctypes.cast(arr, ctypes.c_void_p), len(grads), _C.ptr(z), _C.ptr(out), g0.numel(),
                               _C.stream()), "rr_sum_n")
    return out


def bias_relu_bwd(dy, z, dbias):
    """dy NHWC-memory [*, c]; returns dy*(z>0) (or dy itself when z is None) and adds column sums to dbias."""
    c = dbias.numel()
    npix = dy.numel() // c
    masked = torch.empty_like(dy) if z is not None else None
    _C.check(_C.fn("rr_bias_relu_bwd")(_C.ptr(dy), _C.ptr(z), _C.ptr(masked), _C.ptr(dbias), npix, c, _C.stream()),
             "rr_bias_relu_bwd")
    return masked if z is not None else dy


def upsample_add_fwd(up1, low, bf16_only=False):
    """up1 + nearest-2x(low) (bilinear-align-corners resize for odd sizes).  Either operand may be a bf16-only activation;
    bf16_only: the sum is written as a bf16 image only (phantom_f32)."""
    n, c, h, w = up1.shape
    fast = (2 * low.shape[2] == h and 2 * low.shape[3] == w and c % 4 == 0 and up1.is_cuda)
    if fast and (bf16_only or is_phantom(up1) or is_phantom(low)):
        u_ph, l_ph = is_phantom(up1), is_phantom(low)
        want16 = bf16_only or (_CONV16 and _mode() == MATH_BF16 and c % 128 == 0 and n * h * w >= _CONV16_MIN_PIXELS)
        out = None if bf16_only else empty_nhwc(n, c, h, w, up1.device)
        out16 = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=up1.device).permute(0, 3, 1, 2) if want16 else None
        _C.check(_C.fn("rr_upsample2x_add_b16")(_C.ptr(None if u_ph else up1), _C.ptr(image_of(up1) if u_ph else None),
                                                _C.ptr(None if l_ph else low), _C.ptr(image_of(low) if l_ph else None),
                                                _C.ptr(out), _C.ptr(out16), n, h, w, c, _C.stream()), "rr_upsample2x_add_b16")
        if bf16_only:
            return phantom_f32((n, c, h, w), up1.device, out16)
        if out16 is not None:
            b16_attach(out, out16)
        return out
    up1, low = f32_of(up1), f32_of(low)
    out = empty_nhwc(n, c, h, w, up1.device)
    _C.check(_C.fn("rr_upsample_add_fwd")(_C.ptr(up1), _C.ptr(low), _C.ptr(out), n, h, w, low.shape[2], low.shape[3],
                                          c, _C.stream()), "rr_upsample_add_fwd")
    return out


def upsample_add_bwd(dout, low_shape):
    n, c, h, w = dout.shape
    dlow = empty_nhwc(n, c, low_shape[2], low_shape[3], dout.device)
    _C.check(_C.fn("rr_upsample_add_bwd")(_C.ptr(dout), _C.ptr(dlow), n, h, w, low_shape[2], low_shape[3], c,
                                          _C.stream()), "rr_upsample_add_bwd")
    return dlow


def resize_bilinear_ac(x, scale_factor):
    """F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=True) on an NHWC tensor."""
    import math
    x = to_nhwc(x)
    n, c, h, w = x.shape
    oh, ow = int(math.floor(h * scale_factor)), int(math.floor(w * scale_factor))
    out = empty_nhwc(n, c, oh, ow, x.device)
    _C.check(_C.fn("rr_resize_bilinear_ac")(_C.ptr(x), _C.ptr(out), n, h, w, oh, ow, c, _C.stream()),
             "rr_resize_bilinear_ac")
    return out


def avgpool_fwd(x):
    r, c, h, w = x.shape
    out = empty_nhwc(r, c, 1, 1, x.device)
    _C.check(_C.fn("rr_avgpool_fwd")(_C.ptr(x), _C.ptr(out), r, h * w, c, _C.stream()), "rr_avgpool_fwd")
    return out


def bn_res_relu_avgpool(y, scale, shift, res):
    """[R,C,h,w] (NHWC) x2 -> [R,C,1,1]: mean_p relu(y*scale + shift + res), one pass (inference)."""
    assert is_nhwc(y) and is_nhwc(res) and y.shape == res.shape
    r, c, h, w = y.shape
    out = empty_nhwc(r, c, 1, 1, y.device)
    _C.check(_C.fn("rr_bn_res_relu_avgpool")(_C.ptr(y), _C.ptr(scale), _C.ptr(shift), _C.ptr(res), _C.ptr(out), r, h * w, c,
                                             _C.stream()), "rr_bn_res_relu_avgpool")
    return out


def conv1x1_bn_res_relu_avgpool(h, w, scale, shift, res):
    """h [R,K,ph,pw], w [N,K,1,1], res [R,N,ph,pw] (all NHWC) -> [R,N,1,1]: mean_p relu(conv1x1(h)*scale + shift + res),
    one kernel, the convolution's output stays in LDS (inference)."""
    assert is_nhwc(h) and is_nhwc(res) and is_nhwc(w)
    r, k, ph, pw = h.shape
    n = w.shape[0]
    assert tuple(w.shape[1:]) == (k, 1, 1) and tuple(res.shape) == (r, n, ph, pw)
    out = empty_nhwc(r, n, 1, 1, h.device)
    _C.check(_C.fn("rr_conv1x1_bn_res_relu_avgpool")(_C.ptr(h), _C.ptr(w), _C.ptr(scale), _C.ptr(shift), _C.ptr(res), _C.ptr(out),
                                                     r, ph * pw, k, n, _C.stream()), "rr_conv1x1_bn_res_relu_avgpool")
    return out


def avgpool_bwd(dout, shape):
    r, c, h, w = shape
    dx = empty_nhwc(r, c, h, w, dout.device)
    _C.check(_C.fn("rr_avgpool_bwd")(_C.ptr(dout), _C.ptr(dx), r, h * w, c, _C.stream()), "rr_avgpool_bwd")
    return dx


def wh_shift_sum_fwd(t, bias_w, bias_h, k):
    n, c, h, w = t.shape
    assert c >= 2 * k and is_nhwc(t)
    out = empty_nhwc(n, 2, h, w, t.device)
    _C.check(_C.fn("rr_wh_shift_sum_fwd")(_C.ptr(t), _C.ptr(bias_w), _C.ptr(bias_h), _C.ptr(out), n, h, w, k, c,
                                          _C.stream()), "rr_wh_shift_sum_fwd")
    return out


def wh_shift_sum_bwd(dout, k, ct):
    n, _, h, w = dout.shape
    dt = empty_nhwc(n, ct, h, w, dout.device)
    _C.check(_C.fn("rr_wh_shift_sum_bwd")(_C.ptr(dout), _C.ptr(dt), n, h, w, k, ct, _C.stream()), "rr_wh_shift_sum_bwd")
    return dt


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0):
    assert param.numel() % 4 == 0
    _C.check(_C.fn("rr_adam_step")(_C.ptr(param), _C.ptr(grad), _C.ptr(exp_avg), _C.ptr(exp_avg_sq), param.numel(),
                                   float(lr), float(beta1), float(beta2), float(eps), int(step), float(grad_scale),
                                   _C.stream()), "rr_adam_step")


# ---------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------
def focal_fwd(logits, gt):
    sums = torch.empty(3, dtype=torch.float64, device=logits.device)
    _C.check(_C.fn("rr_focal_loss_fwd")(_C.ptr(logits), _C.ptr(gt), logits.numel(), _C.ptr(sums), _C.stream()),
             "rr_focal_loss_fwd")
    return sums


def focal_bwd(logits, gt, sums, gout, gscale=1.0):
    d = torch.empty_like(logits)
    assert d.stride() == logits.stride()
    _C.check(_C.fn("rr_focal_loss_bwd")(_C.ptr(logits), _C.ptr(gt), logits.numel(), _C.ptr(sums), _C.ptr(gout),
                                        float(gscale), _C.ptr(d), _C.stream()), "rr_focal_loss_bwd")
    return d


def regl1_fwd(pred, mask, ind, target):
    b, c, h, w = pred.shape
    m = ind.shape[1]
    sums = torch.empty(3, dtype=torch.float64, device=pred.device)
    _C.check(_C.fn("rr_regl1_fwd")(_C.ptr(pred), _C.ptr(mask), _C.ptr(ind), _C.ptr(target), b, m, c, h * w, _C.ptr(sums),
                                   _C.stream()), "rr_regl1_fwd")
    return sums


def regl1_bwd(pred, mask, ind, target, sums, gout, gscale=1.0):
    b, c, h, w = pred.shape
    m = ind.shape[1]
    d = empty_nhwc(b, c, h, w, pred.device)
    _C.check(_C.fn("rr_regl1_bwd")(_C.ptr(pred), _C.ptr(mask), _C.ptr(ind), _C.ptr(target), b, m, c, h * w, _C.ptr(sums),
                                   _C.ptr(gout), float(gscale), _C.ptr(d), _C.stream()), "rr_regl1_bwd")
    return d


def stage2_loss(rois, reg, gt_xyxy, scale, want_droi=False):
    """rois [R,5], reg [R,4], gt [B,G,>=4] (xyxy) -> loss (double[3], [0] is the loss), dreg_unit [R,4], tgt, pos,
    npos, droi_unit [R,4] | None."""
    r = rois.shape[0]
    b, g, gs = gt_xyxy.shape
    dev = rois.device
    tgt = torch.empty((max(r, 1), 4), dtype=torch.float32, device=dev)
    pos = torch.zeros(max(r, 1), dtype=torch.int32, device=dev)
    npos = torch.empty(b, dtype=torch.int32, device=dev)
    loss = torch.empty(3, dtype=torch.float64, device=dev)
    dreg = torch.zeros((r, 4), dtype=torch.float32, device=dev)
    droi = torch.zeros((r, 4), dtype=torch.float32, device=dev) if want_droi else None
    _C.check(_C.fn("rr_stage2_loss")(_C.ptr(rois), _C.ptr(reg), r, _C.ptr(gt_xyxy), b, g, gs, float(scale), _C.ptr(tgt),
                                     _C.ptr(pos), _C.ptr(npos), _C.ptr(loss), _C.ptr(dreg), _C.ptr(droi), _C.stream()),
             "rr_stage2_loss")
    return loss, dreg, tgt[:r], pos[:r], npos, droi


# ---------------------------------------------------------------------------------------------
# decode / NMS / RoIAlign
# ---------------------------------------------------------------------------------------------
def decode_topk(hm, wh, off, k, is_logits=True, want_pix=False, peak_filter=False, spread=True, box_mode=0, scale=1.0):
    """spread=False keeps the whole decode in one workgroup per frame (no workspace): the A/B of the two paths.
    box_mode 0: RRNet rows x1,y1,x2,y2,score,cls (feature coordinates, wh clamped at 0); 1: CenterNet rows
    (x,y,w,h)*scale,score,cls+1 (no clamp)."""
    assert is_nhwc(hm) and is_nhwc(wh) and is_nhwc(off)
    b, c, h, w = hm.shape
    out = torch.empty((b, k, 6), dtype=torch.float32, device=hm.device)
    pix = torch.empty((b, k), dtype=torch.int32, device=hm.device) if want_pix else None
    ws, ws_bytes = None, 0
    if spread and h * w * c >= 65536:
        ws_bytes = _C.fn("rr_decode_workspace_bytes")(b)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=hm.device)
    _C.check(_C.fn("rr_decode_topk")(_C.ptr(hm), int(is_logits), int(peak_filter), _C.ptr(wh), _C.ptr(off), b, h, w, c, k,
                                     int(box_mode), float(scale), _C.ptr(out), _C.ptr(pix), _C.ptr(ws), ws_bytes,
                                     _C.stream()), "rr_decode_topk")
    return (out, pix) if want_pix else out


def roi_provenance(rois, scores, clses, decoded, pix):
    r = rois.shape[0]
    roi_pix = torch.full((max(r, 1),), -1, dtype=torch.int32, device=rois.device)
    _C.check(_C.fn("rr_roi_provenance")(_C.ptr(rois), _C.ptr(scores), _C.ptr(clses), r, _C.ptr(decoded), _C.ptr(pix),
                                        decoded.shape[1], _C.ptr(roi_pix), _C.stream()), "rr_roi_provenance")
    return roi_pix[:r]


def proposal_bwd(droi, rois, roi_pix, wh):
    b, _, h, w = wh.shape
    dwh, doff = empty_nhwc(b, 2, h, w, wh.device), empty_nhwc(b, 2, h, w, wh.device)
    _C.check(_C.fn("rr_proposal_bwd")(_C.ptr(droi), _C.ptr(rois), _C.ptr(roi_pix), rois.shape[0], _C.ptr(wh), b, h, w,
                                      _C.ptr(dwh), _C.ptr(doff), _C.stream()), "rr_proposal_bwd")
    return dwh, doff


def peak3x3(hm, is_logits=True):
    """`_ctnet_nms` score map: sigmoid(hm) (or hm itself when is_logits=False) where it equals its 3x3 maximum, else 0."""
    b, c, h, w = hm.shape
    out = empty_nhwc(b, c, h, w, hm.device)
    _C.check(_C.fn("rr_peak3x3")(_C.ptr(hm), int(is_logits), _C.ptr(out), b, h, w, c, _C.stream()), "rr_peak3x3")
    return out


def group_by_class(boxes, num_classes, cls_base=0):
    """boxes [B,K,6] -> grouped [B,K,6], seg_off int32 [B*num_classes+1], seg_len int32 [B*num_classes].
    Rows whose class lies outside [cls_base, cls_base+num_classes) are dropped (a gap of uninitialised rows at the
    end of the image's block): hand seg_len, not offset differences, to the NMS entries."""
    b, k, _ = boxes.shape
    grouped = torch.empty_like(boxes)
    seg_off = torch.empty(b * num_classes + 1, dtype=torch.int32, device=boxes.device)
    seg_len = torch.empty(b * num_classes, dtype=torch.int32, device=boxes.device)
    _C.check(_C.fn("rr_group_by_class")(_C.ptr(boxes), b, k, num_classes, cls_base, _C.ptr(grouped), _C.ptr(seg_off),
                                        _C.ptr(seg_len), _C.stream()), "rr_group_by_class")
    return grouped, seg_off, seg_len


def hard_nms_segments(boxes6, seg_off, max_seg, thresh, seg_len=None):
    nseg = seg_off.numel() - 1
    n_out = torch.zeros(nseg, dtype=torch.int32, device=boxes6.device)
    _C.check(_C.fn("rr_hard_nms_segments")(_C.ptr(boxes6), _C.ptr(seg_off), _C.ptr(seg_len), nseg, int(max_seg),
                                           float(thresh), _C.ptr(n_out), _C.stream()), "rr_hard_nms_segments")
    return n_out


def seg_prefix(n_out):
    """Exclusive prefix of per-segment counts -> int32 [nseg+1] (last entry = total)."""
    nseg = n_out.numel()
    out_off = torch.empty(nseg + 1, dtype=torch.int32, device=n_out.device)
    nul = _C.c_void_p(0)
    _C.check(_C.fn("rr_pack_segments")(nul, nul, _C.ptr(n_out), nseg, 1, _C.ptr(out_off), nul, nul, nul, nul, 0,
                                       _C.stream()), "rr_pack_segments")
    return out_off


def pack_segments(grouped, seg_off, n_out, segs_per_image, want_rois=True, want_rows=False, want_offsets=False):
    """-> (rois [R,5], scores [R], clses [R]) and/or rows [R,6]; one host sync to learn R."""
    nseg = n_out.numel()
    dev = grouped.device
    f = _C.fn("rr_pack_segments")
    out_off = seg_prefix(n_out)
    import time as _time
    t0 = _time.perf_counter()
    r = int(out_off[-1].item())              # the step's one device -> host read: the RoI count sizes the head's tensors
    global SYNC_WAIT_S
    SYNC_WAIT_S += _time.perf_counter() - t0
    rois = torch.empty((r, 5), dtype=torch.float32, device=dev) if want_rois else None
    scores = torch.empty(r, dtype=torch.float32, device=dev) if want_rois else None
    clses = torch.empty(r, dtype=torch.float32, device=dev) if want_rois else None
    rows = torch.empty((r, 6), dtype=torch.float32, device=dev) if want_rows else None
    _C.check(f(_C.ptr(grouped), _C.ptr(seg_off), _C.ptr(n_out), nseg, segs_per_image, _C.ptr(out_off), _C.ptr(rois),
               _C.ptr(scores), _C.ptr(clses), _C.ptr(rows), 1, _C.stream()), "rr_pack_segments")
    if want_offsets:
        return rois, scores, clses, rows, out_off
    return rois, scores, clses, rows


def ctnet_targets(annos, counts, img_h, img_w, scale_factor=4, num_classes=10):
    """annos [B,M,>=6] cuda float32 (padded), counts [B] cuda int32 -> hm (logical [B,C,Hf,Wf], NHWC memory),
    wh [B,M,2], ind [B,M,1], offset [B,M,2], reg_mask [B,M,1]  (collate_fn_ctnet contract)."""
    _C.require_cuda(annos, counts)
    assert annos.dtype == torch.float32 and annos.is_contiguous() and counts.dtype == torch.int32
    b, m, st = annos.shape
    hf, wf = img_h // scale_factor, img_w // scale_factor
    dev = annos.device
    hm = empty_nhwc(b, num_classes, hf, wf, dev)
    wh = torch.empty((b, m, 2), dtype=torch.float32, device=dev)
    ind = torch.empty((b, m, 1), dtype=torch.float32, device=dev)
    off = torch.empty((b, m, 2), dtype=torch.float32, device=dev)
    mask = torch.empty((b, m, 1), dtype=torch.float32, device=dev)
    _C.check(_C.fn("rr_ctnet_targets")(_C.ptr(annos), _C.ptr(counts), b, m, st, img_h, img_w, scale_factor, num_classes,
                                       _C.ptr(hm), _C.ptr(wh), _C.ptr(ind), _C.ptr(off), _C.ptr(mask), _C.stream()),
             "rr_ctnet_targets")
    return hm, wh, ind, off, mask


def refine_boxes(rois, reg, scores, clses, seg_off, scale, score_thr):
    """generate_bbox + score filter + xywh->xyxy for every (frame, class) segment of the packed RoI list.
    -> boxes6 [R,6] (kept rows at the front of each segment's range), seg_len int32 [nseg]."""
    _C.require_cuda(rois, reg, scores, clses, seg_off)
    r = rois.shape[0]
    nseg = seg_off.numel() - 1
    out6 = torch.empty((r, 6), dtype=torch.float32, device=rois.device)
    seg_len = torch.zeros(max(nseg, 0), dtype=torch.int32, device=rois.device)
    _C.check(_C.fn("rr_refine_boxes")(_C.ptr(rois), _C.ptr(reg.contiguous()), _C.ptr(scores), _C.ptr(clses),
                                      _C.ptr(seg_off), nseg, float(scale), float(score_thr), _C.ptr(out6),
                                      _C.ptr(seg_len), _C.stream()), "rr_refine_boxes")
    return out6, seg_len


def finalize_frames(boxes6, seg_off, n_out, nframes, segs_per_frame, max_frame_boxes):
    """Per frame: concatenate kept segment rows, xyxy->xywh, sort by score descending.
    -> out6 [R,6] (R = upper bound rows; frame f occupies rows [frame_off[f], frame_off[f+1])), frame_off int32
    [nframes+1] (device)."""
    out_off = seg_prefix(n_out)
    out6 = torch.empty_like(boxes6)
    _C.check(_C.fn("rr_finalize_frames")(_C.ptr(boxes6), _C.ptr(seg_off), _C.ptr(n_out), _C.ptr(out_off), nframes,
                                         segs_per_frame, int(max_frame_boxes), _C.ptr(out6), _C.stream()),
             "rr_finalize_frames")
    return out6, out_off[::segs_per_frame]


def sort_rows_by_score(rows6):
    """[n,6] device rows -> a new tensor ordered by score descending (ties keep their input order)."""
    rows6 = rows6.contiguous()
    out = torch.empty_like(rows6)
    _C.check(_C.fn("rr_sort_rows_by_score")(_C.ptr(rows6), rows6.shape[0], _C.ptr(out), _C.stream()), "rr_sort_rows_by_score")
    return out


def roi_spatial_order(rois, frame_off):
    """Per-frame spatial processing order of the packed RoI list (frame_off int32 [B+1], device) -> int32 [R]."""
    r = rois.shape[0]
    order = torch.empty(max(r, 1), dtype=torch.int32, device=rois.device)
    _C.check(_C.fn("rr_roi_spatial_order")(_C.ptr(rois), _C.ptr(frame_off), frame_off.numel() - 1, _C.ptr(order),
                                           _C.stream()), "rr_roi_spatial_order")
    return order[:r]


def roi_align_fwd(feat, rois, out_size, spatial_scale=1.0, sampling_ratio=-1, order=None):
    assert is_nhwc(feat)
    b, c, h, w = feat.shape
    r = rois.shape[0]
    ph, pw = out_size
    out = empty_nhwc(r, c, ph, pw, feat.device)
    _C.check(_C.fn("rr_roi_align_fwd")(_C.ptr(feat), _C.ptr(rois), r, h, w, c, ph, pw, float(spatial_scale),
                                       int(sampling_ratio), _C.ptr(order), _C.ptr(out), _C.stream()), "rr_roi_align_fwd")
    return out


def roi_align_bwd(dout, rois, feat_shape, out_size, spatial_scale=1.0, sampling_ratio=-1):
    b, c, h, w = feat_shape
    r = rois.shape[0]
    ph, pw = out_size
    dfeat = empty_nhwc(b, c, h, w, dout.device)
    _C.check(_C.fn("rr_roi_align_bwd")(_C.ptr(dout), _C.ptr(rois), r, b, h, w, c, ph, pw, float(spatial_scale),
                                       int(sampling_ratio), _C.ptr(dfeat), _C.stream()), "rr_roi_align_bwd")
    return dfeat


# ---------------------------------------------------------------------------------------------
# DCNv2
# ---------------------------------------------------------------------------------------------
_DCN_WPACK = True
_DCN_DYB = True      # bf16 data gradient stages dY from a bf16 copy made once per call


def dcn_fwd(x, offset, mask, w, bias, stride, pad, dilation, dg, bf16=False):
    assert is_nhwc(x) and is_nhwc(offset) and is_nhwc(mask) and is_nhwc(w)
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    p, q = offset.shape[2], offset.shape[3]
    y = empty_nhwc(n, k, p, q, x.device)
    if bf16 and _DCN_WPACK:
        # weights re-packed to bf16 inside the call (one small kernel): the B operand then reaches LDS by DMA
        wpack = torch.empty(_C.fn("rr_dcn_wpack_bytes")(c, k, r, s), dtype=torch.uint8, device=x.device)
        _C.check(_C.fn("rr_dcn_fwd_bf16_packed")(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(w), _C.ptr(bias), _C.ptr(y),
                                                 n, h, wd, c, k, r, s, stride, pad[0], pad[1], dilation, dg, _C.ptr(wpack),
                                                 _C.stream()), "rr_dcn_fwd_bf16_packed")
        return y
    name = "rr_dcn_fwd_bf16" if bf16 else "rr_dcn_fwd"
    _C.check(_C.fn(name)(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(w), _C.ptr(bias), _C.ptr(y), n, h, wd,
                         c, k, r, s, stride, pad[0], pad[1], dilation, dg, _C.stream()), name)
    return y


def dcn_im2col(x, offset, mask, r, s, stride, pad, dilation, dg):
    n, c, h, wd = x.shape
    m = offset.shape[0] * offset.shape[2] * offset.shape[3]
    col = torch.empty((1, m, 1, r * s * c), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    _C.check(_C.fn("rr_dcn_im2col")(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(col), n, h, wd, c, r, s, stride,
                                    pad[0], pad[1], dilation, dg, _C.stream()), "rr_dcn_im2col")
    return col                                           # logical [1, r*s*c, M, 1], memory [M][r*s*c]


def dcn_col2im(x, offset, mask, dcol, r, s, stride, pad, dilation, dg):
    n, c, h, wd = x.shape
    dx = empty_nhwc(n, c, h, wd, x.device)
    doff = torch.empty_like(offset)
    dmask = torch.empty_like(mask)
    assert doff.stride() == offset.stride() and dmask.stride() == mask.stride()
    _C.check(_C.fn("rr_dcn_col2im")(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(dcol), _C.ptr(dx), _C.ptr(doff),
                                    _C.ptr(dmask), n, h, wd, c, r, s, stride, pad[0], pad[1], dilation, dg,
                                    _C.stream()), "rr_dcn_col2im")
    return dx, doff, dmask


def dcn_fused_bwd_supported(c, k, r, s, stride, dg, dilation=1):
    """Whether rr_dcn_wgrad / rr_dcn_dgrad take a layer of this shape (else: the column path)."""
    return bool(_C.fn("rr_dcn_fused_bwd_supported_dil")(c, k, r, s, stride, dilation, dg))


def dcn_wgrad(x, offset, mask, dy, dw, stride, pad, dilation, dg, bf16=False, dy_img=None):
    """dw [K,C,R,S] (OHWI memory, pre-zeroed or the running gradient) += dY^T x deformed columns; no column buffer.
    bf16: bf16 matrix operands (dY, samples), input window in LDS (rr_dcn_wgrad_bf16); dy_img: dY's bf16 image, complete on
    the current stream — the dY operand then reaches LDS by DMA (rr_dcn_wgrad_bf16_img)."""
    assert is_nhwc(x) and is_nhwc(offset) and is_nhwc(mask) and is_nhwc(dy) and is_nhwc(dw)
    n, c, h, wd = x.shape
    k, _, r, s = dw.shape
    if bf16 and dy_img is not None and _DCN_DYB:
        _C.check(_C.fn("rr_dcn_wgrad_bf16_img")(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(dy), _C.ptr(dy_img), _C.ptr(dw), n, h,
                                                wd, c, k, r, s, stride, pad[0], pad[1], dilation, dg, _C.stream()),
                 "rr_dcn_wgrad_bf16_img")
        return dw
    name = "rr_dcn_wgrad_bf16" if bf16 else "rr_dcn_wgrad"
    _C.check(_C.fn(name)(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(dy), _C.ptr(dw), n, h, wd, c, k, r, s,
                         stride, pad[0], pad[1], dilation, dg, _C.stream()), name)
    return dw


def dcn_dgrad_accumulates(bf16):
    """True when dcn_dgrad(..., out=buf) adds into buf inside the kernel (the LDS-DMA data gradient of the bf16 path)."""
    return bool(bf16 and _DCN_DYB and _DCN_WPACK)


def dcn_dgrad(x, offset, mask, w, dy, stride, pad, dilation, dg, bf16=False, out=None):
    """-> dx, doffset, dmask; the column gradient never leaves the MFMA accumulators / LDS.  bf16: bf16 matrix operands
    (dY, W) and d input pre-summed in an LDS window before the global atomics (rr_dcn_dgrad_bf16)."""
    assert is_nhwc(x) and is_nhwc(offset) and is_nhwc(mask) and is_nhwc(dy) and is_nhwc(w)
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    # out: a buffer that already holds another consumer's gradient of x — d input is ADDED to it (dcn_dgrad_accumulates only)
    assert out is None or (dcn_dgrad_accumulates(bf16) and is_nhwc(out) and tuple(out.shape) == tuple(x.shape))
    dx = out if out is not None else empty_nhwc(n, c, h, wd, x.device)
    if out is not None:
        amax_drop(out)                # an existing tensor added into through its pointer
    doff = torch.empty_like(offset)
    dmask = torch.empty_like(mask)
    assert doff.stride() == offset.stride() and dmask.stride() == mask.stride()
    img = b16_carry(dy) if (bf16 and _DCN_DYB) else None
    if img is not None and getattr(dy, "_rr_b16")[1] not in (None, torch.cuda.current_stream(dy.device).cuda_stream):
        img = None
    if bf16 and _DCN_DYB and _DCN_WPACK:
        # both sweep operands by LDS-DMA: weights packed to bf16 inside the call; dY = its producer's bf16 image when there is
        # one (the heads' 1x1 data gradient), else rounded once into the workspace
        ws = torch.empty(_C.fn("rr_dcn_dgrad_ws_bytes")(dy.shape[0], dy.shape[2], dy.shape[3], c, k, r, s, int(img is not None)),
                         dtype=torch.uint8, device=x.device)
        _C.check(_C.fn("rr_dcn_dgrad_bf16_packed")(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(w), _C.ptr(dy),
                                                   _C.ptr(img) if img is not None else None, _C.ptr(dx), _C.ptr(doff), _C.ptr(dmask),
                                                   n, h, wd, c, k, r, s, stride, pad[0], pad[1], dilation, dg, int(out is not None),
                                                   _C.ptr(ws), _C.stream()), "rr_dcn_dgrad_bf16_packed")
        return dx, doff, dmask
    if img is not None:
        # dY's producer already left its bf16 image (the heads' 1x1 data gradient): no conversion pass
        _C.check(_C.fn("rr_dcn_dgrad_bf16_img")(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(w), _C.ptr(dy), _C.ptr(img), _C.ptr(dx),
                                                _C.ptr(doff), _C.ptr(dmask), n, h, wd, c, k, r, s, stride, pad[0], pad[1],
                                                dilation, dg, _C.stream()), "rr_dcn_dgrad_bf16_img")
        return dx, doff, dmask
    if bf16 and _DCN_DYB:
        # dY rounded to bf16 once per call (caller scratch): the sweep's eight re-reads of a block's dY tile stay in L2
        dyb = torch.empty(_C.fn("rr_dcn_dyb_bytes")(dy.shape[0], dy.shape[2], dy.shape[3], k), dtype=torch.uint8, device=x.device)
        _C.check(_C.fn("rr_dcn_dgrad_bf16_ws")(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(w), _C.ptr(dy), _C.ptr(dx),
                                               _C.ptr(doff), _C.ptr(dmask), n, h, wd, c, k, r, s, stride, pad[0], pad[1],
                                               dilation, dg, _C.ptr(dyb), _C.stream()), "rr_dcn_dgrad_bf16_ws")
        return dx, doff, dmask
    name = "rr_dcn_dgrad_bf16" if bf16 else "rr_dcn_dgrad"
    _C.check(_C.fn(name)(_C.ptr(x), _C.ptr(offset), _C.ptr(mask), _C.ptr(w), _C.ptr(dy), _C.ptr(dx), _C.ptr(doff),
                         _C.ptr(dmask), n, h, wd, c, k, r, s, stride, pad[0], pad[1], dilation, dg, _C.stream()), name)
    return dx, doff, dmask


def dcn_split_fwd(om):
    """om [N, 3t, P, Q] (NHWC memory) -> offset [N, 2t, P, Q], mask [N, t, P, Q] = sigmoid(last third)."""
    assert is_nhwc(om) and om.shape[1] % 3 == 0
    n, ch, p, q = om.shape
    t = ch // 3
    offset, mask = empty_nhwc(n, 2 * t, p, q, om.device), empty_nhwc(n, t, p, q, om.device)
    _C.check(_C.fn("rr_dcn_split_fwd")(_C.ptr(om), n * p * q, t, _C.ptr(offset), _C.ptr(mask), _C.stream()), "rr_dcn_split_fwd")
    return offset, mask


def dcn_split_bwd(doffset, dmask, mask):
    n, t, p, q = mask.shape
    dom = empty_nhwc(n, 3 * t, p, q, mask.device)
    _C.check(_C.fn("rr_dcn_split_bwd")(_C.ptr(doffset), _C.ptr(dmask), _C.ptr(mask), n * p * q, t, _C.ptr(dom), _C.stream()),
             "rr_dcn_split_bwd")
    return dom


def dcn_psroi_fwd(x, rois, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part,
                  trans_std):
    """Deformable PS-RoI pooling.  x NHWC; rois [n,5]; trans [n,2*classes,part,part] (contiguous NCHW) or None.
    -> out, count: logical [n, output_dim, pooled, pooled], NHWC memory."""
    assert is_nhwc(x)
    b, c, h, w = x.shape
    n = rois.shape[0]
    out = empty_nhwc(n, output_dim, pooled_size, pooled_size, x.device)
    count = empty_nhwc(n, output_dim, pooled_size, pooled_size, x.device)
    tc = 0 if no_trans else trans.shape[1]
    _C.check(_C.fn("rr_dcn_psroi_fwd")(_C.ptr(x), _C.ptr(rois), _C.ptr(None if no_trans else trans), n, h, w, c, int(no_trans),
                                       float(spatial_scale), output_dim, group_size, pooled_size, part_size, sample_per_part,
                                       float(trans_std), tc, _C.ptr(out), _C.ptr(count), _C.stream()), "rr_dcn_psroi_fwd")
    return out, count


def dcn_psroi_bwd(dout, x, rois, trans, count, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
                  sample_per_part, trans_std):
    assert is_nhwc(x) and is_nhwc(dout)
    b, c, h, w = x.shape
    n = rois.shape[0]
    dx = empty_nhwc(b, c, h, w, x.device)
    dtrans = None if no_trans else torch.empty_like(trans)
    tc = 0 if no_trans else trans.shape[1]
    _C.check(_C.fn("rr_dcn_psroi_bwd")(_C.ptr(dout), _C.ptr(x), _C.ptr(rois), _C.ptr(None if no_trans else trans), _C.ptr(count),
                                       n, b, h, w, c, int(no_trans), float(spatial_scale), output_dim, group_size, pooled_size,
                                       part_size, sample_per_part, float(trans_std), tc, _C.ptr(dx), _C.ptr(dtrans),
                                       _C.stream()), "rr_dcn_psroi_bwd")
    return dx, dtrans


class _OpsModule(types.ModuleType):
    """`ops.BF16` (read / assign) = this thread's arithmetic switch."""

    @property
    def BF16(self):
        return _MODE.value

    @BF16.setter
    def BF16(self, v):
        _MODE.value = int(v or 0)


sys.modules[__name__].__class__ = _OpsModule
