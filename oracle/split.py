"""ORACLE (test infrastructure only) — numpy / torch-CPU model of the split-operand ("f16x3") convolution arithmetic of
csrc/conv_bf16.hip (rr_conv_*_f16x3, rr_weight_split_f16, rr_absmax_bits).

Only tests/ import this file.  The arithmetic is builder-defined (the reference multiplies in fp32: nn.Conv2d of
/root/reference/backbones/hourglass.py:12-61); what is modelled here is the kernels' own contract, step by step:

  scale(t)   = 2^(14 - floor(log2 max|t|))          power of two; the tensor's largest magnitude lands in [2^14, 2^15)
  hi, lo     = fp16(s*v), fp16(s*v - hi)            round-to-nearest-even both times; s*v and the subtraction exact in fp32
  a*b       ~= (ha*hb + ha*lb + la*hb) / (sa*sb)    each product of two fp16 values is exact in fp32; the dropped term la*lb
                                                    is <= 2^-22 |a*b|
  conv       = sum over (tap, channel) of those three products, accumulated in fp32 on the device; modelled in fp64 here
               (the model carries NO summation-order noise: what separates it from the kernel is fp32 accumulation only)."""
import numpy as np
import torch


def absmax_bits(t):
    """rr_absmax_bits: the bit pattern of max |t| (non-negative floats order as their bit patterns)."""
    m = np.float32(np.abs(np.asarray(t, dtype=np.float32)).max()) if np.asarray(t).size else np.float32(0)
    return int(np.array([m], dtype=np.float32).view(np.uint32)[0])


def scale_of(bits):
    """The kernels' scale from the maximum's bit pattern: 2^(14 - (e - 127)), e the biased exponent clamped to >= 15 — below that
    (a maximum under 2^-112, zero and denormals included) `(268 - e) << 23` would run into the Inf / NaN encodings
    (csrc/conv_bf16.hip: split_scale_from_bits)."""
    e = max((bits >> 23) & 0xff, 15)
    return np.array([(268 - e) << 23], dtype=np.uint32).view(np.float32)[0]


def split(v, scale):
    """split_bf4<2, true>: v (float32 array) -> (hi, lo) float16 arrays."""
    sv = (np.asarray(v, dtype=np.float32) * np.float32(scale)).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):          # non-finite inputs propagate (as in fp32)
        hi = sv.astype(np.float16)
        lo = (sv - hi.astype(np.float32)).astype(np.float32).astype(np.float16)
    return hi, lo


def conv2d(x, w, stride=1, pad=(0, 0)):
    """x [N,C,H,W], w [K,C,R,S] float32 -> y [N,K,P,Q] float64: the f16x3 arithmetic with exact accumulation."""
    x = np.asarray(x, dtype=np.float32)
    w = np.asarray(w, dtype=np.float32)
    sa, sb = scale_of(absmax_bits(x)), scale_of(absmax_bits(w))
    xh, xl = split(x, sa)
    wh, wl = split(w, sb)

    def c(a, b):
        return torch.nn.functional.conv2d(torch.from_numpy(a.astype(np.float64)), torch.from_numpy(b.astype(np.float64)), None,
                                          stride, tuple(pad)).numpy()
    y = c(xh, wh) + (c(xh, wl) + c(xl, wh))
    return y / (np.float64(sa) * np.float64(sb))
