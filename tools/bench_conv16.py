"""Micro-benchmark + parity of the 16-bit-activation convolutions (csrc/conv16.hip) against the fp32 kernels of csrc/conv.hip run on
the same bf16-rounded operands.
  python tools/bench_conv16.py [--shape n,c,h,w,k,r,stride] [--reps 20]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rrnet_amd import _C, ops  # noqa: E402


def to_bf16_nhwc(t):
    """logical NCHW fp32 (NHWC memory) -> bf16 tensor with the same layout."""
    return t.to(torch.bfloat16)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def run_shape(n, c, h, w, k, r, stride, reps=20, check=True):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    pad = (r // 2, r // 2)
    x = ops.to_nhwc(torch.randn((n, c, h, w), device=dev, generator=g).relu_())
    wt = ops.to_nhwc(torch.randn((k, c, r, r), device=dev, generator=g) * (1.0 / (c * r * r) ** 0.5))
    x16, w16 = to_bf16_nhwc(x), to_bf16_nhwc(wt)
    p, q = ops.out_hw(h, w, r, r, stride, pad[0], pad[1])
    y = ops.empty_nhwc(n, k, p, q, dev)
    slab = torch.empty(_C.fn("rr_conv16_stat_slab_bytes")(n, p, q, k) // 8, dtype=torch.float64, device=dev)
    f = _C.fn("rr_conv16_fprop")

    def fprop(with_stats=True):
        _C.check(f(_C.ptr(x16), _C.ptr(w16), None, _C.ptr(y), None, _C.ptr(slab if with_stats else None), n, h, w, c, k, r, r, stride,
                   pad[0], pad[1], 0, _C.stream()), "rr_conv16_fprop")
    out = {"shape": [n, c, h, w, k, r, stride]}
    flops = 2.0 * n * p * q * k * c * r * r
    if check:
        fprop()
        xr, wr = x16.float(), w16.float()
        ref, rslab = ops.conv_fprop(xr, wr, None, stride, pad, False, want_stats=True)
        scale = float(ref.abs().max())
        out["fprop_err"] = float((y - ref).abs().max()) / scale
        sums = ops.bn_reduce_slab(slab, k)
        rsums = ops.bn_reduce_slab(rslab, k)
        out["stats_err"] = float(((sums - rsums).abs() / rsums.abs().clamp_min(1e-3 * float(rsums.abs().max()))).max())
        # bf16 output + bias + relu
        bias = torch.randn(k, device=dev, generator=g)
        y16 = torch.empty((n, p, q, k), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2)
        _C.check(f(_C.ptr(x16), _C.ptr(w16), _C.ptr(bias), _C.ptr(y), _C.ptr(y16), None, n, h, w, c, k, r, r, stride, pad[0], pad[1], 1,
                   _C.stream()), "rr_conv16_fprop")
        ref2 = ops.conv_fprop(xr, wr, bias, stride, pad, True)
        out["bias_relu_err"] = float((y - ref2).abs().max()) / float(ref2.abs().max())
        out["bf16_out_err"] = float((y16.float() - y).abs().max() / y.abs().max())       # <= 2^-9
        if stride == 1:
            # data gradient: dy [n,k,p,q] -> dx [n,c,h,w] through the flipped / transposed filter
            dy = ops.to_nhwc(torch.randn((n, k, p, q), device=dev, generator=g))
            dy16 = to_bf16_nhwc(dy)
            wflip = torch.empty(k * c * r * r, dtype=torch.float32, device=dev)
            _C.check(_C.fn("rr_weight_flip_transpose")(_C.ptr(wr), _C.ptr(wflip), k, c, r, r, _C.stream()), "flip")
            wflip16 = wflip.to(torch.bfloat16)
            if ops.__dict__.get("_dummy") is None and _C.fn("rr_conv16_supported")(k, c, r, r, 1):
                dx = ops.empty_nhwc(n, c, h, w, dev)
                base = torch.randn_like(dx)
                dx.copy_(base)
                fd = _C.fn("rr_conv16_dgrad_s1")
                _C.check(fd(_C.ptr(dy16), _C.ptr(wflip16), _C.ptr(dx), None, n, h, w, c, k, r, r, pad[0], pad[1], 1, _C.stream()), "dgrad")
                refd = ops.conv_dgrad(dy16.float(), wr, (n, c, h, w), 1, pad) + base
                out["dgrad_acc_err"] = float((dx - refd).abs().max()) / float(refd.abs().max())
                ms = timed(lambda: _C.check(fd(_C.ptr(dy16), _C.ptr(wflip16), _C.ptr(dx), None, n, h, w, c, k, r, r, pad[0], pad[1], 0,
                                               _C.stream()), "dgrad"), reps)
                out["dgrad_ms"], out["dgrad_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
    if _C.fn("rr_conv16_wgrad_supported")(c, k, r, r, stride):
        dy = ops.to_nhwc(torch.randn((n, k, p, q), device=dev, generator=g))
        dy16 = to_bf16_nhwc(dy)
        dw = ops.zeros_nhwc(k, c, r, r, dev)
        fw = _C.fn("rr_conv16_wgrad")

        def wgrad():
            _C.check(fw(_C.ptr(x16), _C.ptr(dy16), _C.ptr(dw), n, h, w, c, k, r, r, stride, pad[0], pad[1], _C.stream()), "rr_conv16_wgrad")
        if check:
            wgrad()
            refw = ops.conv_wgrad(x16.float(), dy16.float(), ops.zeros_nhwc(k, c, r, r, dev), stride, pad)
            out["wgrad_err"] = float((dw - refw).abs().max()) / float(refw.abs().max())
        ms = timed(wgrad, reps)
        out["wgrad_ms"], out["wgrad_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
    ms = timed(lambda: fprop(True), reps)
    out["fprop_ms"], out["fprop_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
    ms = timed(lambda: fprop(False), reps)
    out["fprop_nostats_ms"], out["fprop_nostats_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default=None)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    shapes = [tuple(int(v) for v in a.shape.split(","))] if a.shape else [
        (2, 64, 24, 40, 256, 3, 1), (1, 128, 33, 47, 256, 3, 2), (2, 256, 32, 32, 256, 1, 1),
        (2, 128, 20, 36, 384, 3, 1), (8, 256, 256, 256, 256, 3, 1), (8, 256, 128, 128, 256, 3, 1), (8, 128, 512, 512, 256, 3, 2), (8, 384, 64, 64, 384, 3, 1),
        (8, 256, 128, 128, 384, 3, 2), (8, 384, 32, 32, 384, 3, 1)]
    for s in shapes:
        print(json.dumps(run_shape(*s, reps=a.reps)), flush=True)


if __name__ == "__main__":
    main()
