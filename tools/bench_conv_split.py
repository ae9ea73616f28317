"""The three arithmetics of the convolution kernels side by side at the headline layer shapes: fp32 MFMA (csrc/conv.hip),
bf16 operands, split fp16 operands ("f16x3", csrc/conv_bf16.hip) — time of fprop / dgrad / wgrad and the forward's error
against an fp64 convolution.

  python tools/bench_conv_split.py [shape index ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rrnet_amd import ops  # noqa: E402

MODES = (("f32", ops.MATH_F32), ("bf16", ops.MATH_BF16), ("f16x3", ops.MATH_F16X3))
SHAPES = [(8, 256, 256, 256, 3), (8, 256, 128, 256, 3), (8, 384, 64, 384, 3), (8, 384, 32, 384, 3), (8, 384, 16, 384, 3),
          (8, 512, 8, 512, 3), (8, 256, 256, 256, 1)]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def accuracy(verbose=True):
    torch.manual_seed(0)
    n, c, h, k, r = 2, 256, 128, 256, 3
    x = torch.randn(n, c, h, h, device="cuda") * torch.rand(n, c, 1, 1, device="cuda") * 3
    w = torch.randn(k, c, r, r, device="cuda") * 0.05
    cols = torch.nn.functional.unfold(x.double(), r, padding=1)
    ref = (w.double().reshape(k, -1) @ cols).reshape(n, k, h, h)
    rms = ref.pow(2).mean().sqrt()
    out = {}
    saved = ops.BF16
    for name, mode in MODES:
        ops.BF16 = mode
        y = ops.conv_fprop(ops.to_nhwc(x), ops.to_nhwc(w), None, 1, (1, 1), False).double()
        out[name] = {"rms_err_over_rms_y": float((y - ref).pow(2).mean().sqrt() / rms),
                     "max_err_over_max_y": float((y - ref).abs().max() / ref.abs().max())}
        if verbose:
            print("%-6s forward error vs fp64 (n2 c256 128x128 k256 3x3): rms %.3e of rms y, max %.3e of max |y|" % (
                name, out[name]["rms_err_over_rms_y"], out[name]["max_err_over_max_y"]))
    ops.BF16 = saved
    return out


def layer_rates(shape=(8, 256, 256, 256, 3), iters=10):
    """fprop / dgrad / wgrad of one layer shape under the three arithmetics -> {math: {op: {"ms", "tflops"}}} (the operand
    maxima of the split kernels are reduced inside the timed calls, as in a train step without a fused producer)."""
    n, c, h, k, r = shape
    pad = r // 2
    x = ops.to_nhwc(torch.randn(n, c, h, h, device="cuda"))
    w = ops.to_nhwc(torch.randn(k, c, r, r, device="cuda") * 0.02)
    dy = ops.to_nhwc(torch.randn(n, k, h, h, device="cuda") * 1e-3)
    dw = torch.zeros((k, r, r, c), device="cuda").permute(0, 3, 1, 2)
    flops = 2.0 * n * h * h * k * c * r * r
    saved, out = ops.BF16, {}
    for name, mode in MODES:
        ops.BF16 = mode
        t = {"fprop": timeit(lambda: ops.conv_fprop(x.view_as(x), w.view_as(w), None, 1, (pad, pad), False), iters),
             "dgrad": timeit(lambda: ops.conv_dgrad(dy.view_as(dy), w.view_as(w), (n, c, h, h), 1, (pad, pad)), iters),
             "wgrad": timeit(lambda: ops.conv_wgrad(x.view_as(x), dy.view_as(dy), dw, 1, (pad, pad)), iters)}
        out[name] = {kk: {"ms": round(v, 4), "tflops": round(flops / v / 1e9, 1)} for kk, v in t.items()}
    ops.BF16 = saved
    return out


def main():
    accuracy()
    only = [int(a) for a in sys.argv[1:]]
    print("%-28s %-6s %9s %9s %9s   (ms | TFLOP/s)" % ("shape", "math", "fprop", "dgrad", "wgrad"))
    for i, (n, c, h, k, r) in enumerate(SHAPES):
        if only and i not in only:
            continue
        pad = r // 2
        x = ops.to_nhwc(torch.randn(n, c, h, h, device="cuda"))
        w = ops.to_nhwc(torch.randn(k, c, r, r, device="cuda") * 0.02)
        dy = ops.to_nhwc(torch.randn(n, k, h, h, device="cuda") * 1e-3)
        dw = torch.zeros((k, r, r, c), device="cuda").permute(0, 3, 1, 2)
        flops = 2.0 * n * h * h * k * c * r * r
        for name, mode in MODES:
            ops.BF16 = mode
            # a fresh Python object per call: the operands' maxima are reduced every time, as in a train step
            t = [timeit(lambda: ops.conv_fprop(x.view_as(x), w.view_as(w), None, 1, (pad, pad), False)),
                 timeit(lambda: ops.conv_dgrad(dy.view_as(dy), w.view_as(w), (n, c, h, h), 1, (pad, pad))),
                 timeit(lambda: ops.conv_wgrad(x.view_as(x), dy.view_as(dy), dw, 1, (pad, pad)))]
            print("N%d C%d %dx%d K%d %dx%d %-4s %-6s " % (n, c, h, h, k, r, r, "", name) +
                  " ".join("%5.3f|%5.1f" % (v, flops / v / 1e9) for v in t))
        ops.BF16 = ops.MATH_F32


if __name__ == "__main__":
    main()
