"""BatchNorm backward passes at the largest layer of the hourglass (8 x 256 x 256^2): time and achieved HBM rate of
rr_bn_bwd_reduce / rr_bn_bwd_apply, fp32 tensors and the bf16-only forms of config 4 (z, y as images; dx as an image).
  python tools/bench_bn.py        (GPU box)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rrnet_amd import ops  # noqa: E402


def _t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n, c, h, w = 8, 256, 256, 256
    g = torch.Generator(device="cuda").manual_seed(1)
    y = ops.to_nhwc(torch.randn(n, c, h, w, device="cuda", generator=g))
    z = ops.to_nhwc(torch.relu(y))
    dz = ops.to_nhwc(torch.randn(n, c, h, w, device="cuda", generator=g))
    mean = y.mean((0, 2, 3)).contiguous()
    invstd = (1.0 / y.std((0, 2, 3))).contiguous()
    gamma = torch.ones(c, device="cuda")
    el = y.numel()
    out = {"shape": [n, c, h, w]}
    ms = _t(lambda: ops.bn_bwd_reduce(dz, z, y, mean, invstd))
    out["reduce_fp32"] = {"ms": round(ms, 4), "GBps": round(el * 12 / ms / 1e6, 1), "bytes_per_element": 12}
    sums = ops.bn_bwd_reduce(dz, z, y, mean, invstd)
    ms = _t(lambda: ops.bn_bwd_apply(dz, z, y, mean, invstd, gamma, sums, float(n * h * w)))
    out["apply_fp32"] = {"ms": round(ms, 4), "GBps": round(el * 16 / ms / 1e6, 1), "bytes_per_element": 16}
    with ops.bf16_scope(True):
        zp = ops.phantom_f32((n, c, h, w), y.device, ops.bf16_of(z))
        yp = ops.phantom_f32((n, c, h, w), y.device, ops.bf16_of(y))
        ms = _t(lambda: ops.bn_bwd_reduce(dz, zp, yp, mean, invstd))
        out["reduce_b16"] = {"ms": round(ms, 4), "GBps": round(el * 8 / ms / 1e6, 1), "bytes_per_element": 8}
        ms = _t(lambda: ops.bn_bwd_apply(dz, zp, yp, mean, invstd, gamma, sums, float(n * h * w), bf16_only=True))
        out["apply_b16"] = {"ms": round(ms, 4), "GBps": round(el * 10 / ms / 1e6, 1), "bytes_per_element": 10}
    scale, shift = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    ms = _t(lambda: ops.bn_apply(y, scale, shift, relu=True))
    out["fwd_apply_fp32"] = {"ms": round(ms, 4), "GBps": round(el * 8 / ms / 1e6, 1), "bytes_per_element": 8}
    with ops.bf16_scope(True):
        yp = ops.phantom_f32((n, c, h, w), y.device, ops.bf16_of(y))
        ms = _t(lambda: ops.bn_apply(yp, scale, shift, relu=True, bf16_only=True))
        out["fwd_apply_b16"] = {"ms": round(ms, 4), "GBps": round(el * 4 / ms / 1e6, 1), "bytes_per_element": 4}
        rp = ops.phantom_f32((n, c, h, w), y.device, ops.bf16_of(z))
        ms = _t(lambda: ops.bn_apply(yp, scale, shift, residual=rp, relu=True, bf16_only=True))
        out["fwd_apply_res_b16"] = {"ms": round(ms, 4), "GBps": round(el * 6 / ms / 1e6, 1), "bytes_per_element": 6}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
