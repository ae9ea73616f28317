"""Stage-2 head kernels at config-5 size (192 000 RoIs of 3x3x256 per 128 frames): conv1 1x1 256->64, conv2 3x3 64->64 (padding
taps skipped), fused tail (conv3 + bn3 + residual + ReLU + average pool), regressor — ms per launch, one JSON line.
  python tools/bench_head.py [--rois 192000] [--iters 20]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rois", type=int, default=192000)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from rrnet_amd import ops
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    r = a.rois
    x = ops.to_nhwc(torch.randn(r, 256, 3, 3, device=dev))
    w1 = ops.to_nhwc(torch.randn(64, 256, 1, 1, device=dev) / 16)
    w2 = ops.to_nhwc(torch.randn(64, 64, 3, 3, device=dev) / 24)
    w3 = ops.to_nhwc(torch.randn(256, 64, 1, 1, device=dev) / 8)
    b = torch.randn(64, device=dev) * 0.1
    sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1

    def t(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / a.iters, 4), out

    t1, h1 = t(lambda: ops.conv_fprop(x, w1, b, 1, (0, 0), True))
    t2, h2 = t(lambda: ops.conv_fprop(h1, w2, b, 1, (1, 1), True))
    t3, f = t(lambda: ops.conv1x1_bn_res_relu_avgpool(h2, w3, sc, sh, x))
    gb = lambda *ts: sum(q.numel() * 4 for q in ts) / 1e9
    print(json.dumps({"rois": r, "conv1_ms": t1, "conv2_ms": t2, "tail_ms": t3,
                      "conv1_tbps": round(gb(x, h1) / t1, 2), "conv2_tflops": round(r * 49 * 64 * 64 * 2 / t2 / 1e9, 1),
                      "tail_tbps": round(gb(x, h2, f) / t3, 2), "tail_tflops": round(r * 9 * 64 * 256 * 2 / t3 / 1e9, 1)}))


if __name__ == "__main__":
    main()
