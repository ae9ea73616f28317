"""Micro-benchmark of the bf16-operand conv kernels (csrc/conv_bf16.hip) beside the fp32 ones at the config-2 layer shapes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rrnet_amd import ops


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


SHAPES = [  # N, C, H, W, K, R, stride
    (8, 256, 256, 256, 256, 3, 1),
    (8, 256, 128, 128, 256, 3, 1),
    (8, 384, 64, 64, 384, 3, 1),
    (8, 384, 32, 32, 384, 3, 1),
    (8, 384, 16, 16, 384, 3, 1),
    (8, 512, 8, 8, 512, 3, 1),
    (8, 256, 256, 256, 256, 1, 1),
    (8, 128, 512, 512, 256, 3, 2),
    (8, 256, 256, 256, 256, 3, 2),
]
only = [int(a) for a in sys.argv[1:]]
for i, (n, c, h, w, k, r, st) in enumerate(SHAPES):
    if only and i not in only:
        continue
    pad = r // 2
    x = ops.to_nhwc(torch.randn(n, c, h, w, device="cuda"))
    wt = ops.to_nhwc(torch.randn(k, c, r, r, device="cuda") * 0.02)
    p, q = ops.out_hw(h, w, r, r, st, pad, pad)
    dy = ops.to_nhwc(torch.randn(n, k, p, q, device="cuda"))
    dw = torch.zeros((k, r, r, c), device="cuda").permute(0, 3, 1, 2)
    dx = ops.empty_nhwc(n, c, h, w, "cuda")
    flops = 2.0 * n * p * q * k * c * r * r
    row = "N%d C%d %dx%d K%d r%d s%d |" % (n, c, h, w, k, r, st)
    # the filter's bf16 copies as FlatParams keeps them (refreshed once per optimizer step): [k][r][s][c] and flipped [c][r'][s'][k]
    w16 = wt.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
    wt32 = torch.empty(k * c * r * r, dtype=torch.float32, device="cuda")
    ops._C.check(ops._C.fn("rr_weight_flip_transpose")(ops._C.ptr(wt), ops._C.ptr(wt32), k, c, r, r, ops._C.stream()), "flip")
    wt16 = wt32.to(torch.bfloat16)
    for bf in (False, True, "w16"):
        ops.BF16 = bool(bf)
        kw = dict(w16=w16) if bf == "w16" else {}
        kd = dict(wt=wt32, wt16=wt16) if bf == "w16" else {}
        t1 = timeit(lambda: ops.conv_fprop(x, wt, None, st, (pad, pad), False, want_stats=True, **kw))
        t2 = timeit(lambda: ops.conv_dgrad(dy, wt, (n, c, h, w), st, (pad, pad), out=dx, **kd))
        t3 = timeit(lambda: ops.conv_wgrad(x, dy, dw, st, (pad, pad)))
        row += " %s fprop %.3f ms %.0f TF, dgrad %.3f ms %.0f TF, wgrad %.3f ms %.0f TF |" % (
            {False: "fp32", True: "bf16", "w16": "bf16 + bf16 filter copies"}[bf], t1, flops / t1 / 1e9, t2, flops / t2 / 1e9,
            t3, flops / t3 / 1e9)
    ops.BF16 = False
    print(row, flush=True)
