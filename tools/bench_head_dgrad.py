"""Times the heads' narrow 1x1 data gradient (+ ReLU mask + bias sums) at the bench size: rr_head_dgrad_relubias against the
round-3 implicit-GEMM path (rr_conv_dgrad_s1_relubias on zero-padded channels)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rrnet_amd import ops

n, c, h, w = 8, 256, 256, 256
g = torch.Generator(device="cuda").manual_seed(1)
z = ops.to_nhwc(torch.randn((n, c, h, w), device="cuda", generator=g).relu_())
for k in (10, 2, 34):
    dy = ops.to_nhwc(torch.randn((n, k, h, w), device="cuda", generator=g))
    wt = ops.to_nhwc(torch.randn((k, c, 1, 1), device="cuda", generator=g))
    for mode in (True, False):
        ops._HEAD_DGRAD = mode

        def run():
            link = ops.BnLink()
            link.relu_bias = link.use_z = True
            return ops.conv_dgrad(dy, wt, (n, c, h, w), 1, (0, 0), bnsum=link, bnsum_z=z)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            run()
        e.record()
        torch.cuda.synchronize()
        print("K=%2d %-28s %.3f ms" % (k, "rr_head_dgrad_relubias" if mode else "implicit GEMM (padded)", s.elapsed_time(e) / 10))
