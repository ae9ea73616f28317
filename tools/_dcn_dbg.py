import os, sys, torch
sys.path.insert(0, os.getcwd())
from rrnet_amd import ops
n, c, h, w, k = 8, 256, 256, 256, 256
g = torch.Generator(device="cuda").manual_seed(0)
x = ops.to_nhwc(torch.randn(n, c, h, w, device="cuda", generator=g))
off = ops.to_nhwc(torch.randn(n, 18, h, w, device="cuda", generator=g))
mask = ops.to_nhwc(torch.sigmoid(torch.randn(n, 9, h, w, device="cuda", generator=g)))
wt = ops.to_nhwc(torch.randn(k, c, 3, 3, device="cuda", generator=g) / 48.0)
dy = ops.to_nhwc(torch.randn(n, k, h, w, device="cuda", generator=g))
for _ in range(2): ops.dcn_dgrad(x, off, mask, wt, dy, 1, (1, 1), 1, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3): ops.dcn_dgrad(x, off, mask, wt, dy, 1, (1, 1), 1, 1)
e.record(); torch.cuda.synchronize()
print("RR_DCN_DBG=%s dgrad %.2f ms" % (os.environ.get("RR_DCN_DBG"), s.elapsed_time(e) / 3))
