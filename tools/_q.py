import os, sys, torch
sys.path.insert(0, os.getcwd())
from rrnet_amd import ops
torch.manual_seed(0)
n, c, h, w, k = 8, 256, 256, 256, 256
dev = "cuda"
x = ops.to_nhwc(torch.randn(n, c, h, w, device=dev))
off = ops.to_nhwc(torch.randn(n, 18, h, w, device=dev))
mask = ops.to_nhwc(torch.rand(n, 9, h, w, device=dev))
wt = ops.to_nhwc(torch.randn(k, c, 3, 3, device=dev) * 0.02)
dy = ops.to_nhwc(torch.randn(n, k, h, w, device=dev))
for _ in range(3): ops.dcn_dgrad(x, off, mask, wt, dy, 1, (1, 1), 1, 1, bf16=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.dcn_dgrad(x, off, mask, wt, dy, 1, (1, 1), 1, 1, bf16=True)
e1.record(); torch.cuda.synchronize()
print("RR_DCN_EXP=%s  %.3f ms per dgrad call (incl. memset, dy->bf16, pack)" % (os.environ.get("RR_DCN_EXP"), e0.elapsed_time(e1) / 10))
