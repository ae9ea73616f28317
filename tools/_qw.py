import os, sys, torch
sys.path.insert(0, os.getcwd())
from rrnet_amd import ops
torch.manual_seed(0)
n, c, h, w, k = 8, 256, 256, 256, 256
dev = "cuda"
x = ops.to_nhwc(torch.randn(n, c, h, w, device=dev))
off = ops.to_nhwc(torch.randn(n, 18, h, w, device=dev))
mask = ops.to_nhwc(torch.rand(n, 9, h, w, device=dev))
dw = ops.to_nhwc(torch.zeros(k, c, 3, 3, device=dev))
dy = ops.to_nhwc(torch.randn(n, k, h, w, device=dev))
IMG = ops.bf16_of(dy) if os.environ.get("IMG", "1") == "1" else None
for _ in range(3): ops.dcn_wgrad(x, off, mask, dy, dw, 1, (1, 1), 1, 1, bf16=True, dy_img=IMG)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.dcn_wgrad(x, off, mask, dy, dw, 1, (1, 1), 1, 1, bf16=True, dy_img=IMG)
e1.record(); torch.cuda.synchronize()
print("RR_DCN_EXPW=%s  %.3f ms per wgrad call" % (os.environ.get("RR_DCN_EXPW"), e0.elapsed_time(e1) / 10))
