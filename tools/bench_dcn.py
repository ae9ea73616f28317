"""DCNv2 forward micro-benchmark (BASELINE config 4 shape: 256->256 3x3 deformable head conv on the 256x256 stride-4 map,
B=8): fp32 MFMA operands vs bf16 operands, TFLOP/s on the 2*M*K*9*C algorithmic FLOPs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rrnet_amd import ops  # noqa: E402

n, c, h, w, k = 8, 256, 256, 256, 256
g = torch.Generator(device="cuda").manual_seed(0)
x = ops.to_nhwc(torch.randn(n, c, h, w, device="cuda", generator=g))
off = ops.to_nhwc(torch.randn(n, 18, h, w, device="cuda", generator=g))
mask = ops.to_nhwc(torch.sigmoid(torch.randn(n, 9, h, w, device="cuda", generator=g)))
wt = ops.to_nhwc(torch.randn(k, c, 3, 3, device="cuda", generator=g) / 48.0)
flops = 2.0 * n * h * w * k * c * 9
for bf in (False, True):
    for _ in range(2):
        ops.dcn_fwd(x, off, mask, wt, None, 1, (1, 1), 1, 1, bf16=bf)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.dcn_fwd(x, off, mask, wt, None, 1, (1, 1), 1, 1, bf16=bf)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print("dcn fwd %s operands: %.3f ms  %.1f TFLOP/s" % ("bf16" if bf else "fp32", ms, flops / ms / 1e9))

# backward (fp32): columns materialised + conv GEMM kernels + col2im, as assembled by rrnet_amd.functional._DCNv2
from rrnet_amd.functional import dcn_v2_conv  # noqa: E402
xg = x.clone().requires_grad_()
og = off.clone().requires_grad_()
mg = mask.clone().requires_grad_()
wg = wt.clone().requires_grad_()
y = dcn_v2_conv(xg, og, mg, wg, None, 1, 1, 1, 1)
gy = torch.randn_like(y)
y.backward(gy)
torch.cuda.synchronize()
for t in (xg, og, mg, wg):
    t.grad = None
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
y = dcn_v2_conv(xg, og, mg, wg, None, 1, 1, 1, 1)
s.record()
y.backward(gy)
e.record()
torch.cuda.synchronize()
print("dcn bwd fp32 (im2col + wgrad + dgrad + col2im): %.2f ms" % s.elapsed_time(e))
