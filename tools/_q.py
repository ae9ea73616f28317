import os, sys, torch
sys.path.insert(0, os.getcwd())
from rrnet_amd import ops, _C
torch.manual_seed(0)
n, c, h, w, k = 8, 256, 256, 256, 28
x = ops.to_nhwc(torch.randn(n, c, h, w, device="cuda"))
dy = ops.to_nhwc(torch.randn(n, k, h, w, device="cuda"))
def run(name):
    dw = ops.zeros_nhwc(k, c, 3, 3, device="cuda")
    f = _C.fn(name)
    def call():
        _C.check(f(_C.ptr(x), _C.ptr(dy), _C.ptr(dw), n, h, w, c, k, 3, 3, 1, 1, 1, h, w, _C.stream()), name)
    call(); torch.cuda.synchronize()
    ref = dw.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    return ref, e0.elapsed_time(e1) / 10
import inspect
print(inspect.signature(ops.conv_wgrad))
a, ta = run("rr_conv_wgrad")
b, tb = run("rr_conv_wgrad_bf16")
print("fp32 %.3f ms, bf16 %.3f ms, rel diff %.3e" % (ta, tb, float((a - b).abs().max() / a.abs().max())))
