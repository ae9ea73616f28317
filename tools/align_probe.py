"""Does the distance between the base addresses of the streams of an HBM-bound kernel matter?  bn_apply (read y, write out) on two
8 x 256 x 256 x 256 fp32 tensors (exactly 512 MiB each) whose bases differ by 512 MiB + d for several d.
  python tools/align_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rrnet_amd import ops, _C

n, c, h, w = 8, 256, 256, 256
numel = n * c * h * w
pool = torch.empty(3 * numel + (64 << 20), dtype=torch.float32, device="cuda")      # one slab: addresses under our control
scale = torch.ones(c, device="cuda"); shift = torch.zeros(c, device="cuda")
base = pool.data_ptr()
print("slab base %#x (mod 2 MiB: %#x)" % (base, base % (2 << 20)))
for d_bytes in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 6 << 20, (6 << 20) + 4096, 32 << 20):
    d = d_bytes // 4
    y = pool[:numel]
    out = pool[numel + d: 2 * numel + d]
    y.normal_()
    f = _C.fn("rr_bn_apply")
    def run():
        _C.check(f(_C.ptr(y), _C.ptr(scale), _C.ptr(shift), None, None, None, _C.ptr(out), numel, c, 1, _C.stream()), "bn")
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("out - y = 512 MiB + %9d B: %.3f ms  %.2f TB/s" % (d_bytes, ms, 2 * numel * 4 / ms / 1e9))
