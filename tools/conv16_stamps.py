"""In-kernel timeline of the conv16 ping-pong loop (csrc/conv16.hip, DBG instantiation): six s_memtime stamps per sub-phase and wave of
one workgroup in the middle of the grid -> where a sub-phase's cycles go (load segment issue, its waits, barrier 1, the 32 MFMAs, barrier 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rrnet_amd import _C, ops

dev = torch.device("cuda")
n, c, h, w, k = 8, 256, 256, 256, 256
g = torch.Generator(device=dev).manual_seed(1)
x16 = ops.to_nhwc(torch.randn((n, c, h, w), device=dev, generator=g).relu_()).to(torch.bfloat16)
w16 = ops.to_nhwc(torch.randn((k, c, 3, 3), device=dev, generator=g) / 48.0).to(torch.bfloat16)
y = ops.empty_nhwc(n, k, h, w, dev)
buf = torch.zeros(8 * 72 * 6, dtype=torch.int64, device=dev)
f = _C.fn("rr_conv16_fprop")
def run():
    _C.check(f(_C.ptr(x16), _C.ptr(w16), None, _C.ptr(y), None, None, n, h, w, c, k, 3, 3, 1, 1, 1, 0, _C.stream()), "fprop")
for _ in range(20):
    run()
_C.fn("rr_conv16_debug_stamps")(_C.ptr(buf))
for _ in range(3):
    run()
torch.cuda.synchronize()
_C.fn("rr_conv16_debug_stamps")(None)
t = buf.cpu().numpy().reshape(8, 72, 6).astype(np.int64)
names = ["load issue (12 ds_read + 4 DMA)", "wait vmcnt(4) lgkmcnt(0)", "barrier 1", "32 MFMA", "barrier 2", "loop back -> next"]
for wv in (0, 1, 4, 5):
    seg = np.diff(t[wv, 8:64, :], axis=1)                      # [s][5]
    nxt = t[wv, 9:65, 0] - t[wv, 8:64, 5]
    per = t[wv, 9:65, 0] - t[wv, 8:64, 0]
    print("wave %d: sub-phase period median %d cycles (min %d max %d)" % (wv, np.median(per), per.min(), per.max()))
    for i in range(5):
        print("    %-34s median %5d  p10 %5d  p90 %5d" % (names[i], np.median(seg[:, i]), np.percentile(seg[:, i], 10), np.percentile(seg[:, i], 90)))
    print("    %-34s median %5d" % (names[5], np.median(nxt)))
print("total loop cycles wave0:", t[0, 71, 5] - t[0, 0, 0], " = per sub-phase", (t[0, 71, 5] - t[0, 0, 0]) / 72.0)
print("stagger: wave4 start - wave0 start of sub-phase 20:", t[4, 20, 0] - t[0, 20, 0], "; wave0 MFMA start", t[0, 20, 3] - t[0, 20, 0], " wave4 MFMA start", t[4, 20, 3] - t[0, 20, 0])
