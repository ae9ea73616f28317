"""fprop + dgrad + wgrad of the dominant layer (256 -> 256 3x3 on 8 x 256 x 256) on the 16-bit-activation kernels (csrc/conv16.hip),
a few times: the program the PMC passes of tools/pmc_conv16.sh profile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rrnet_amd import ops
torch.manual_seed(0)
n, c, h, k, r = 8, 256, 256, 256, 3
x = ops.to_nhwc(torch.randn(n, c, h, h, device="cuda").relu_())
w = ops.to_nhwc(torch.randn(k, c, r, r, device="cuda") * 0.02)
dy = ops.to_nhwc(torch.randn(n, k, h, h, device="cuda") * 1e-3)
dw = torch.zeros((k, r, r, c), device="cuda").permute(0, 3, 1, 2)
ops.BF16 = ops.MATH_BF16
ops.bf16_of(x); ops.bf16_of(dy); ops.bf16_of(w)          # the images the producers would have left
for _ in range(3):
    ops.conv_fprop(x, w, None, 1, (1, 1), False, want_stats=True)
    ops.conv_dgrad(dy, w, (n, c, h, h), 1, (1, 1))
    ops.conv_wgrad(x, dy, dw, 1, (1, 1))
torch.cuda.synchronize()
