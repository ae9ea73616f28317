"""Drop-in for the reference's ext/dcn/dcn_v2.py: `dcn_v2_conv`, `DCNv2`, `DCN` (:16-128) on the HIP
gather-GEMM (librrnet_hip.so: rr_dcn_fwd / rr_dcn_im2col / rr_dcn_col2im) instead of the `_ext` CUDA
extension.  Same constructor arguments, parameter names (`weight`, `bias`, `conv_offset_mask.*`) and
initialisation (uniform(-1/sqrt(n), 1/sqrt(n)) weights, zero bias, zero-initialised offset/mask conv).
DCNPooling / deformable PS-RoI pooling have no caller anywhere in the reference and are out of scope."""
import math

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from rrnet_amd import functional as RF

dcn_v2_conv = RF.dcn_v2_conv


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.deformable_groups = deformable_groups
        self.bf16 = None      # extension: True/False selects bf16 / fp32 matrix operands; None = RR_DCN_BF16
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        self.bias.data.zero_()

    def forward(self, input, offset, mask):
        assert 2 * self.deformable_groups * self.kernel_size[0] * self.kernel_size[1] == offset.shape[1]
        assert self.deformable_groups * self.kernel_size[0] * self.kernel_size[1] == mask.shape[1]
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups, bf16=self.bf16)


class DCN(DCNv2):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        channels_ = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(self.in_channels, channels_, kernel_size=self.kernel_size,
                                          stride=self.stride, padding=self.padding, bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, input):
        # `input` has two consumers here (the offset/mask conv and the deformable conv): plain fan-out views, so that a
        # shared gradient accumulator tagged on `input` by an outer fan-out is not mistaken for a single-consumer tag
        xa, xb = RF.fanout(input, 2)
        out = RF.conv_bias(xa, self.conv_offset_mask)             # the offset / mask conv runs on the MFMA conv kernels
        offset, mask = RF.dcn_offset_mask(out)                    # chunk(3) + cat(o1, o2) + sigmoid(mask): one kernel
        return dcn_v2_conv(xb, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups, bf16=self.bf16)
