"""Drop-in for the reference's ext/dcn/dcn_v2.py: `dcn_v2_conv`, `DCNv2`, `DCN` (:16-128) on the HIP
gather-GEMM (librrnet_hip.so: rr_dcn_fwd / rr_dcn_im2col / rr_dcn_col2im) instead of the `_ext` CUDA
extension.  Same constructor arguments, parameter names (`weight`, `bias`, `conv_offset_mask.*`) and
initialisation (uniform(-1/sqrt(n), 1/sqrt(n)) weights, zero bias, zero-initialised offset/mask conv).
`dcn_v2_pooling`, `DCNv2Pooling`, `DCNPooling` (:130-300, deformable PS-RoI pooling; no caller anywhere in the
reference) on rr_dcn_psroi_fwd / _bwd."""
import math
import os

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from rrnet_amd import functional as RF

dcn_v2_conv = RF.dcn_v2_conv
dcn_v2_pooling = RF.dcn_v2_pooling


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


_PAD_OFFSET_CONV = True     # A/B switch


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
        # `input` has two consumers here (the offset/mask conv and the deformable conv): views of a shared fan-out — both add
        # their input gradient into ONE buffer inside their kernels (the convolution in its epilogue, the deformable data
        # gradient through the atomics it scatters with anyway); an outer fan-out's accumulator on `input` is joined
        xa, xb, _acc = RF.fanout_shared(input, 2)
        # the offset / mask conv runs on the MFMA conv kernels.  Its 3 * dg * kh * kw output channels (27 for a 3x3) are
        # not a multiple of 4, which would put its two gradients on the scalar-gather kernels (28 ms per config-4 step
        # for the six head layers): the filter bank is zero-padded to the next multiple of 4 and the extra channel
        # dropped — same values, vector (and, under cfg.Model.bf16, bf16-operand) kernels forward and backward
        com = self.conv_offset_mask
        k = com.weight.shape[0]
        kp = (k + 3) // 4 * 4
        if kp != k and _PAD_OFFSET_CONV:
            wp = torch.cat((com.weight, com.weight.new_zeros((kp - k,) + tuple(com.weight.shape[1:]))), 0)
            bp = torch.cat((com.bias, com.bias.new_zeros(kp - k))) if com.bias is not None else None
            out = RF.conv_weight(xa, wp, bp, com.stride[0], tuple(com.padding))[:, :k]
        else:
            out = RF.conv_bias(xa, com)
        offset, mask = RF.dcn_offset_mask(out)                    # chunk(3) + cat(o1, o2) + sigmoid(mask): one kernel
        return dcn_v2_conv(xb, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups, bf16=self.bf16)


class DCNv2Pooling(nn.Module):
    """ext/dcn/dcn_v2.py:185-219."""

    def __init__(self, spatial_scale, pooled_size, output_dim, no_trans, group_size=1, part_size=None, sample_per_part=4,
                 trans_std=.0):
        super().__init__()
        self.spatial_scale = spatial_scale
        self.pooled_size = pooled_size
        self.output_dim = output_dim
        self.no_trans = no_trans
        self.group_size = group_size
        self.part_size = pooled_size if part_size is None else part_size
        self.sample_per_part = sample_per_part
        self.trans_std = trans_std

    def forward(self, input, rois, offset):
        assert input.shape[1] == self.output_dim
        if self.no_trans:
            offset = input.new()
        return dcn_v2_pooling(input, rois, offset, self.spatial_scale, self.pooled_size, self.output_dim, self.no_trans,
                              self.group_size, self.part_size, self.sample_per_part, self.trans_std)


class DCNPooling(DCNv2Pooling):
    """ext/dcn/dcn_v2.py:222-300: plain pooling -> fully connected offset / mask predictor (zero-initialised last
    layer) -> deformable pooling * sigmoid(mask).  The three small Linear layers are library GEMMs (torch), as in the
    reference."""

    def __init__(self, spatial_scale, pooled_size, output_dim, no_trans, group_size=1, part_size=None, sample_per_part=4,
                 trans_std=.0, deform_fc_dim=1024):
        super().__init__(spatial_scale, pooled_size, output_dim, no_trans, group_size, part_size, sample_per_part, trans_std)
        self.deform_fc_dim = deform_fc_dim
        if not no_trans:
            self.offset_mask_fc = nn.Sequential(
                nn.Linear(self.pooled_size * self.pooled_size * self.output_dim, self.deform_fc_dim), nn.ReLU(inplace=True),
                nn.Linear(self.deform_fc_dim, self.deform_fc_dim), nn.ReLU(inplace=True),
                nn.Linear(self.deform_fc_dim, self.pooled_size * self.pooled_size * 3))
            self.offset_mask_fc[4].weight.data.zero_()
            self.offset_mask_fc[4].bias.data.zero_()

    def forward(self, input, rois):
        offset = input.new()
        args = (self.spatial_scale, self.pooled_size, self.output_dim)
        tail = (self.group_size, self.part_size, self.sample_per_part, self.trans_std)
        if not self.no_trans:
            n = rois.shape[0]
            roi = dcn_v2_pooling(input, rois, offset, *args, True, *tail)
            offset_mask = self.offset_mask_fc(roi.reshape(n, -1))       # logical (c, ph, pw) order, as the reference's view
            offset_mask = offset_mask.view(n, 3, self.pooled_size, self.pooled_size)
            o1, o2, mask = torch.chunk(offset_mask, 3, dim=1)
            offset = torch.cat((o1, o2), dim=1)
            mask = torch.sigmoid(mask)
            return dcn_v2_pooling(input, rois, offset, *args, self.no_trans, *tail) * mask
        return dcn_v2_pooling(input, rois, offset, *args, self.no_trans, *tail)
