"""CenterNet heads on the MI355X kernels — module tree of the reference's
detectors/centernet_detector.py (CenterNetDetector :6-23, CenterNetWHDetector :26-55,
HCov :58-67, WCov :69-77, BasicCov :80-93); bias and ReLU run in the conv epilogue."""
import torch
import torch.nn as nn

from rrnet_amd import functional as RF


class BasicCov(nn.Module):
    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        pad = (k - 1) // 2
        self.conv = nn.Conv2d(inp_dim, out_dim, (k, k), padding=(pad, pad), stride=(stride, stride), bias=not with_bn)
        self.bn = nn.BatchNorm2d(out_dim) if with_bn else nn.Sequential()
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        if isinstance(self.bn, nn.Sequential):
            return RF.conv_bias(x, self.conv, relu=True)
        return RF.conv_bn_act(x, self.conv, self.bn, relu=True)


class DCNCov(nn.Module):
    """Builder-defined (BASELINE configs[3], "RRNet + ext/dcn deformable-conv heads"; the reference wires DCN into no
    RRNet head): BasicCov(3, ., ., with_bn=False) with its 3x3 convolution replaced by ext/dcn's `DCN` (modulated
    deformable conv whose offsets / masks come from a zero-initialised 3x3 conv), then ReLU.  `bf16`: bf16 matrix
    operands in the deformable conv's forward (fp32 accumulation)."""

    def __init__(self, inp_dim, out_dim, bf16=False):
        super().__init__()
        from rrnet_amd.ext.dcn.dcn_v2 import DCN
        self.conv = DCN(inp_dim, out_dim, 3, stride=1, padding=1)
        self.conv.bf16 = bool(bf16)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return RF.relu(self.conv(x))


class HCov(nn.Module):
    """k x 1 convolution (a column of k taps)."""

    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        self.conv = nn.Conv2d(inp_dim, out_dim, (k, 1), padding=((k - 1) // 2, 0), stride=(stride, stride), bias=not with_bn)

    def forward(self, x):
        return RF.conv_bias(x, self.conv)


class WCov(nn.Module):
    """1 x k convolution (a row of k taps)."""

    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        self.conv = nn.Conv2d(inp_dim, out_dim, (1, k), padding=(0, (k - 1) // 2), stride=(stride, stride), bias=not with_bn)

    def forward(self, x):
        return RF.conv_bias(x, self.conv)


def _head_conv(dcn, bf16):
    return DCNCov(256, 256, bf16=bf16) if dcn else BasicCov(3, 256, 256, with_bn=False)


class CenterNetDetector(nn.Module):
    def __init__(self, planes, hm=True, num_stacks=2, dcn=False, dcn_bf16=False):
        super().__init__()
        self.hm = hm
        self.num_stacks = num_stacks
        self.detect_layer = nn.ModuleList([
            nn.Sequential(_head_conv(dcn, dcn_bf16), nn.Conv2d(256, planes, (1, 1)))
            for _ in range(num_stacks)])
        if self.hm:
            for head in self.detect_layer:
                head[-1].bias.data.fill_(-2.19)

    def forward(self, input, index):
        head = self.detect_layer[index]
        return RF.conv_bias(head[0](input), head[1])


class CenterNetWHDetector(nn.Module):
    def __init__(self, planes, hm=True, num_stacks=2, dcn=False, dcn_bf16=False):
        super().__init__()
        self.hm = hm
        self.num_stacks = num_stacks
        self.detect_conv_layer = nn.ModuleList([nn.Sequential(_head_conv(dcn, dcn_bf16))
                                                for _ in range(num_stacks)])
        self.detect_H_layer = nn.ModuleList([nn.Sequential(HCov(17, 256, planes, with_bn=False))
                                             for _ in range(num_stacks)])
        self.detect_W_layer = nn.ModuleList([nn.Sequential(WCov(17, 256, planes, with_bn=False))
                                             for _ in range(num_stacks)])

    def forward(self, input, index):
        conv = self.detect_conv_layer[index][0](input)
        hc, wc = self.detect_H_layer[index][0].conv, self.detect_W_layer[index][0].conv
        if hc.out_channels == 1 and wc.out_channels == 1 and hc.kernel_size[0] == wc.kernel_size[1]:
            # planes == 1 (the only configuration the reference instantiates): both 17-tap convolutions as
            # ONE 1x1 convolution to 34 per-tap partial products + a shift-sum (see rr_wh_shift_sum_fwd)
            k = hc.kernel_size[0]
            c = hc.in_channels
            ct = (2 * k + 3) // 4 * 4       # pad the tap count to a multiple of 4 (vector dgrad path)
            taps = torch.cat((hc.weight.permute(0, 2, 3, 1).reshape(k, c), wc.weight.permute(0, 2, 3, 1).reshape(k, c),
                              hc.weight.new_zeros(ct - 2 * k, c)))
            t = RF.conv_weight(conv, taps.view(ct, 1, 1, c).permute(0, 3, 1, 2))
            return RF.wh_shift_sum(t, wc.bias, hc.bias, k)
        ca, cb = RF.fanout(conv, 2)
        H = self.detect_H_layer[index][0](ca)
        W = self.detect_W_layer[index][0](cb)
        # channels interleaved [W0, H0, W1, H1, ...] exactly as centernet_detector.py:50-53
        H = H.reshape(H.size(0), -1, 1, H.size(2), H.size(3))
        W = W.reshape(W.size(0), -1, 1, W.size(2), W.size(3))
        out = torch.cat((W, H), dim=2).reshape(H.size(0), -1, H.size(3), H.size(4))
        return out.contiguous(memory_format=torch.channels_last)
