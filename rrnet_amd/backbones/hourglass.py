"""Stacked-hourglass backbone on the MI355X kernels.

Same module tree / attribute names / construction order as the reference's
backbones/hourglass.py (ResidualBlock :12-40, ConvBNRelu :43-61, Hourglass :64-124,
HourglassNet :127-199, hourglass_net :202-210), so `state_dict()` keys, default initialisation
under a seed and checkpoints are interchangeable.  nn.Conv2d / nn.BatchNorm2d objects are only
parameter holders here: every forward goes through rrnet_amd.functional (fused conv + BN
statistics + BN apply + ReLU + residual add on NHWC tensors), never through ATen/MIOpen.

Differences that are not behavioural: the reference hard-codes n=5, inplanes=[256,256,384,384,
384,512], layer_nums=[2,2,2,2,2,4], stem 128 and 256 output features in HourglassNet.__init__;
here they are keyword arguments with those defaults so that the builder-defined
"hourglass-tiny" of BASELINE.json config 1 is the same class with smaller numbers.
"""
import os

import torch
import torch.nn as nn

from rrnet_amd import functional as RF
from rrnet_amd import ops

__all__ = ['HourglassNet', 'Hourglass', 'ResidualBlock', 'ConvBNRelu', 'hourglass_net', 'hourglass_tiny']

HG104 = dict(n=5, inplanes=(256, 256, 384, 384, 384, 512), layer_nums=(2, 2, 2, 2, 2, 4), stem=128, num_feats=256)
HG_TINY = dict(n=2, inplanes=(32, 32, 48), layer_nums=(1, 1, 2), stem=16, num_feats=256)


class ResidualBlock(nn.Module):
    expansion = 2

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        projected = stride != 1 or inplanes != planes
        self.skip_connection = nn.Sequential(
            nn.Conv2d(inplanes, planes, (1, 1), stride=stride, bias=False),
            nn.BatchNorm2d(planes)) if projected else nn.Sequential()
        self.stride = stride

    def first_layers(self):
        """The conv -> bn [-> relu] layers that read the block's input: conv1 and, in a projection block, the skip."""
        layers = [(self.conv1, self.bn1, True)]
        if len(self.skip_connection):
            layers.append((self.skip_connection[0], self.skip_connection[1], False))
        return layers

    def tail(self, out, skip):
        """relu(bn2(conv2(out)) + skip): BN apply, residual add and ReLU are one kernel."""
        return RF.conv_bn_act(out, self.conv2, self.bn2, relu=True, residual=skip)

    def forward(self, x):
        # both consumers of x (conv1 and the skip path) accumulate their input gradients into one buffer
        if len(self.skip_connection) and RF.sync_coalescing(self.bn1):
            # SyncBN across ranks: the projection and conv1 read the same x — one joint node, ONE statistics exchange per
            # direction for the pair (conv1 first: its 3x3 data gradient writes the shared buffer, the 1x1 accumulates)
            out, skip = RF.conv_bn_act_multi(x, self.first_layers())
            return self.tail(out, skip)
        xa, xb, _ = RF.fanout_shared(x, 2)
        if len(self.skip_connection):
            # the projection runs BEFORE conv1: autograd then runs conv1's backward first, so the 3x3 data gradient is
            # the one that WRITES the shared buffer and the cheap 1x1 one accumulates into it (a stride-2 3x3 dgrad that
            # has to read-modify-write its strided output is 0.4-0.7 ms slower per launch at the top levels)
            skip = RF.conv_bn_act(xb, self.skip_connection[0], self.skip_connection[1], relu=False)
        else:
            skip = xb                                   # identity skip: the raw view, its tag names the accumulator
        out = RF.conv_bn_act(xa, self.conv1, self.bn1, relu=True)
        # relu(bn2(conv2(out)) + skip): BN apply, residual add and ReLU are one kernel
        return RF.conv_bn_act(out, self.conv2, self.bn2, relu=True, residual=skip)


class ConvBNRelu(nn.Module):
    def __init__(self, kernel_size, inplane, plane, stride=1, with_bn=True, with_relu=True):
        super().__init__()
        pad = (kernel_size - 1) // 2
        self.conv = nn.Conv2d(inplane, plane, (kernel_size, kernel_size), padding=(pad, pad),
                              stride=(stride, stride), bias=not with_bn)
        self.bn = nn.BatchNorm2d(plane) if with_bn else nn.Sequential()
        self.with_relu = with_relu
        if with_relu:
            self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        if isinstance(self.bn, nn.Sequential):
            return RF.conv_bias(x, self.conv, relu=self.with_relu)
        return RF.conv_bn_act(x, self.conv, self.bn, relu=self.with_relu)


def _blocks(cin, cout, count, first_stride=1, widen_last=False):
    """`count` ResidualBlocks; the channel change sits on the first block, or on the last one
    for the up-path (`make_reverse_residual_layer`, hourglass.py:96-102)."""
    if widen_last:
        mods = [ResidualBlock(cin, cin) for _ in range(count - 1)] + [ResidualBlock(cin, cout)]
    else:
        mods = [ResidualBlock(cin, cout, first_stride)] + [ResidualBlock(cout, cout) for _ in range(count - 1)]
    return nn.Sequential(*mods)


class Hourglass(nn.Module):
    def __init__(self, n, inplanes, layer_nums):
        super().__init__()
        self.n = n
        cur, nxt = inplanes[0], inplanes[1]
        cur_n, nxt_n = layer_nums[0], layer_nums[1]
        self.up1 = _blocks(cur, cur, cur_n)
        self.max1 = nn.Sequential()                      # the reference pools with the stride-2 block of low1
        self.low1 = _blocks(cur, nxt, cur_n, first_stride=2)
        self.low2 = Hourglass(n - 1, inplanes[1:], layer_nums[1:]) if n > 1 else _blocks(nxt, nxt, nxt_n)
        self.low3 = _blocks(nxt, cur, cur_n, widen_last=True)
        self.up2 = nn.Upsample(scale_factor=2)

    # kept for API parity with the reference's static factory methods
    make_residual_layer = staticmethod(lambda inplane, plane, layer_num, stride=1: _blocks(inplane, plane, layer_num, stride))
    make_hg_layer = staticmethod(lambda inplane, plane, layer_num: _blocks(inplane, plane, layer_num, 2))
    make_reverse_residual_layer = staticmethod(lambda inplane, plane, layer_num, stride=1: _blocks(inplane, plane, layer_num, widen_last=True))
    make_pool_layer = staticmethod(lambda: nn.Sequential())
    make_upsample_layer = staticmethod(lambda: nn.Upsample(scale_factor=2))

    def forward(self, x):
        u0, l0 = self.up1[0], self.low1[0]
        if RF.sync_coalescing(u0.bn1) and not len(u0.skip_connection) and len(l0.skip_connection):
            # SyncBN across ranks: the first layers of both branches (up1's conv1, low1's stride-2 conv1 and its projection)
            # read the same x and are independent given x — ONE statistics exchange per direction for the three of them (the
            # joint node carries one sample count per layer).  x's other consumer is the identity skip of up1's first block.
            xa, xb, _ = RF.fanout_shared(x, 2)
            outs = RF.conv_bn_act_multi(xa, u0.first_layers() + l0.first_layers())
            up1 = u0.tail(outs[0], xb)
            for blk in list(self.up1)[1:]:
                up1 = blk(up1)
            low1 = l0.tail(outs[1], outs[2])
            for blk in list(self.low1)[1:]:
                low1 = blk(low1)
            return RF.upsample_add(up1, self.low3(self.low2(low1)))
        xa, xb, _ = RF.fanout_shared(x, 2)              # both branches start with a residual block: one accumulator
        up1 = self.up1(xa)
        low3 = self.low3(self.low2(self.low1(xb)))
        # nearest x2 -> bilinear(align_corners) to up1's size -> add, without the 4x intermediate
        return RF.upsample_add(up1, low3)


class HourglassNet(nn.Module):
    def __init__(self, num_stacks=2, n=HG104['n'], inplanes=HG104['inplanes'], layer_nums=HG104['layer_nums'],
                 stem=HG104['stem'], num_feats=HG104['num_feats']):
        super().__init__()
        inplanes, layer_nums = list(inplanes), list(layer_nums)
        self.inplanes = stem
        self.num_feats = num_feats
        self.num_stacks = num_stacks
        self.pre_layer = nn.Sequential(
            nn.Conv2d(3, stem, kernel_size=7, stride=2, padding=3, bias=False),
            nn.BatchNorm2d(stem),
            nn.ReLU(inplace=True),
            ResidualBlock(stem, 2 * stem, 2))
        assert 2 * stem == inplanes[0], "the stem's ResidualBlock must produce inplanes[0] channels"
        self.hgs = nn.ModuleList([Hourglass(n, inplanes, layer_nums) for _ in range(num_stacks)])
        self.convs = nn.ModuleList([ConvBNRelu(3, inplanes[0], num_feats, with_relu=False) for _ in range(num_stacks)])
        self.residual = nn.ModuleList([ResidualBlock(inplanes[0], inplanes[0]) for _ in range(num_stacks - 1)])
        self.inter_ = nn.ModuleList([nn.Sequential(nn.Conv2d(inplanes[0], inplanes[0], (1, 1), bias=False),
                                                   nn.BatchNorm2d(inplanes[0])) for _ in range(num_stacks - 1)])
        self.conv_ = nn.ModuleList([nn.Sequential(nn.Conv2d(num_feats, inplanes[0], (1, 1), bias=False),
                                                  nn.BatchNorm2d(inplanes[0])) for _ in range(num_stacks - 1)])
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        """-> list of num_stacks pre-ReLU feature maps [B, num_feats, H/4, W/4] (NHWC memory)."""
        # Under cfg.Model.bf16 the activations that stay inside the backbone may exist as bf16 images only (ops.phantom_scope:
        # every layer that reads them here reads the image); what leaves it — the feature maps — is produced outside the scope.
        # Training only: at inference the memory saving buys nothing, the up-path sums keep their fp32 values.
        with ops.phantom_scope(self.training):
            pre = RF.conv_bn_act(x, self.pre_layer[0], self.pre_layer[1], relu=True)
            pre = self.pre_layer[3](pre)
            outs = []
            for i in range(self.num_stacks):
                last = i == self.num_stacks - 1
                pa, pb = (pre, None) if last else RF.fanout_shared(pre, 2)[:2]
                hg = self.hgs[i](pa)
                with ops.phantom_scope(False):
                    feat = self.convs[i](hg)
                outs.append(feat)
                if not last:
                    act = RF.relu(feat)
                    a = RF.conv_bn_act(pb, self.inter_[i][0], self.inter_[i][1], relu=False)
                    pre = RF.conv_bn_act(act, self.conv_[i][0], self.conv_[i][1], relu=True, residual=a)
                    pre = self.residual[i](pre)
        return outs


def hourglass_net(num_stacks=2, pretrained_path='./hourglass.pth'):
    """backbones/hourglass.py:202-210.  The reference unconditionally torch.load()s
    './hourglass.pth' (strict=False); that file ships with neither repository, so a missing file
    leaves the default initialisation in place instead of raising — with a warning, because a training run that
    expected the pretrained backbone would otherwise silently start from scratch.  `pretrained_path=None` asks for
    the random initialisation explicitly (tests, the synthetic benchmark)."""
    model = HourglassNet(num_stacks=num_stacks)
    if pretrained_path:
        if os.path.exists(pretrained_path):
            model.load_state_dict(torch.load(pretrained_path, map_location='cpu'), strict=False)
        else:
            import warnings
            warnings.warn("hourglass_net: %r not found — the backbone keeps its random initialisation (the reference "
                          "raises FileNotFoundError here)" % pretrained_path, RuntimeWarning, stacklevel=2)
    return model


def hourglass_tiny(num_stacks=2):
    """Builder-defined "hourglass-tiny" (BASELINE.json config 1): the same classes, small numbers."""
    return HourglassNet(num_stacks=num_stacks, **HG_TINY)
