"""Re-regression (stage-2) head — detectors/fasterrcnn_detector.py:6-18 of the reference:
Bottleneck(256,64) -> global average pool over the 3x3 RoI -> 1x1 conv to 4 deltas."""
import os

import torch
import torch.nn as nn

from rrnet_amd import functional as RF
from rrnet_amd import ops
from rrnet_amd.backbones.resnet import Bottleneck

FUSED_TAIL = True      # A/B switch: conv3 fused into the inference tail


class FasterRCNNDetector(nn.Module):
    def __init__(self):
        super().__init__()
        self.top_layer = Bottleneck(inplanes=256, planes=64)
        self.regressor = nn.Conv2d(256, 4, kernel_size=1)

    def forward(self, feat):
        if not self.training and not torch.is_grad_enabled() and self.top_layer.downsample is None:
            # inference: bn3 + residual + ReLU + average pool in one pass, the activated tensor is never written
            b = self.top_layer
            x = ops.to_nhwc(feat)
            out = RF.conv_bn_act(x, b.conv1, b.bn1, relu=True)
            out = RF.conv_bn_act(out, b.conv2, b.bn2, relu=True)
            scale, shift = ops.bn_eval_coeffs(b.bn3.weight, b.bn3.bias, b.bn3.running_mean, b.bn3.running_var, b.bn3.eps)
            k3 = b.conv3.in_channels
            if FUSED_TAIL and k3 in (32, 64) and b.conv3.out_channels <= 256 and b.conv3.out_channels % 4 == 0:
                # conv3 + bn3 + residual + ReLU + average pool: conv3's [R*9, 256] output never reaches HBM
                feat = ops.conv1x1_bn_res_relu_avgpool(out, ops.to_nhwc(b.conv3.weight), scale, shift, x)
            else:
                y = ops.conv_fprop(out, ops.to_nhwc(b.conv3.weight), None, 1, (0, 0), False)
                feat = ops.bn_res_relu_avgpool(y, scale, shift, x)
        else:
            feat = self.top_layer(feat)
            feat = RF.global_avg_pool(feat)
        reg = RF.conv_bias(feat, self.regressor)
        return reg.reshape(reg.size(0), reg.size(1))
