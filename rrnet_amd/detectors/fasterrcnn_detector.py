"""Re-regression (stage-2) head — detectors/fasterrcnn_detector.py:6-18 of the reference:
Bottleneck(256,64) -> global average pool over the 3x3 RoI -> 1x1 conv to 4 deltas."""
import torch.nn as nn

from rrnet_amd import functional as RF
from rrnet_amd.backbones.resnet import Bottleneck


class FasterRCNNDetector(nn.Module):
    def __init__(self):
        super().__init__()
        self.top_layer = Bottleneck(inplanes=256, planes=64)
        self.regressor = nn.Conv2d(256, 4, kernel_size=1)

    def forward(self, feat):
        feat = self.top_layer(feat)
        feat = RF.global_avg_pool(feat)
        reg = RF.conv_bias(feat, self.regressor)
        return reg.reshape(reg.size(0), reg.size(1))
