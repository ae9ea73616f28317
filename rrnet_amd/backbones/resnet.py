"""Only the piece of the reference's backbones/resnet.py that is on the RRNet path: Bottleneck
(:17-53), used by the stage-2 head (detectors/fasterrcnn_detector.py:10).  The ResNet
backbones themselves belong to the RetinaNet model and are out of scope."""
import torch.nn as nn

from rrnet_amd import functional as RF


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        xa, xb = RF.fanout(x, 2)
        out = RF.conv_bn_act(xa, self.conv1, self.bn1, relu=True)
        out = RF.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        residual = self.downsample(xb) if self.downsample is not None else xb
        return RF.conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=residual)
