"""CenterNet (models/centernet.py:8-32 of the reference): backbone + hm / wh / reg heads."""
import torch.nn as nn

from rrnet_amd import functional as RF
from rrnet_amd.detectors.centernet_detector import CenterNetDetector, CenterNetWHDetector
from rrnet_amd.utils.model_tools import get_backbone


class CenterNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_stacks = cfg.Model.num_stacks
        self.num_classes = cfg.num_classes
        self.backbone = get_backbone(cfg.Model.backbone, num_stacks=self.num_stacks)
        self.hm = CenterNetDetector(planes=cfg.num_classes, num_stacks=self.num_stacks, hm=True)
        self.wh = CenterNetWHDetector(planes=1, num_stacks=self.num_stacks)
        self.reg = CenterNetDetector(planes=2, num_stacks=self.num_stacks)

    def forward(self, input):
        feats = self.backbone(input)
        hms, whs, regs = [], [], []
        for i in range(self.num_stacks):
            fa, fb, fc = RF.fanout(RF.relu(feats[i]), 3)
            hms.append(self.hm(fa, i))
            whs.append(self.wh(fb, i))
            regs.append(self.reg(fc, i))
        return hms, whs, regs
