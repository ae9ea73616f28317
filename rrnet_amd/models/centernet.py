"""CenterNet = backbone + the three stage-1 heads, the single-stage sibling of RRNet
(reference: models/centernet.py:8-32; the offset head is called `reg` there, `offset_reg` in RRNet)."""
import torch.nn as nn

from rrnet_amd import functional as RF
from rrnet_amd.detectors.centernet_detector import CenterNetDetector, CenterNetWHDetector
from rrnet_amd.utils.model_tools import get_backbone


def run_stage1_heads(feats, heads, num_stacks):
    """Per stack: ReLU of the backbone feature, fanned out to the heads (their first 3x3 convolutions share one
    gradient accumulator).  `heads`: modules called as head(x, stack) -> one list of per-stack outputs per head."""
    outs = tuple([] for _ in heads)
    for stack in range(num_stacks):
        views = RF.fanout_shared(RF.relu(feats[stack]), len(heads))[:-1]
        for head, view, acc in zip(heads, views, outs):
            acc.append(head(view, stack))
    return outs


class CenterNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        stacks = cfg.Model.num_stacks
        self.num_stacks, self.num_classes = stacks, cfg.num_classes
        self.backbone = get_backbone(cfg.Model.backbone, num_stacks=stacks)
        self.hm = CenterNetDetector(planes=self.num_classes, num_stacks=stacks, hm=True)
        self.wh = CenterNetWHDetector(planes=1, num_stacks=stacks)
        self.reg = CenterNetDetector(planes=2, num_stacks=stacks)

    def forward(self, input):
        return run_stage1_heads(self.backbone(input), (self.hm, self.wh, self.reg), self.num_stacks)
