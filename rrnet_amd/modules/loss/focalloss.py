"""modules/loss/focalloss.py:15-20 (FocalLossHM)."""
import torch.nn as nn

from .functional import focal_loss_for_hm
from rrnet_amd.functional import focal_loss_hm_from_logits


class FocalLossHM(nn.Module):
    def forward(self, out, target):
        return focal_loss_for_hm(out, target)

    @staticmethod
    def from_logits(logits, target):
        return focal_loss_hm_from_logits(logits, target)
