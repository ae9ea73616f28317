"""modules/loss/regl1loss.py:5-17 (RegL1Loss) on the gather/scatter kernels."""
import torch.nn as nn

from rrnet_amd.functional import reg_l1_loss


class RegL1Loss(nn.Module):
    def forward(self, output, mask, ind, target):
        return reg_l1_loss(output, mask, ind, target)
