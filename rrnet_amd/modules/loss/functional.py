"""modules/loss/functional.py:25-51 of the reference (`focal_loss_for_hm`) on the fused kernel.

The reference takes `pred` = clamp(sigmoid(logits), 1e-4, 1-1e-4) (operators/rrnet_operator.py:55);
the fused kernel wants the logits, so that sigmoid, clamp, the focal terms and the three global
sums are a single HBM pass.  `focal_loss_for_hm(pred, gt)` keeps the reference signature by
inverting the (monotone) sigmoid; the operators call `focal_loss_hm_from_logits` directly."""
import torch

from rrnet_amd.functional import focal_loss_hm_from_logits


def focal_loss_for_hm(pred, gt):
    p = pred.clamp(1e-4, 1 - 1e-4)
    logits = torch.log(p) - torch.log1p(-p)
    return focal_loss_hm_from_logits(logits, gt)
