"""ORACLE (test infrastructure only) — single-scale inference post-process of one frame on the CPU, composed
from the restatements in oracle/ops.py, oracle/model.py and oracle/nms.py.  Only tests/, smoke() and the
cpu_baseline leg of the benchmarks import this.

Follows models/rrnet.py:30-54 (decode, stage-1 NMS, RoIAlign on relu(feat), stage-2 head) and the body of
operators/rrnet_operator.py:262-279 for one scale (generate_bbox, `score > 0.01`, sort by score, `_ext_nms`,
sort by score).  `torch.sort` is asked for a stable order here so that the result is reproducible; the
reference's order among exactly equal scores is unspecified."""
import numpy as np
import torch

from oracle import model as omodel
from oracle import nms as onms
from oracle import ops as oops


def postprocess_frame(P, hm, wh, offset, feat, k=1500, scale_factor=4, score_thr=0.01, nms_type='nms',
                      relu_feat=True):
    """hm [1,C,H,W] logits, wh/offset [1,2,H,W], feat [1,256,H,W]; P = oracle.model.Params holding the
    `head_detector.*` weights (eval-mode BN).  -> float32 numpy [n,6] = x,y,w,h,score,cls+1, score-descending."""
    with torch.no_grad():
        bboxs = oops.transform_bbox(hm, wh, offset, k)
        kept = oops.stage1_nms(bboxs[0], nms_type, True)
        rois = torch.cat((torch.zeros((kept.size(0), 1)), kept[:, :4]), dim=1)
        f = torch.relu(feat) if relu_feat else feat
        roi_feat = oops.roi_align(f, rois, (3, 3))
        reg = omodel.stage2_head(P, roi_feat)
        outs = (None, None, None, reg, rois, kept[:, 4], kept[:, 5])
        _, pred = oops.generate_bbox(outs, 0, scale_factor)
        pred = pred[pred[:, 4] > score_thr]
        idx = torch.sort(pred[:, 4], descending=True, stable=True)[1]
        pred = pred[idx]
        out = onms.ext_nms(pred.numpy())
        if out.shape[0] == 0:
            return out.reshape(0, 6)
        idx = np.argsort(-out[:, 4], kind='stable')
        return out[idx]
