"""Batched inference post-process: heat-map decode -> stage-1 NMS -> RoIAlign -> re-regression head ->
stage-2 boxes -> score filter -> per-class gaussian Soft-NMS -> per-frame score order.

This is what `RRNet.forward` (after the backbone and stage-1 heads, models/rrnet.py:30-54) followed by the
single-scale body of `RRNetOperator.evaluation_process` (operators/rrnet_operator.py:262-279: generate_bbox,
`score > 0.01`, sort, `_ext_nms`, sort) computes for ONE frame with Python loops over classes and a D2H copy
per class — here for a whole batch of frames with one launch per stage and two host reads per batch (the RoI
count that sizes the head's tensors, and the final box count).

Ordering note: decode emits rows score-descending, grouping / hard NMS are stable, so inside every
(frame, class) segment the stage-2 boxes already are in the order the reference's global
`torch.sort(score, descending=True)` + `pred_bbox[:, 5] == cls` selection hands to `soft_nms`
(ties: torch.sort leaves them unspecified; here they keep decode order)."""
import os

import torch

from rrnet_amd import ops
from rrnet_amd.ext.nms.nms_wrapper import soft_nms_segments
from rrnet_amd.models.rrnet import stage1_proposals


# RoIAlign processing order: per-frame spatial sort (rr_roi_spatial_order).  Off by default since round 3: with eight
# footprint pixels in flight per wave the kernel runs as fast in decode order (2.66-2.73 ms against 2.62-2.87 ms sorted, per
# 128 frames at config 5) and the sort is a launch of its own (0.05 ms); RR_ROI_ORDER=1 turns it on.
ROI_SPATIAL_ORDER = False


@torch.no_grad()
def refine_frames(hm, wh, offset, feat, head_detector, k=1500, num_classes=10, scale_factor=4, score_thr=0.01,
                  nms_type='nms', relu_feat=True):
    """hm [B,C,H,W] logits, wh / offset [B,2,H,W], feat [B,256,H,W] (pre-ReLU backbone output unless
    relu_feat=False), head_detector = FasterRCNNDetector in eval mode.
    -> boxes [n,6] = x,y,w,h,score,cls+1 (image coordinates, frames back to back, each score-descending),
       frame_off int32 [B+1] (device).  Raises ZeroDivisionError where the reference's soft_nms would."""
    b = hm.shape[0]
    rois, scores, clses, row_off = stage1_proposals(hm, wh, offset, k, num_classes, nms_type, True, want_offsets=True)
    feat = ops.to_nhwc(feat)
    if relu_feat:
        feat = ops.relu_fwd(feat)
    # RoIs of a frame in spatial order (overlapping footprints back to back on one XCD: shared rows come from L2)
    order = ops.roi_spatial_order(rois, row_off[::num_classes].contiguous()) if ROI_SPATIAL_ORDER else None
    roi_feat = ops.roi_align_fwd(feat, rois, (3, 3), order=order)
    reg = head_detector(roi_feat)                                   # [R,4]
    boxes6, seg_len = ops.refine_boxes(rois, reg, scores, clses, row_off, scale_factor, score_thr)
    n_out, err = soft_nms_segments(boxes6, row_off, k, sigma=0.5, Nt=0.7, threshold=0.1, method=2, seg_len=seg_len,
                                   check=False)
    out6, frame_off = ops.finalize_frames(boxes6, row_off, n_out, b, num_classes, k)
    fo = frame_off.cpu()                                            # one sync: final counts (+ error flag)
    if int(err.item()) != 0:
        raise ZeroDivisionError("float division")
    return out6[:int(fo[-1])], frame_off
