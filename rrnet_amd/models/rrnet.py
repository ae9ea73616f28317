"""RRNet on the MI355X kernels — API of the reference's models/rrnet.py (RRNet :11-157).

forward(x, k=1500) returns the same 7-tuple
    (hms: list[num_stacks], whs, offsets, stage2_reg [R,4], bxyxys [R,5], scores [R], clses [R]).
The reference's per-image x per-class Python loops with host round trips (:37-46, :56-80) are
one batched sequence of launches here: decode (sigmoid + top-K + gather) -> stable grouping by
class -> hard / soft NMS over all (image, class) segments at once -> packing; the row order is
the reference's (images ascending, classes in unique() order, NMS order inside a class).
"""
import torch
import torch.nn as nn

from rrnet_amd import functional as RF
from rrnet_amd import ops
from rrnet_amd.detectors.centernet_detector import CenterNetDetector, CenterNetWHDetector
from rrnet_amd.detectors.fasterrcnn_detector import FasterRCNNDetector
from rrnet_amd.ext.nms.nms_wrapper import soft_nms_segments
from rrnet_amd.utils.model_tools import get_backbone


def stage1_proposals(hm, wh, offset, k, num_classes, nms_type='nms', nms_per_class=True, peak_filter=False,
                     want_provenance=False, want_offsets=False):
    """Decode + stage-1 NMS for the whole batch (models/rrnet.py:31-49, 56-138).
    -> bxyxys [R,5], scores [R], clses [R] (detached device tensors) and, on request, the heat-map pixel
    every RoI was decoded from (for the backward of the box assembly)."""
    with torch.no_grad():
        hm, wh, offset = ops.to_nhwc(hm.detach()), ops.to_nhwc(wh.detach()), ops.to_nhwc(offset.detach())
        # peak_filter: optional `_ctnet_nms` (the reference's decode never applies it), tested inside the scan
        boxes, pix = ops.decode_topk(hm, wh, offset, k, is_logits=True, want_pix=True, peak_filter=peak_filter)
        b = boxes.shape[0]
        if nms_per_class:
            grouped, seg_off, seg_len = ops.group_by_class(boxes, num_classes)
            segs_per_image = num_classes
        else:
            grouped = boxes.clone()
            seg_off = torch.arange(0, (b + 1) * k, k, dtype=torch.int32, device=boxes.device)
            seg_len = None
            segs_per_image = 1
        rows = grouped.view(-1, 6)
        if nms_type == 'soft_nms':
            n_out = soft_nms_segments(rows, seg_off, k, sigma=0.5, Nt=0.7, threshold=0.1, method=2, seg_len=seg_len)
        else:
            n_out = ops.hard_nms_segments(rows, seg_off, k, 0.7, seg_len)
        rois, scores, clses, _, row_off = ops.pack_segments(rows, seg_off, n_out, segs_per_image, want_offsets=True)
        roi_pix = ops.roi_provenance(rois, scores, clses, boxes, pix) if want_provenance else None
    if want_offsets:   # row offsets of the (image, class) segments inside the packed list
        return rois, scores, clses, row_off
    return (rois, scores, clses, roi_pix) if want_provenance else (rois, scores, clses)


class RRNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_stacks = cfg.Model.num_stacks
        self.num_classes = cfg.num_classes
        self.nms_type = cfg.Model.nms_type_for_stage1
        self.nms_per_class = cfg.Model.nms_per_class_for_stage1
        self.backbone = get_backbone(cfg.Model.backbone, num_stacks=self.num_stacks)
        # builder-defined option (BASELINE configs[3]): the three heads' 3x3 convolutions as ext/dcn DCN layers,
        # optionally with bf16 matrix operands in their forward
        dcn = bool(getattr(cfg.Model, "dcn_heads", False))
        dcn_bf16 = bool(getattr(cfg.Model, "dcn_bf16", False))
        self.hm = CenterNetDetector(planes=self.num_classes, num_stacks=self.num_stacks, hm=True, dcn=dcn, dcn_bf16=dcn_bf16)
        self.wh = CenterNetWHDetector(planes=1, num_stacks=self.num_stacks, dcn=dcn, dcn_bf16=dcn_bf16)
        self.offset_reg = CenterNetDetector(planes=2, num_stacks=self.num_stacks, dcn=dcn, dcn_bf16=dcn_bf16)
        self.head_detector = FasterRCNNDetector()
        # builder-defined (BASELINE configs[3] "bf16"; the reference is fp32-only): bf16 matrix operands with fp32
        # accumulation in every convolution of the backbone and the heads (csrc/conv_bf16.hip); activations, weights,
        # BatchNorm statistics, losses and the optimizer stay fp32
        # cfg.Model.conv_math = "f16x3": split-operand kernels (two fp16 parts per operand, fp32-level accuracy; ops.math_mode)
        self.bf16 = ops.math_mode(cfg.Model)

    def forward(self, x, k=1500):
        with ops.bf16_scope(self.bf16):
            return self._forward(x, k)

    def _forward(self, x, k=1500):
        feats = self.backbone(x)
        last_a, last_b = RF.fanout(feats[-1], 2)
        hms, whs, offsets = self.forward_stage1(list(feats[:-1]) + [last_a])
        # hard NMS keeps the boxes attached to the graph in the reference (models/rrnet.py:69-70: index select
        # of a differentiable tensor); the soft-NMS path detaches them (:65)
        diff = torch.is_grad_enabled() and self.nms_type != 'soft_nms' and whs[-1].requires_grad
        if diff:
            bxyxys, scores, clses, roi_pix = stage1_proposals(hms[-1], whs[-1], offsets[-1], k, self.num_classes,
                                                              self.nms_type, self.nms_per_class, want_provenance=True)
            bxyxys = RF.differentiable_proposals(whs[-1], offsets[-1], bxyxys, roi_pix)
        else:
            bxyxys, scores, clses = stage1_proposals(hms[-1], whs[-1], offsets[-1], k, self.num_classes,
                                                     self.nms_type, self.nms_per_class)
        roi_feat = RF.roi_align(RF.relu(last_b), bxyxys, (3, 3))
        stage2_reg = self.forward_stage2(roi_feat)
        return hms, whs, offsets, stage2_reg, bxyxys, scores, clses

    def forward_stage1(self, feats):
        from rrnet_amd.models.centernet import run_stage1_heads
        return run_stage1_heads(feats, (self.hm, self.wh, self.offset_reg), self.num_stacks)

    def forward_stage2(self, feats):
        return self.head_detector(feats)

    def transform_bbox(self, hm, wh, offset, k=250):
        """models/rrnet.py:117-138 -> [B,k,6] in feature coordinates."""
        return ops.decode_topk(ops.to_nhwc(hm), ops.to_nhwc(wh), ops.to_nhwc(offset), k, is_logits=True)

    # ---- the reference's helper methods (models/rrnet.py:56-115), same signatures and results, on the kernels ----
    def nms(self, bbox):
        """models/rrnet.py:56-80 for ONE image: bbox [K,6] (score-descending, as transform_bbox emits) -> kept rows,
        classes in unique() order, NMS order inside a class.  `forward` runs the batched form (stage1_proposals)."""
        if bbox.size(0) == 0:
            return bbox
        b = bbox.detach().float().contiguous()
        k = b.size(0)
        with torch.no_grad():
            if self.nms_per_class:
                lo, hi = int(b[:, 5].min()), int(b[:, 5].max())
                nc = hi - lo + 1
                grouped, seg_off, seg_len = ops.group_by_class(b.view(1, k, 6), nc, cls_base=lo)
            else:
                nc, grouped, seg_len = 1, b.view(1, k, 6).clone(), None
                seg_off = torch.tensor([0, k], dtype=torch.int32, device=b.device)
            rows = grouped.view(-1, 6)
            if self.nms_type == 'soft_nms':
                n_out = soft_nms_segments(rows, seg_off, k, sigma=0.5, Nt=0.7, threshold=0.1, method=2, seg_len=seg_len)
            else:
                n_out = ops.hard_nms_segments(rows, seg_off, k, 0.7, seg_len)
            return ops.pack_segments(rows, seg_off, n_out, nc, want_rois=False, want_rows=True)[3]

    @staticmethod
    def _gather_feat(feat, ind, mask=None):
        """models/rrnet.py:82-91: feat [B,N,D], ind [B,k] -> [B,k,D] (index plumbing, torch)."""
        dim = feat.size(2)
        ind = ind.long().unsqueeze(2).expand(ind.size(0), ind.size(1), dim)
        feat = feat.gather(1, ind)
        if mask is not None:
            mask = mask.unsqueeze(2).expand_as(feat)
            feat = feat[mask].view(-1, dim)
        return feat

    def _topk(self, scores, k=1500):
        """models/rrnet.py:93-109: scores [B,C,H,W] (already sigmoid-ed) -> (score [B,k], pixel index [B,k], class
        [B,k] int, ys [B,k], xs [B,k]).  The reference's two torch.topk calls equal one global top-k per image, which
        the decode kernel computes (ties: reference flat index ascending)."""
        b, c, h, w = scores.shape
        zero = ops.zeros_nhwc(b, 2, h, w, scores.device)
        rows, pix = ops.decode_topk(ops.to_nhwc(scores.detach().float()), zero, zero, k, is_logits=False, want_pix=True)
        pix = pix.long()
        return rows[..., 4], pix, rows[..., 5].int(), (pix // w).float(), (pix % w).float()

    def _transpose_and_gather_feat(self, feat, ind):
        """models/rrnet.py:111-115: feat [B,D,H,W], ind [B,k] -> [B,k,D]; on NHWC memory the permute is a view."""
        feat = ops.to_nhwc(feat).permute(0, 2, 3, 1)
        return self._gather_feat(feat.reshape(feat.size(0), -1, feat.size(3)), ind)
