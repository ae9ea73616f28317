"""ORACLE (test infrastructure only) — torch-CPU / numpy restatement of the reference's loss,
decode, stage-1 NMS, RoIAlign, box-target and box-decode arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this file.
Everything here runs on CPU tensors in fp32, in the operation order of the cited reference
lines, so that it can be pinned to goldens generated from the reference (tools/gen_goldens.py).

torchvision (roi_align / nms / box_iou) is a third-party dependency of the reference that is
NOT under /root/reference and is not installed in this image; the reference pins it only as
"PyTorch 1.1.0" (readme.md:19-22 => torchvision 0.3.0).  Those three functions restate the
published torchvision-0.3 algorithms and are **parity-unpinned** by the reference (no test or
golden vector of the reference covers them).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import nms as onms


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def focal_loss_for_hm(pred, gt):
    """/root/reference/modules/loss/functional.py:25-51.  `pred` is already
    clamp(sigmoid(x), 1e-4, 1-1e-4) (rrnet_operator.py:55)."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    neg_w = torch.pow(1 - gt, 4)
    pos_term = torch.log(pred) * torch.pow(1 - pred, 2) * pos
    neg_term = torch.log(1 - pred) * torch.pow(pred, 2) * neg_w * neg
    n_pos = pos.sum()
    pos_sum = pos_term.sum()
    neg_sum = neg_term.sum()
    if n_pos == 0:
        return 0 - neg_sum
    return 0 - (pos_sum + neg_sum) / n_pos


def hm_loss_from_logits(logits, gt):
    """rrnet_operator.py:55-57: clamp(sigmoid) then focal."""
    p = torch.clamp(torch.sigmoid(logits), min=1e-4, max=1 - 1e-4)
    return focal_loss_for_hm(p, gt)


def reg_l1_loss(output, mask, ind, target):
    """/root/reference/modules/loss/regl1loss.py:9-17.  output [B,C,H,W]; mask, ind [B,M,1]
    (float); target [B,M,C]."""
    b, c = output.size(0), output.size(1)
    pred = output.permute(0, 2, 3, 1).contiguous().view(b, -1, c)
    idx = ind.long().expand(ind.size(0), ind.size(1), c)
    pred = pred.gather(1, idx)
    m = mask.expand_as(pred).float()
    loss = F.l1_loss(pred * m, target * m, reduction='sum')
    return loss / (m.sum() + 1e-4)


# --------------------------------------------------------------------------------------
# decode (models/rrnet.py:82-138)
# --------------------------------------------------------------------------------------
def _gather_rows(feat, ind):
    """rrnet.py:82-91 without the mask branch (never used on this path)."""
    dim = feat.size(2)
    return feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim))


def topk_decode(scores, k):
    """rrnet.py:93-109 (two-level top-k)."""
    batch, cat, height, width = scores.size()
    s1, i1 = torch.topk(scores.view(batch, cat, -1), k)
    i1 = i1 % (height * width)
    ys = (i1 / width).int().float()
    xs = (i1 % width).int().float()
    s2, i2 = torch.topk(s1.view(batch, -1), k)
    clses = (i2 / k).int()
    inds = _gather_rows(i1.view(batch, -1, 1), i2).view(batch, k)
    ys = _gather_rows(ys.view(batch, -1, 1), i2).view(batch, k)
    xs = _gather_rows(xs.view(batch, -1, 1), i2).view(batch, k)
    return s2, inds, clses, ys, xs


def _nhwc_gather(feat, ind):
    """rrnet.py:111-115."""
    feat = feat.permute(0, 2, 3, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(3))
    return _gather_rows(feat, ind)


def transform_bbox(hm, wh, offset, k):
    """rrnet.py:117-138: logits -> [B,k,6] = [x1,y1,x2,y2,score,cls] in feature coordinates.
    No 3x3 peak filter (the reference never applies `_ctnet_nms`)."""
    b = hm.size(0)
    hm = torch.sigmoid(hm)
    scores, inds, clses, ys, xs = topk_decode(hm, k)
    off = _nhwc_gather(offset, inds).view(b, k, 2)
    xs = xs.view(b, k, 1) + off[:, :, 0:1]
    ys = ys.view(b, k, 1) + off[:, :, 1:2]
    whg = _nhwc_gather(wh, inds).clamp(min=0).view(b, k, 2)
    clses = clses.view(b, k, 1).float()
    scores = scores.view(b, k, 1)
    px = xs - whg[..., 0:1] / 2
    py = ys - whg[..., 1:2] / 2
    pw = whg[..., 0:1]
    ph = whg[..., 1:2]
    return torch.cat([px, py, pw + px, ph + py, scores, clses], dim=2)


def ctnet_transform_bbox(hm, wh, offset, k=250, scale_factor=4):
    """/root/reference/operators/centernet_operator.py:152-178: logits -> rows [x,y,w,h,score,cls+1] of IMAGE 0 in
    image coordinates (x scale_factor), wh NOT clamped, `offset=None` -> +0.5 centres, rows with score <= 0.01 dropped."""
    b = hm.size(0)
    hm = torch.sigmoid(hm)
    scores, inds, clses, ys, xs = topk_decode(hm, k)
    if offset is not None:
        off = _nhwc_gather(offset, inds).view(b, k, 2)
        xs = xs.view(b, k, 1) + off[:, :, 0:1]
        ys = ys.view(b, k, 1) + off[:, :, 1:2]
    else:
        xs = xs.view(b, k, 1) + 0.5
        ys = ys.view(b, k, 1) + 0.5
    whg = _nhwc_gather(wh, inds).view(b, k, 2)
    clses = clses.view(b, k, 1).float() + 1
    scores = scores.view(b, k, 1)
    px = (xs - whg[..., 0:1] / 2) * scale_factor
    py = (ys - whg[..., 1:2] / 2) * scale_factor
    pw = whg[..., 0:1] * scale_factor
    ph = whg[..., 1:2] * scale_factor
    pred = torch.cat([px[0], py[0], pw[0], ph[0], scores[0], clses[0]], dim=1)
    return pred[pred[:, 4] > 0.01, :]


def ctnet_peak_filter(heat, kernel=3):
    """/root/reference/operators/centernet_operator.py:204-210 (`_ctnet_nms`, dead code in the
    reference; restated because north_star names the 3x3 peak pick)."""
    pad = (kernel - 1) // 2
    hmax = F.max_pool2d(heat, (kernel, kernel), stride=1, padding=pad)
    keep = (hmax == heat).float()
    return heat * keep


# --------------------------------------------------------------------------------------
# torchvision-0.3 restatements (parity unpinned by the reference — see module docstring)
# --------------------------------------------------------------------------------------
def box_iou(a, b):
    """torchvision.ops.box_iou as called at rrnet_operator.py:72 (no +1 convention)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area_a[:, None] + area_b - inter)


def tv_nms(boxes, scores, thresh):
    """torchvision.ops.nms as called at models/rrnet.py:69,78 -> kept indices, score-descending."""
    keep = onms.hard_nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), float(thresh))
    return torch.from_numpy(keep).to(boxes.device)


def _bilinear_weights(y, x, height, width):
    """RoIAlign sample -> (valid, 4 flat indices, 4 weights); torchvision-0.3 roi_align
    `bilinear_interpolate` (legacy, aligned=False)."""
    if y < -1.0 or y > height or x < -1.0 or x > width:
        return False, None, None
    y = np.float32(max(y, np.float32(0)))
    x = np.float32(max(x, np.float32(0)))
    y_low = int(y)
    x_low = int(x)
    if y_low >= height - 1:
        y_high = y_low = height - 1
        y = np.float32(y_low)
    else:
        y_high = y_low + 1
    if x_low >= width - 1:
        x_high = x_low = width - 1
        x = np.float32(x_low)
    else:
        x_high = x_low + 1
    ly = np.float32(y - np.float32(y_low))
    lx = np.float32(x - np.float32(x_low))
    hy = np.float32(np.float32(1) - ly)
    hx = np.float32(np.float32(1) - lx)
    w = (np.float32(hy * hx), np.float32(hy * lx), np.float32(ly * hx), np.float32(ly * lx))
    idx = (y_low * width + x_low, y_low * width + x_high, y_high * width + x_low, y_high * width + x_high)
    return True, idx, w


def roi_align(feat, rois, output_size, spatial_scale=1.0, sampling_ratio=-1):
    """torchvision.ops.roi_align(input[N,C,H,W], rois[K,5]=(batch,x1,y1,x2,y2), output_size)
    as called at models/rrnet.py:51: spatial_scale 1, sampling_ratio -1 (adaptive
    ceil(roi_size/bins) samples per bin), RoI width/height clamped to >= 1, legacy
    (aligned=False) coordinates.  Differentiable w.r.t. `feat` (sparse-matrix formulation:
    out[r] = S_r @ feat[b_r] with the bilinear weights in S_r), sample positions and weights in fp32 arithmetic
    (an fp64 `feat` only widens the final weighted sum: used for the tests' fp64 'truth' runs)."""
    ph, pw = output_size
    n, c, height, width = feat.shape
    rois_np = rois.detach().cpu().numpy().astype(np.float32)
    k = rois_np.shape[0]
    rows, cols, vals = [], [], []
    f32 = np.float32
    for r in range(k):
        b = int(rois_np[r, 0])
        x1 = f32(rois_np[r, 1] * f32(spatial_scale))
        y1 = f32(rois_np[r, 2] * f32(spatial_scale))
        x2 = f32(rois_np[r, 3] * f32(spatial_scale))
        y2 = f32(rois_np[r, 4] * f32(spatial_scale))
        rw = f32(max(f32(x2 - x1), f32(1)))
        rh = f32(max(f32(y2 - y1), f32(1)))
        bh = f32(rh / f32(ph))
        bw = f32(rw / f32(pw))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        count = f32(gh * gw)
        for i in range(ph):
            for j in range(pw):
                orow = (r * ph + i) * pw + j
                for iy in range(gh):
                    y = f32(y1 + f32(i) * bh + f32(f32(iy) + f32(0.5)) * bh / f32(gh))
                    for ix in range(gw):
                        x = f32(x1 + f32(j) * bw + f32(f32(ix) + f32(0.5)) * bw / f32(gw))
                        ok, idx, w = _bilinear_weights(y, x, height, width)
                        if not ok:
                            continue
                        for q in range(4):
                            rows.append(orow)
                            cols.append(b * height * width + idx[q])
                            vals.append(w[q] / count)
    if k == 0:
        return feat.new_zeros((0, c, ph, pw))
    S = torch.sparse_coo_tensor(torch.tensor([rows, cols], dtype=torch.long),
                                torch.tensor(np.array(vals, dtype=np.float32)).to(feat.dtype),
                                (k * ph * pw, n * height * width)).coalesce()
    flat = feat.permute(0, 2, 3, 1).reshape(n * height * width, c)
    out = torch.sparse.mm(S, flat)                      # [k*ph*pw, c]
    return out.view(k, ph, pw, c).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------
# stage-1 NMS driver (models/rrnet.py:56-80)
# --------------------------------------------------------------------------------------
def stage1_nms(bbox, nms_type='nms', per_class=True):
    """bbox [K,6] -> kept rows.  Class loop in `unique()` (ascending) order."""
    if per_class:
        outs = []
        for cls in bbox[:, 5].unique():
            sub = bbox[bbox[:, 5] == cls]
            if nms_type == 'soft_nms':
                kept = onms.soft_nms(sub.detach().cpu().numpy(), Nt=0.7, threshold=0.1, method=2)
                outs.append(torch.from_numpy(kept))
            else:
                outs.append(sub[tv_nms(sub[:, :4], sub[:, 4], 0.7)])
        return torch.cat(outs)
    if nms_type == 'soft_nms':
        return torch.from_numpy(onms.soft_nms(bbox.detach().cpu().numpy(), Nt=0.7, threshold=0.1, method=2))
    return bbox[tv_nms(bbox[:, :4], bbox[:, 4], 0.7)]


# --------------------------------------------------------------------------------------
# stage-2 targets / loss / decode (operators/rrnet_operator.py)
# --------------------------------------------------------------------------------------
def generate_bbox_target(ex, gt):
    """rrnet_operator.py:86-102 (Faster-RCNN deltas with the +1 width convention)."""
    ew = ex[:, 2] - ex[:, 0] + 1.0
    eh = ex[:, 3] - ex[:, 1] + 1.0
    ecx = ex[:, 0] + 0.5 * ew
    ecy = ex[:, 1] + 0.5 * eh
    gw = gt[:, 2] - gt[:, 0] + 1.0
    gh = gt[:, 3] - gt[:, 1] + 1.0
    gcx = gt[:, 0] + 0.5 * gw
    gcy = gt[:, 1] + 0.5 * gh
    return torch.stack(((gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)), dim=1)


def criterion(outs, targets, num_stacks=2, scale_factor=4):
    """rrnet_operator.py:42-84.  Mutates targets' gt_annos in place (xywh -> xyxy, :67) exactly
    as the reference does.  Returns (hm, wh, off, s2) losses."""
    s1_hms, s1_whs, s1_offs, s2_reg, bxyxy, _scores, _ = outs
    gt_hms, gt_whs, gt_inds, gt_offs, gt_masks, gt_annos = targets
    bs = s1_hms[0].size(0)
    hm_loss = 0
    wh_loss = 0
    off_loss = 0
    for s in range(num_stacks):
        hm_loss = hm_loss + hm_loss_from_logits(s1_hms[s], gt_hms) / num_stacks
        wh_loss = wh_loss + reg_l1_loss(s1_whs[s], gt_masks, gt_inds, gt_whs) / num_stacks
        off_loss = off_loss + reg_l1_loss(s1_offs[s], gt_masks, gt_inds, gt_offs) / num_stacks
    s2_loss = 0
    gt_annos[:, :, 2:4] += gt_annos[:, :, 0:2]
    for b in range(bs):
        flag = bxyxy[:, 0] == b
        bbox = bxyxy[flag][:, 1:]
        gt = gt_annos[b]
        iou = box_iou(bbox * scale_factor, gt[:, :4])
        max_iou, max_idx = torch.max(iou, dim=1)
        pos = max_iou > 0.5
        if pos.sum() == 0:
            pos = torch.zeros_like(max_iou).bool()
            pos[0] = True
            factor = 0
        else:
            factor = 1
        tgt = generate_bbox_target(bbox[pos, :] * scale_factor, gt[max_idx[pos], :4])
        s2_loss = s2_loss + F.smooth_l1_loss(s2_reg[flag][pos], tgt) * factor / bs
    return hm_loss, wh_loss, off_loss, s2_loss


def generate_bbox(outs, batch_idx=0, scale_factor=4):
    """rrnet_operator.py:188-209.  The reference aliases one tensor through the xyxy->xywh and
    `+= 1` steps; the stage-1 boxes are materialised before the `+= 1`, the stage-2 boxes after."""
    _, _, _, s2_reg, bxyxy, scores, clses = outs
    flag = bxyxy[:, 0] == batch_idx
    reg = s2_reg[flag]
    box = bxyxy[flag, 1:] * scale_factor
    score = scores[flag]
    cls = clses[flag]
    box[:, 2:4] -= box[:, 0:2]                               # xywh
    s1 = torch.cat((box, score.view(-1, 1), torch.zeros((box.size(0), 1))), dim=1)
    box[:, 2:4] += 1
    cx = reg[:, 0] * box[:, 2] + box[:, 0] + box[:, 2] / 2
    cy = reg[:, 1] * box[:, 3] + box[:, 1] + box[:, 3] / 2
    w = reg[:, 2].exp() * box[:, 2]
    h = reg[:, 3].exp() * box[:, 3]
    s2 = torch.stack((cx - w / 2., cy - h / 2., w, h, score, cls.float() + 1), dim=1)
    return s1, s2


def save_result_lines(pred_bbox):
    """rrnet_operator.py:234-244 text format (returned as a list of lines)."""
    pred_bbox = torch.clamp(pred_bbox, min=0.)
    lines = []
    for i in range(pred_bbox.size(0)):
        b = pred_bbox[i]
        lines.append('%f,%f,%f,%f,%.4f,%d,-1,-1\n' % (float(b[0]), float(b[1]), float(b[2]), float(b[3]),
                                                     float(b[4]), int(b[5])))
    return lines
