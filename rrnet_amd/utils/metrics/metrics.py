"""VisDrone-DET AP / AR evaluation — the reference's utils/metrics/metrics.py (bbox_iou :10-49, get_tp :52-131,
calculate_ap_rc :134-176, evaluate_once :179-207, evaluate_results :210-253, auto_evaluate_results :256-306,
_ext_nms :309-323) with the same function names, arguments and printed report.

Host logic (file parsing, greedy matching, precision/recall integration) stays on the host as in the reference; it
is pinned against goldens produced by the reference's own get_tp / calculate_ap_rc / evaluate_once
(tests/golden/metrics.npz).  The part that dominates a threshold sweep — per-file, per-class Soft-NMS, called
files x classes times per (ctnet, softnms) threshold pair — runs as ONE batched launch of the bit-exact HIP
Soft-NMS over every (file, class) segment (`ext_nms_batch`).

Differences from the reference, all in code that cannot run as written under numpy >= 1.24 / torch 2:
`np.int` / `np.float` (:234,:287) are spelled int64 / float64; the reference's `_ext_nms` (:309-323) has no return
statement and `auto_evaluate_results` mixes tensors and arrays (:286-289) — restated with the evident intent
(xywh float32 array back); evaluate_results / auto_evaluate_results also RETURN (ap, rc) besides printing.
Quirk kept: detections of a class are dropped (not counted as false positives) in images that hold no ground
truth of that class (:112-113)."""
import glob
import os
import time

import numpy as np
import torch

THRESHOLDS = torch.arange(0.5, 1.0, 0.05)


def bbox_iou(a, b, x1y1x2y2=True, overlap=False):
    """IoU [m,n] between box sets a [m,4], b [n,4]; with overlap=True also intersection / area(a)."""
    assert isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)
    a, b = a.clone().float(), b.clone().float()
    if not x1y1x2y2:
        a[:, 2:4] += a[:, 0:2]
        b[:, 2:4] += b[:, 0:2]
    a_area = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    b_area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = (torch.min(a[:, 2:3], b[:, 2]) - torch.max(a[:, 0:1], b[:, 0])).clamp(min=0)
    ih = (torch.min(a[:, 3:4], b[:, 3]) - torch.max(a[:, 1:2], b[:, 1])).clamp(min=0)
    inter = iw * ih
    union = (a_area.unsqueeze(1) + b_area - inter).clamp(min=1e-8)
    iou = inter / union
    if overlap:
        return iou, inter / a_area.unsqueeze(1)
    return iou


def _greedy_match(tp_iou):
    """tp_iou [D,G,T]: IoU of detection d (score order) with same-class ground truth g where it clears threshold
    t, else 0.  Every (g, t) is given to the first detection whose best remaining match it is -> flags [D,T]."""
    d_n, g_n, t_n = tp_iou.shape
    work = tp_iou.clone()
    flags = torch.zeros(d_n, t_n)
    cols = torch.arange(t_n)
    for d in range(d_n):
        best, arg = work[d].max(dim=0)
        hit = best != 0
        if hit.any():
            work[:, arg[hit], cols[hit]] = 0
            flags[d, hit] = 1
    return flags


def get_tp(pred, target, cls_tp_flags, cls_tp_confs, cls_target_count, cls_in_img_count,
           thresholds=THRESHOLDS, cls_num=11):
    """One image: pred [m,6] = x,y,w,h,score,cls; target [n,>=6] VisDrone rows (cls 0 = ignored region).
    Appends per-class true-positive flags [d,T] / confidences and bumps the ground-truth counters."""
    order = torch.sort(pred[:, 4], descending=True)[1]
    pred = pred[order, :]
    # ground truth mostly inside an ignored region is removed (the regions themselves stay for the next step)
    ignore = target[:, 5] == 0
    if ignore.sum() != 0:
        _, gt_ov = bbox_iou(target[:, :4], target[:, :4], x1y1x2y2=False, overlap=True)
        keep = (gt_ov[:, ignore].max(dim=1)[0] < 0.5) | ignore
        target = target[keep, :]
    ignore = target[:, 5] == 0
    iou, ov = bbox_iou(pred[:, :4], target[:, :4], x1y1x2y2=False, overlap=True)
    if ignore.sum() != 0:
        keep = ov[:, ignore].max(dim=1)[0] < 0.5
        pred = pred[keep, :]
        iou = iou[keep, :]
    pred_cls = pred[:, 5].long()
    target_cls = target[:, 5].long()
    thr = thresholds.to(iou.dtype)
    for cls in range(1, cls_num):
        g_sel = target_cls == cls
        n_gt = int(g_sel.sum())
        cls_target_count[cls - 1] += n_gt
        cls_in_img_count[cls - 1] += 1 if n_gt != 0 else 0
        d_sel = pred_cls == cls
        if n_gt == 0 or int(d_sel.sum()) == 0:
            continue
        sub = iou[d_sel][:, g_sel]                                       # [D,G]
        clears = (sub.unsqueeze(2) - thr) >= 0
        flags = _greedy_match(sub.unsqueeze(2) * clears.float())
        cls_tp_flags[cls - 1] = torch.cat((cls_tp_flags[cls - 1], flags))
        cls_tp_confs[cls - 1] = torch.cat((cls_tp_confs[cls - 1], pred[d_sel, 4]))
    return cls_tp_flags, cls_tp_confs, cls_target_count, cls_in_img_count


def calculate_ap_rc(cls_tp_flags, cls_tp_confs, cls_target_count, cls_in_img_count):
    """-> AP per IoU threshold [T] (classes weighted by the number of images they occur in), mean max recall."""
    cls_num = cls_target_count.size(0)
    t_n = cls_tp_flags[0].size(1)
    total_ap = torch.zeros(t_n)
    total_rc = torch.zeros(t_n)
    for cls in range(cls_num):
        if cls_target_count[cls] == 0:
            continue
        order = torch.sort(cls_tp_confs[cls], descending=True)[1]
        tp_cum = cls_tp_flags[cls][order, :].cumsum(dim=0)
        rank = torch.arange(1., tp_cum.size(0) + 1, step=1.).unsqueeze(1)
        prec = tp_cum / rank
        rec = tp_cum / cls_target_count[cls].clamp(min=1)
        mrec = torch.cat((torch.zeros(1, t_n), rec, torch.ones(1, t_n)))
        mpre = torch.cat((torch.zeros(1, t_n), prec, torch.zeros(1, t_n)))
        mpre = torch.flip(torch.cummax(torch.flip(mpre, [0]), dim=0)[0], [0])     # precision envelope
        step = ((mrec[1:] - mrec[:-1]) > 0).float()
        total_ap += torch.sum((mrec[1:] * step - mrec[:-1] * step) * mpre[1:] * step, dim=0) * cls_in_img_count[cls]
        total_rc += mrec[:-1].max(dim=0)[0] * cls_in_img_count[cls]
    ap = total_ap / cls_in_img_count.sum()
    rc = (total_rc / cls_in_img_count.sum()).mean()
    return ap, rc


def _fresh(cls_num, t_n):
    return ([torch.zeros(0, t_n) for _ in range(1, cls_num)], [torch.zeros(0) for _ in range(1, cls_num)],
            torch.zeros(cls_num - 1), torch.zeros(cls_num - 1))


def evaluate_once(pred, target, thresholds=THRESHOLDS, cls_num=11, max_det_num=500):
    assert isinstance(pred, torch.Tensor) and isinstance(target, torch.Tensor)
    flags, confs, tc, ic = _fresh(cls_num, thresholds.size(0))
    flags, confs, tc, ic = get_tp(pred[:max_det_num], target, flags, confs, tc, ic, thresholds, cls_num)
    ap, rc = calculate_ap_rc(flags, confs, tc, ic)
    print(ap)
    return ap, rc


def _read(path):
    import pandas as pd
    return np.array(pd.read_csv(path, header=None, float_precision='high'))


def _names(pred_dir):
    return [os.path.splitext(os.path.basename(x))[0] for x in glob.glob(os.path.join(pred_dir, '*.txt'))]


def _snap(pred):
    """xywh -> integer corner coordinates -> xywh (:232-235)."""
    pred[:, 2:4] += pred[:, 0:2]
    pred[:, :4] = pred[:, :4].astype(np.int64).astype(np.float64)
    pred[:, 2:4] -= pred[:, 0:2]
    return pred


def _report(ap, rc, st):
    print("Average Precision  (AP) @[ IoU=0.50:0.95] = {:.4}.".format(ap.mean().item()))
    print("Average Precision  (AP) @[ IoU=0.50     ] = {:.4}.".format(ap[0].item()))
    print("Average Precision  (AP) @[ IoU=0.75     ] = {:.4}.".format(ap[5].item()))
    print("Average Recall     (AR) @[ IoU=0.50:0.95] = {:.4}.".format(rc.item()))
    print("Cost Time: {}s".format(time.time() - st))


def evaluate_results(pred_dir, target_dir, thresholds=THRESHOLDS, cls_num=11, max_det_num=500):
    st = time.time()
    flags, confs, tc, ic = _fresh(cls_num, thresholds.size(0))
    for name in _names(pred_dir):
        pred = _snap(_read(os.path.join(pred_dir, "{}.txt".format(name))).astype(np.float64))
        pred = torch.from_numpy(pred).float()[:max_det_num]
        target = torch.from_numpy(_read(os.path.join(target_dir, "{}.txt".format(name)))).float()[:max_det_num]
        flags, confs, tc, ic = get_tp(pred, target, flags, confs, tc, ic, thresholds, cls_num)
    ap, rc = calculate_ap_rc(flags, confs, tc, ic)
    _report(ap, rc, st)
    return ap, rc


def ext_nms_batch(preds, threshold, max_classes=32):
    """Per-class gaussian Soft-NMS (Nt 0.7) of MANY detection sets in one launch.  preds: list of float32 [n_i,6]
    xywh,score,cls arrays / tensors -> list of float32 numpy [n_i',6] xywh arrays, each in the order
    `np.concatenate` over ascending classes gives (:313-323)."""
    from rrnet_amd import ops
    from rrnet_amd.ext.nms.nms_wrapper import soft_nms_segments
    nf = len(preds)
    kmax = max((int(p.shape[0]) for p in preds), default=0)
    if kmax == 0:
        return [np.zeros((0, 6), np.float32) for _ in preds]
    host = np.full((nf, kmax, 6), -1.0, np.float32)          # class -1 rows are padding: grouping drops them
    for i, p in enumerate(preds):
        p = p.detach().cpu().numpy() if torch.is_tensor(p) else np.asarray(p)
        host[i, :p.shape[0]] = p[:, :6]
        cls = p[:, 5].astype(np.int64)
        assert cls.size == 0 or (cls.min() >= 0 and cls.max() < max_classes), "class id outside [0, %d)" % max_classes
    dev = torch.device("cuda", torch.cuda.current_device())
    b = torch.from_numpy(host).to(dev)
    b[:, :, 2:4] += b[:, :, 0:2]
    grouped, seg_off, seg_len = ops.group_by_class(b, max_classes)   # explicit lengths: the padding leaves gaps
    rows = grouped.view(-1, 6)
    n_out = soft_nms_segments(rows, seg_off, kmax, sigma=0.5, Nt=0.7, threshold=threshold, method=2, seg_len=seg_len)
    _, _, _, kept, out_off = ops.pack_segments(rows, seg_off, n_out, max_classes, want_rois=False, want_rows=True,
                                               want_offsets=True)
    kept[:, 2:4] -= kept[:, 0:2]
    kept = kept.cpu().numpy()
    fo = out_off.cpu().numpy()[::max_classes]
    return [kept[fo[i]:fo[i + 1]] for i in range(nf)]


def _ext_nms(pred_bbox, threshold):
    """One detection set (:309-323)."""
    if pred_bbox.shape[0] == 0:
        return pred_bbox
    return ext_nms_batch([pred_bbox], threshold)[0]


def auto_evaluate_results(pred_dir, target_dir, ctnet_min_threshold, softnms_min_threshold, thresholds=THRESHOLDS,
                          cls_num=11, max_det_num=500):
    st = time.time()
    names = _names(pred_dir)
    preds, targets = [], []
    for name in names:
        pred = _read(os.path.join(pred_dir, "{}.txt".format(name)))
        pred = torch.from_numpy(pred[pred[:, 4] > ctnet_min_threshold]).float()
        preds.append(pred[torch.sort(pred[:, 4], descending=True)[1]])
        targets.append(_read(os.path.join(target_dir, "{}.txt".format(name))))
    kept = ext_nms_batch(preds, softnms_min_threshold)        # every file and class in one launch
    flags, confs, tc, ic = _fresh(cls_num, thresholds.size(0))
    for pred, target in zip(kept, targets):
        pred = torch.from_numpy(_snap(pred.astype(np.float64))).float()
        pred = pred[torch.sort(pred[:, 4], descending=True)[1]][:max_det_num]
        target = torch.from_numpy(target).float()[:max_det_num]
        flags, confs, tc, ic = get_tp(pred, target, flags, confs, tc, ic, thresholds, cls_num)
    ap, rc = calculate_ap_rc(flags, confs, tc, ic)
    _report(ap, rc, st)
    return ap, rc
