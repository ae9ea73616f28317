"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (build container only; see
tools/ref_shims.py for the import recipe).  The fixtures hold data only: inputs and the
reference's outputs.  Weights are never stored: both sides rebuild them with
tests/helpers.det_fill (numpy PCG64 streams keyed by the state_dict key).

Goldens that pass through torchvision.ops.{roi_align,nms,box_iou} use the oracle's own
restatement of torchvision 0.3 (not installed here) and are flagged `tv_unpinned=1`.

  G2 decode.npz       RRNet.transform_bbox                       models/rrnet.py:93-138
  G3 losses.npz       focal_loss_for_hm / RegL1Loss (+grads)     modules/loss/*.py ; to_heatmap targets
  G4 blocks.npz       ResidualBlock / ConvBNRelu / Hourglass     backbones/hourglass.py
  G5 ctnet_tiny.npz   hourglass-tiny + 3 heads + CenterNetOperator.criterion (+grads)
  G6 stage2.npz       FasterRCNNDetector, generate_bbox(_target), criterion stage-2 branch
  G7 extnms.npz       RRNetOperator._ext_nms on a mixed-class set
  G8 rrnet_tiny.npz   RRNet.forward end-to-end on the tiny backbone (tv_unpinned)
  G11 helpers.npz     CenterNetOperator.transform_bbox / _ctnet_nms / save_result, RRNet._topk / gathers / nms
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from helpers import det_fill, shapes_of, synth_annos  # noqa: E402

from backbones.hourglass import ConvBNRelu, Hourglass, ResidualBlock  # noqa: E402  (reference)
from datasets.transforms.functional import to_heatmap  # noqa: E402
from detectors.centernet_detector import CenterNetDetector, CenterNetWHDetector  # noqa: E402
from detectors.fasterrcnn_detector import FasterRCNNDetector  # noqa: E402
from models.rrnet import RRNet  # noqa: E402
from modules.loss.focalloss import FocalLossHM  # noqa: E402
from modules.loss.regl1loss import RegL1Loss  # noqa: E402
from operators.centernet_operator import CenterNetOperator  # noqa: E402
from operators.rrnet_operator import RRNetOperator  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, d):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()})
    print("wrote %-18s %8d bytes" % (name, os.path.getsize(path)))


def load_det(module, seed):
    sd = det_fill(shapes_of(module.state_dict()), seed)
    module.load_state_dict(sd, strict=True)
    return module


# ------------------------------------------------------------------ builder-defined tiny backbone
TINY = dict(stem=16, n=2, inplanes=[32, 32, 48], layer_nums=[1, 1, 2], num_feats=256)


class TinyHourglassNet(nn.Module):
    """The reference's HourglassNet.__init__/forward (hourglass.py:127-199) with its hard-coded
    sizes replaced by arguments — composed from the REFERENCE's own Hourglass / ResidualBlock /
    ConvBNRelu classes, same attribute names, hence the same state_dict keys."""

    def __init__(self, num_stacks, stem, n, inplanes, layer_nums, num_feats):
        super().__init__()
        self.num_stacks = num_stacks
        self.pre_layer = nn.Sequential(
            nn.Conv2d(3, stem, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(stem),
            nn.ReLU(inplace=True), ResidualBlock(stem, 2 * stem, 2))
        self.hgs = nn.ModuleList([Hourglass(n, inplanes, layer_nums) for _ in range(num_stacks)])
        self.convs = nn.ModuleList([ConvBNRelu(3, inplanes[0], num_feats, with_relu=False) for _ in range(num_stacks)])
        self.residual = nn.ModuleList([ResidualBlock(inplanes[0], inplanes[0]) for _ in range(num_stacks - 1)])
        self.inter_ = nn.ModuleList([nn.Sequential(nn.Conv2d(inplanes[0], inplanes[0], (1, 1), bias=False),
                                                   nn.BatchNorm2d(inplanes[0])) for _ in range(num_stacks - 1)])
        self.conv_ = nn.ModuleList([nn.Sequential(nn.Conv2d(num_feats, inplanes[0], (1, 1), bias=False),
                                                  nn.BatchNorm2d(inplanes[0])) for _ in range(num_stacks - 1)])
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        pre_feat = self.pre_layer(x)
        outs = []
        for i in range(self.num_stacks):
            feat = self.convs[i](self.hgs[i](pre_feat))
            outs.append(feat)
            feat = torch.relu(feat)
            if i < self.num_stacks - 1:
                pre_feat = self.residual[i](self.relu(self.inter_[i](pre_feat) + self.conv_[i](feat)))
        return outs


def make_targets(rng, bs, img, n_obj):
    """Reference to_heatmap (functional.py:230-262) + collate_fn_ctnet (drones_det.py:70-94)."""
    per = []
    for b in range(bs):
        annos = torch.from_numpy(synth_annos(rng, n_obj[b], img, img))
        _, a, hm, wh, ind, off, mask = to_heatmap((torch.zeros(3, img, img), annos), scale_factor=4)
        per.append((a, hm, wh, ind, off, mask))
    m = max(p[0].size(0) for p in per)
    annos = torch.zeros(bs, m, 8); whs = torch.zeros(bs, m, 2); offs = torch.zeros(bs, m, 2)
    inds = torch.zeros(bs, m, 1); masks = torch.zeros(bs, m, 1)
    hms = []
    for b, (a, hm, wh, ind, off, mask) in enumerate(per):
        n = a.size(0)
        annos[b, :n] = a[:, :8]; whs[b, :n] = wh; inds[b, :n] = ind; offs[b, :n] = off; masks[b, :n] = mask
        hms.append(hm.unsqueeze(0))
    return annos, torch.cat(hms), whs, inds, offs, masks


# ------------------------------------------------------------------ G2
def g2_decode():
    rng = np.random.default_rng(2)
    b, c, h, w, k = 2, 10, 32, 32, 50
    vals = rng.permutation(b * c * h * w).astype(np.float32)          # tie-free
    hm = torch.from_numpy(((vals / vals.size) * 8 - 6).reshape(b, c, h, w).astype(np.float32))
    wh = torch.from_numpy(rng.normal(3, 3, (b, 2, h, w)).astype(np.float32))     # some negative -> clamp
    off = torch.from_numpy(rng.uniform(0, 1, (b, 2, h, w)).astype(np.float32))
    net = RRNet.__new__(RRNet)
    nn.Module.__init__(net)
    out = net.transform_bbox(hm, wh, off, k)
    save("decode.npz", dict(hm=hm, wh=wh, offset=off, k=k, out=out))


# ------------------------------------------------------------------ G3
def g3_losses():
    rng = np.random.default_rng(3)
    annos, hms, whs, inds, offs, masks = make_targets(rng, 2, 128, [9, 5])
    d = dict(annos=annos, gt_hm=hms, gt_wh=whs, gt_ind=inds, gt_off=offs, gt_mask=masks)
    logits = torch.from_numpy(rng.normal(-2.19, 1.5, (2, 10, 32, 32)).astype(np.float32)).requires_grad_()
    p = torch.clamp(torch.sigmoid(logits), min=1e-4, max=1 - 1e-4)
    loss = FocalLossHM()(p, hms)
    loss.backward()
    d.update(hm_logits=logits, focal=loss, focal_grad=logits.grad)
    # N_pos == 0 branch
    lg0 = torch.from_numpy(rng.normal(-2.19, 1.5, (1, 10, 16, 16)).astype(np.float32)).requires_grad_()
    gt0 = torch.from_numpy(rng.uniform(0, 0.9, (1, 10, 16, 16)).astype(np.float32))
    l0 = FocalLossHM()(torch.clamp(torch.sigmoid(lg0), min=1e-4, max=1 - 1e-4), gt0)
    l0.backward()
    d.update(hm0_logits=lg0, hm0_gt=gt0, focal0=l0, focal0_grad=lg0.grad)
    pred = torch.from_numpy(rng.normal(2, 2, (2, 2, 32, 32)).astype(np.float32)).requires_grad_()
    l1 = RegL1Loss()(pred, masks, inds, whs)
    l1.backward()
    d.update(reg_pred=pred, regl1=l1, regl1_grad=pred.grad)
    save("losses.npz", d)


# ------------------------------------------------------------------ G4
def run_block(mod, x, train):
    mod.train(train)
    x = x.clone().requires_grad_()
    y = mod(x)
    gy = torch.from_numpy(np.random.default_rng(44).normal(0, 1, tuple(y.shape)).astype(np.float32))
    y.backward(gy)
    grads = {k: p.grad.clone() for k, p in mod.named_parameters()}
    bufs = {k: b.clone() for k, b in mod.named_buffers()}
    return y, gy, x.grad, grads, bufs


def g4_blocks():
    rng = np.random.default_rng(4)
    d = {}
    specs = {
        "res_id": (lambda: ResidualBlock(16, 16, 1), (2, 16, 12, 10)),
        "res_s2": (lambda: ResidualBlock(16, 24, 2), (2, 16, 12, 10)),
        "res_s2_odd": (lambda: ResidualBlock(8, 12, 2), (2, 8, 13, 11)),
        "cbr": (lambda: ConvBNRelu(3, 16, 24, with_relu=False), (2, 16, 9, 9)),
        "hg": (lambda: Hourglass(2, [8, 8, 12], [1, 1, 2]), (2, 8, 16, 24)),
        "hg_odd": (lambda: Hourglass(2, [8, 8, 12], [1, 1, 2]), (1, 8, 37, 41)),   # bilinear resize path
    }
    for name, (ctor, shape) in specs.items():
        x = torch.from_numpy(rng.normal(0, 1, shape).astype(np.float32))
        for train in (True, False):
            mod = load_det(ctor(), seed=40)
            y, gy, gx, grads, bufs = run_block(mod, x, train)
            t = "%s/%s" % (name, "train" if train else "eval")
            d[t + "/x"] = x; d[t + "/y"] = y; d[t + "/gy"] = gy; d[t + "/gx"] = gx
            for k, g in grads.items():
                d[t + "/grad/" + k] = g
            for k, b in bufs.items():
                d[t + "/buf/" + k] = b
    save("blocks.npz", d)


# ------------------------------------------------------------------ G5
class TinyCenterNet(nn.Module):
    """models/centernet.py:8-32 with the tiny backbone (reference head classes)."""

    def __init__(self, num_stacks):
        super().__init__()
        self.num_stacks = num_stacks
        self.backbone = TinyHourglassNet(num_stacks, **TINY)
        self.hm = CenterNetDetector(planes=10, num_stacks=num_stacks, hm=True)
        self.wh = CenterNetWHDetector(planes=1, num_stacks=num_stacks)
        self.reg = CenterNetDetector(planes=2, num_stacks=num_stacks)

    def forward(self, x):
        feats = self.backbone(x)
        hms, whs, regs = [], [], []
        for i in range(self.num_stacks):
            f = torch.relu(feats[i])
            hms.append(self.hm(f, i)); whs.append(self.wh(f, i)); regs.append(self.reg(f, i))
        return hms, whs, regs


GRAD_KEYS_G5 = ["backbone.pre_layer.0.weight", "backbone.pre_layer.1.weight", "backbone.pre_layer.3.conv1.weight",
                "backbone.hgs.0.up1.0.conv1.weight", "backbone.hgs.0.low1.0.skip_connection.0.weight",
                "backbone.hgs.0.low2.low2.1.bn2.bias", "backbone.hgs.1.low3.0.conv2.weight",
                "backbone.inter_.0.0.weight", "backbone.conv_.0.1.weight",
                "hm.detect_layer.1.1.bias", "hm.detect_layer.0.1.weight",
                "wh.detect_H_layer.1.0.conv.weight", "wh.detect_W_layer.0.0.conv.weight", "reg.detect_layer.1.1.weight"]


def g5_ctnet_tiny():
    rng = np.random.default_rng(5)
    model = load_det(TinyCenterNet(2), seed=50).train()
    for i in range(2):
        model.hm.detect_layer[i][-1].bias.data.fill_(-2.19)
    x = torch.from_numpy(rng.normal(0, 1, (2, 3, 128, 128)).astype(np.float32))
    annos, hms, whs, inds, offs, masks = make_targets(rng, 2, 128, [12, 7])
    outs = model(x)
    ns = SimpleNamespace(cfg=SimpleNamespace(Model=SimpleNamespace(num_stacks=2)), focal_loss=FocalLossHM(),
                         l1_loss=RegL1Loss())
    hm_l, wh_l, off_l = CenterNetOperator.criterion(ns, outs, (hms, whs, inds, offs, masks))
    loss = hm_l + 0.1 * wh_l + off_l
    loss.backward()
    d = dict(x=x, annos=annos, gt_hm=hms, gt_wh=whs, gt_ind=inds, gt_off=offs, gt_mask=masks,
             hm_loss=hm_l, wh_loss=wh_l, off_loss=off_l)
    for i in range(2):
        d["hm%d" % i] = outs[0][i]; d["wh%d" % i] = outs[1][i]; d["reg%d" % i] = outs[2][i]
    named = dict(model.named_parameters())
    for k in GRAD_KEYS_G5:
        d["grad/" + k] = named[k].grad
    d["grad_l2"] = np.array([float(p.grad.norm()) for _, p in sorted(named.items())], dtype=np.float64)
    d["grad_keys"] = np.array(sorted(named))
    d["buf/backbone.pre_layer.1.running_mean"] = model.backbone.pre_layer[1].running_mean
    d["buf/backbone.pre_layer.1.running_var"] = model.backbone.pre_layer[1].running_var
    save("ctnet_tiny.npz", d)


# ------------------------------------------------------------------ G6
def g6_stage2():
    rng = np.random.default_rng(6)
    d = {}
    head = load_det(FasterRCNNDetector(), seed=60).train()
    roi_feat = torch.from_numpy(np.maximum(rng.normal(0, 1, (7, 256, 3, 3)), 0).astype(np.float32)).requires_grad_()
    reg = head(roi_feat)
    g = torch.from_numpy(rng.normal(0, 1, (7, 4)).astype(np.float32))
    reg.backward(g)
    d.update(roi_feat=roi_feat, reg=reg, reg_gy=g, roi_feat_grad=roi_feat.grad)
    for k, p in head.named_parameters():
        d["headgrad/" + k] = p.grad
    head.eval()
    d["reg_eval"] = head(roi_feat.detach())

    ex = torch.from_numpy(np.concatenate([rng.uniform(0, 100, (9, 2)), rng.uniform(110, 200, (9, 2))], 1).astype(np.float32))
    gt = torch.from_numpy(np.concatenate([rng.uniform(0, 100, (9, 2)), rng.uniform(110, 200, (9, 2))], 1).astype(np.float32))
    d.update(tgt_ex=ex, tgt_gt=gt, tgt_out=RRNetOperator.generate_bbox_target(ex, gt))

    # generate_bbox + stage-2 criterion on a synthetic `outs`; image 1 has no positive RoI
    bs, hf = 2, 32
    gt_annos = torch.zeros(bs, 4, 8)
    gt_annos[0, :, :4] = torch.tensor([[8., 8., 24., 20.], [60., 40., 30., 30.], [90., 90., 20., 12.], [10., 80., 16., 16.]])
    gt_annos[1, :, :4] = torch.tensor([[100., 100., 10., 10.], [5., 5., 6., 6.], [50., 90., 12., 8.], [70., 20., 9., 9.]])
    gt_annos[:, :, 4] = 1; gt_annos[:, :, 5] = 3
    rois0 = torch.tensor([[0, 2.1, 2.0, 7.9, 7.1], [0, 15.2, 10.1, 22.3, 17.4], [0, 1.0, 1.0, 3.0, 3.0],
                          [0, 22.4, 22.6, 27.6, 25.4], [0, 2.4, 19.8, 6.7, 24.1]])
    rois1 = torch.tensor([[1, 0.5, 10.0, 4.0, 14.0], [1, 20.0, 1.0, 24.0, 3.0], [1, 12.0, 12.0, 18.0, 30.0]])
    bxyxy = torch.cat([rois0, rois1])
    r = bxyxy.size(0)
    s2_reg = torch.from_numpy(rng.normal(0, 0.3, (r, 4)).astype(np.float32)).requires_grad_()
    scores = torch.from_numpy(rng.uniform(0.1, 1, r).astype(np.float32))
    clses = torch.from_numpy(rng.integers(0, 10, r).astype(np.float32))
    zeros = [torch.zeros(bs, 10, hf, hf, requires_grad=True) for _ in range(2)]
    z2 = [torch.zeros(bs, 2, hf, hf, requires_grad=True) for _ in range(4)]
    outs = (zeros, z2[:2], z2[2:], s2_reg, bxyxy, scores, clses)
    targets = (torch.zeros(bs, 10, hf, hf), torch.zeros(bs, 4, 2), torch.zeros(bs, 4, 1), torch.zeros(bs, 4, 2),
               torch.zeros(bs, 4, 1), gt_annos.clone())
    ns = SimpleNamespace(cfg=SimpleNamespace(Model=SimpleNamespace(num_stacks=2), Train=SimpleNamespace(scale_factor=4)),
                         hm_focal_loss=FocalLossHM(), l1_loss=RegL1Loss(),
                         generate_bbox_target=RRNetOperator.generate_bbox_target)
    _, _, _, s2_loss = RRNetOperator.criterion(ns, outs, targets)
    s2_loss.backward()
    d.update(crit_bxyxy=bxyxy, crit_s2_reg=s2_reg, crit_scores=scores, crit_clses=clses, crit_gt_annos=gt_annos,
             crit_s2_loss=s2_loss, crit_s2_reg_grad=s2_reg.grad, tv_unpinned=1)
    # the same criterion with the boxes attached to the graph (what the hard-NMS path of models/rrnet.py:69-70
    # hands over): F.smooth_l1_loss differentiates its target, so d loss / d boxes is non-zero
    bx = bxyxy.clone().requires_grad_()
    s2b = s2_reg.detach().clone().requires_grad_()
    outs_b = (zeros, z2[:2], z2[2:], s2b, bx, scores, clses)
    targets_b = (torch.zeros(bs, 10, hf, hf), torch.zeros(bs, 4, 2), torch.zeros(bs, 4, 1), torch.zeros(bs, 4, 2),
                 torch.zeros(bs, 4, 1), gt_annos.clone())
    _, _, _, s2_loss_b = RRNetOperator.criterion(ns, outs_b, targets_b)
    s2_loss_b.backward()
    d.update(crit_bxyxy_grad=bx.grad, crit_s2_reg_grad_attached=s2b.grad)
    with torch.no_grad():
        for b in range(bs):
            outs_b = (None, None, None, s2_reg.detach(), bxyxy.clone(), scores, clses)
            s1, s2 = RRNetOperator.generate_bbox(ns, outs_b, batch_idx=b)
            d["genbbox%d_s1" % b] = s1; d["genbbox%d_s2" % b] = s2
    save("stage2.npz", d)


# ------------------------------------------------------------------ G7
def g7_extnms():
    rng = np.random.default_rng(7)
    n = 60
    xy = rng.uniform(0, 160, (n, 2)); wh = rng.uniform(10, 70, (n, 2))
    s = rng.uniform(0.05, 1, (n, 1)); c = rng.integers(1, 5, (n, 1)).astype(np.float64)
    pred = torch.from_numpy(np.concatenate([xy, wh, s, c], 1).astype(np.float32))
    out_cls = RRNetOperator._ext_nms(pred.clone(), per_cls=True)
    out_all = RRNetOperator._ext_nms(pred.clone(), per_cls=False)
    save("extnms.npz", dict(pred=pred, out_per_cls=out_cls, out_all=out_all,
                            empty=RRNetOperator._ext_nms(torch.zeros(0, 6))))


# ------------------------------------------------------------------ G8
class TinyRRNet(RRNet):
    """models/rrnet.py:11-23 with the tiny backbone; forward/nms/decode are the reference's."""

    def __init__(self, nms_type):
        nn.Module.__init__(self)
        self.num_stacks = 2
        self.num_classes = 10
        self.nms_type = nms_type
        self.nms_per_class = True
        self.backbone = TinyHourglassNet(2, **TINY)
        self.hm = CenterNetDetector(planes=10, num_stacks=2, hm=True)
        self.wh = CenterNetWHDetector(planes=1, num_stacks=2)
        self.offset_reg = CenterNetDetector(planes=2, num_stacks=2)
        self.head_detector = FasterRCNNDetector()


def g8_rrnet_tiny():
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.normal(0, 1, (2, 3, 128, 128)).astype(np.float32))
    annos, hms, whs, inds, offs, masks = make_targets(rng, 2, 128, [10, 6])
    d = dict(x=x, annos=annos, gt_hm=hms, gt_wh=whs, gt_ind=inds, gt_off=offs, gt_mask=masks, tv_unpinned=1, k=60)
    for nms_type in ("nms", "soft_nms"):
        model = load_det(TinyRRNet(nms_type), seed=80).train()
        for i in range(2):
            model.hm.detect_layer[i][-1].bias.data.fill_(-2.19)
        # make wh positive-ish so that boxes overlap and the NMS has work to do
        for i in range(2):
            model.wh.detect_H_layer[i][0].conv.bias.data.fill_(3.0)
            model.wh.detect_W_layer[i][0].conv.bias.data.fill_(3.0)
        outs = model(x, k=60)
        ns = SimpleNamespace(cfg=SimpleNamespace(Model=SimpleNamespace(num_stacks=2), Train=SimpleNamespace(scale_factor=4)),
                             hm_focal_loss=FocalLossHM(), l1_loss=RegL1Loss(),
                             generate_bbox_target=RRNetOperator.generate_bbox_target)
        hm_l, wh_l, off_l, s2_l = RRNetOperator.criterion(ns, outs, (hms, whs, inds, offs, masks, annos.clone()))
        (hm_l + 0.1 * wh_l + off_l + s2_l).backward()
        t = nms_type
        d[t + "/hm1"] = outs[0][1]; d[t + "/wh1"] = outs[1][1]; d[t + "/off1"] = outs[2][1]
        d[t + "/s2_reg"] = outs[3]; d[t + "/bxyxy"] = outs[4]; d[t + "/scores"] = outs[5]; d[t + "/clses"] = outs[6]
        d[t + "/losses"] = torch.stack([hm_l, wh_l, off_l, torch.as_tensor(s2_l, dtype=torch.float32)])
        named = dict(model.named_parameters())
        d[t + "/grad_l2"] = np.array([float(p.grad.norm()) if p.grad is not None else -1.0
                                      for _, p in sorted(named.items())], dtype=np.float64)
        d[t + "/grad/head_detector.regressor.weight"] = named["head_detector.regressor.weight"].grad
        d[t + "/grad/backbone.pre_layer.0.weight"] = named["backbone.pre_layer.0.weight"].grad
        with torch.no_grad():
            s1b, s2b = RRNetOperator.generate_bbox(ns, tuple(o.detach().clone() if torch.is_tensor(o) else o for o in outs))
            d[t + "/s1_bboxes"] = s1b; d[t + "/s2_bboxes"] = s2b
            d[t + "/s2_after_extnms"] = RRNetOperator._ext_nms(s2b)
    d["grad_keys"] = np.array(sorted(named))
    save("rrnet_tiny.npz", d)


# ------------------------------------------------------------------ to_heatmap golden (host-side target contract)
def g9_targets():
    rng = np.random.default_rng(9)
    d = {}
    for i, (img, n) in enumerate(((128, 14), (256, 40))):
        annos = torch.from_numpy(synth_annos(rng, n, img, img, max_wh=img / 2))
        _, a, hm, wh, ind, off, mask = to_heatmap((torch.zeros(3, img, img), annos), scale_factor=4)
        d["c%d/annos" % i] = annos; d["c%d/hm" % i] = hm; d["c%d/wh" % i] = wh
        d["c%d/ind" % i] = ind; d["c%d/off" % i] = off; d["c%d/mask" % i] = mask.float()
        d["c%d/img" % i] = img
    save("targets.npz", d)


def _metric_case(rng, m, n, n_ignore, img=400):
    """Predictions [m,6] xywh,score,cls(1..10) scattered around the targets; targets [n,8] VisDrone rows with
    `n_ignore` ignored regions (cls 0)."""
    txy = rng.uniform(0, img - 60, (n, 2)); twh = rng.uniform(8, 60, (n, 2))
    tcls = rng.integers(1, 11, (n, 1)).astype(np.float64)
    tcls[:n_ignore] = 0
    target = np.concatenate([txy, twh, np.ones((n, 1)), tcls, np.zeros((n, 2))], 1).astype(np.float32)
    src = rng.integers(0, n, m)
    pxy = txy[src] + rng.normal(0, 4, (m, 2)); pwh = twh[src] * rng.uniform(0.7, 1.3, (m, 2))
    pcls = np.where(rng.uniform(size=(m, 1)) < 0.8, np.maximum(tcls[src], 1), rng.integers(1, 11, (m, 1)))
    pred = np.concatenate([pxy, pwh, rng.uniform(0.02, 1, (m, 1)), pcls], 1).astype(np.float32)
    return torch.from_numpy(pred), torch.from_numpy(target)


def g10_metrics():
    """utils/metrics/metrics.py: bbox_iou :10-49, get_tp :52-131, calculate_ap_rc :134-176, evaluate_once :179-207
    (evaluate_results / auto_evaluate_results use np.int / np.float, removed in numpy 2: not runnable here)."""
    import contextlib
    import io
    from utils.metrics import metrics as M
    rng = np.random.default_rng(10)
    d = {}
    thresholds = torch.arange(0.5, 1.0, 0.05)
    cases = [_metric_case(rng, 120, 40, 3), _metric_case(rng, 60, 25, 0), _metric_case(rng, 200, 70, 6),
             _metric_case(rng, 30, 12, 2)]
    tc, ic = torch.zeros(10), torch.zeros(10)
    flags = [torch.zeros(0, 10) for _ in range(10)]
    confs = [torch.zeros(0) for _ in range(10)]
    for i, (pred, target) in enumerate(cases):
        d["c%d/pred" % i], d["c%d/target" % i] = pred, target
        iou, ov = M.bbox_iou(pred[:, :4], target[:, :4], x1y1x2y2=False, overlap=True)
        d["c%d/iou" % i], d["c%d/overlap" % i] = iou, ov
        with contextlib.redirect_stdout(io.StringIO()):
            ap, rc = M.evaluate_once(pred.clone(), target.clone(), max_det_num=100 if i == 2 else 500)
        d["c%d/ap" % i], d["c%d/rc" % i] = ap, rc
        flags, confs, tc, ic = M.get_tp(pred.clone(), target.clone(), flags, confs, tc, ic, thresholds, 11)
        d["c%d/target_count" % i], d["c%d/in_img_count" % i] = tc.clone(), ic.clone()
    for c in range(10):
        d["all/flags%d" % c], d["all/confs%d" % c] = flags[c], confs[c]
    ap, rc = M.calculate_ap_rc(flags, confs, tc, ic)
    d["all/ap"], d["all/rc"] = ap, rc
    save("metrics.npz", d)


def g11_helpers():
    """CenterNetOperator.transform_bbox :152-178, _ctnet_nms :204-210, save_result :238-249;
    RRNet._topk / _gather_feat / _transpose_and_gather_feat / nms (models/rrnet.py:56-115, hard-NMS path through the
    oracle's torchvision restatement -> tv_unpinned)."""
    import tempfile
    rng = np.random.default_rng(11)
    b, c, h, w, k = 2, 10, 24, 40, 60
    vals = rng.permutation(b * c * h * w).astype(np.float32)          # tie-free
    hm = torch.from_numpy(((vals / vals.size) * 9 - 6).reshape(b, c, h, w).astype(np.float32))
    wh = torch.from_numpy(rng.normal(3, 3, (b, 2, h, w)).astype(np.float32))     # some negative: CenterNet keeps them
    off = torch.from_numpy(rng.uniform(0, 1, (b, 2, h, w)).astype(np.float32))
    op = CenterNetOperator.__new__(CenterNetOperator)
    d = dict(hm=hm, wh=wh, offset=off, k=k)
    d["ct_pred"] = op.transform_bbox(hm, wh, off, k=k, scale_factor=4)
    d["ct_pred_nooff"] = op.transform_bbox(hm, wh, None, k=k, scale_factor=4)
    # `_ctnet_nms` on a score map with plateaus and border maxima
    heat = torch.sigmoid(torch.from_numpy(rng.normal(-2, 2, (2, 4, 19, 23)).astype(np.float32)))
    heat[0, 1, 5:8, 5:8] = 0.75
    heat[1, 2, 0, 0:2] = 0.9
    heat[1, 3] = torch.round(heat[1, 3] * 8) / 8
    d["heat"], d["heat_nms"] = heat, op._ctnet_nms(heat)
    with tempfile.TemporaryDirectory() as td:
        rows = torch.from_numpy(np.concatenate([rng.uniform(-5, 300, (7, 4)), rng.uniform(0, 1, (7, 1)),
                                                rng.integers(1, 11, (7, 1))], 1).astype(np.float32))
        op.save_result(os.path.join(td, "r.txt"), rows.clone())
        d["save_rows"] = rows
        d["save_text"] = np.frombuffer(open(os.path.join(td, "r.txt"), "rb").read(), dtype=np.uint8)
    net = RRNet.__new__(RRNet)
    nn.Module.__init__(net)
    scores = torch.sigmoid(hm)
    ts, ti, tc, ty, tx = net._topk(scores, k)
    d["topk_score"], d["topk_inds"], d["topk_clses"], d["topk_ys"], d["topk_xs"] = ts, ti, tc, ty, tx
    d["tg_feat"] = net._transpose_and_gather_feat(wh, ti)
    net.nms_per_class, net.nms_type = True, 'nms'
    bbox = net.transform_bbox(hm, wh.clamp(min=1.0) * 3, off, k)[0]
    d["nms_in"], d["nms_out"] = bbox, net.nms(bbox)
    net.nms_type = 'soft_nms'
    d["softnms_out"] = net.nms(bbox)
    save("helpers.npz", d)


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, fn in (("g2", g2_decode), ("g3", g3_losses), ("g4", g4_blocks), ("g5", g5_ctnet_tiny),
                     ("g6", g6_stage2), ("g7", g7_extnms), ("g8", g8_rrnet_tiny), ("g9", g9_targets),
                     ("g10", g10_metrics), ("g11", g11_helpers)):
        if not only or name in only:
            fn()
