"""Diagnostic (builder tool): gradient w.r.t. the backbone features, oracle fp64 / fp32 vs the HIP path."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rrnet_amd import ops, functional as RF
from oracle import ops as oo, model as om
from test_configs_gpu import _cfg, _matched_batch
from rrnet_amd.datasets.synthetic import synth_batch
from rrnet_amd.models.rrnet import RRNet
CL = torch.channels_last
k = 100

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())

torch.manual_seed(219)
model = RRNet(_cfg("hourglass"))
for i in range(2):
    model.wh.detect_H_layer[i][0].conv.bias.data.fill_(3.0)
    model.wh.detect_W_layer[i][0].conv.bias.data.fill_(3.0)
sd0 = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
batch = _matched_batch(sd0, synth_batch(2, 256, 256, boxes_per_image=4, seed=219)[0], k)
model = model.cuda().to(memory_format=CL).train()
wkeys = ["backbone.pre_layer.1.weight", "backbone.convs.1.conv.weight", "backbone.convs.1.bn.weight", "offset_reg.detect_layer.1.0.conv.weight"]

def oracle(dtype, wts):
    sd = {kk: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for kk, v in sd0.items()}
    for kk in wkeys: sd[kk].requires_grad_()
    imgs, annos, hms, whs, inds, offs, masks = [t.to(dtype) for t in batch]
    P = om.Params(sd, True)
    feats = om.hourglass_net(P, imgs)
    for f in feats: f.retain_grad()
    hms_, whs_, offs_ = om.stage1(P, feats)
    for t in offs_ + hms_: t.retain_grad()
    L = [sum(oo.hm_loss_from_logits(hms_[i], hms) / 2 for i in range(2)),
         sum(oo.reg_l1_loss(whs_[i], masks, inds, whs) / 2 for i in range(2)),
         sum(oo.reg_l1_loss(offs_[i], masks, inds, offs) / 2 for i in range(2))]
    sum(w * l for w, l in zip(wts, L)).backward()
    return dict(feat0=feats[0].grad, feat1=feats[1].grad, off1=offs_[1].grad, hm1=hms_[1].grad, f1=feats[1].detach(),
                **{kk: sd[kk].grad for kk in wkeys})

def mine(wts):
    model.zero_grad()
    imgs, annos, hms, whs, inds, offs, masks = [t.cuda() for t in batch]
    feats = model.backbone(imgs)
    for f in feats: f.retain_grad()
    hms_, whs_, offs_ = model.forward_stage1(list(feats))
    for t in offs_ + hms_: t.retain_grad()
    L = [sum(RF.focal_loss_hm_from_logits(hms_[i], hms) / 2 for i in range(2)),
         sum(RF.reg_l1_loss(whs_[i], masks, inds, whs) / 2 for i in range(2)),
         sum(RF.reg_l1_loss(offs_[i], masks, inds, offs) / 2 for i in range(2))]
    sum(w * l for w, l in zip(wts, L)).backward()
    named = dict(model.named_parameters())
    return dict(feat0=feats[0].grad, feat1=feats[1].grad, off1=offs_[1].grad, hm1=hms_[1].grad, f1=feats[1].detach(),
                **{kk: named[kk].grad for kk in wkeys})

for name, wts in (("off only", (0, 0, 1)), ("hm only", (1, 0, 0))):
    t = oracle(torch.float64, wts); r = oracle(torch.float32, wts); m = mine(wts)
    for key in t:
        if t[key] is None or t[key].abs().max() == 0: continue
        print("%-9s %-42s ref(max,l2)=%.1e,%.1e  mine=%.1e,%.1e" % ((name, key) + rel(r[key], t[key]) + rel(m[key], t[key])))
