"""Diagnostic (builder tool): localise a gradient deviation of the full-depth model."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.nn.functional as F
from rrnet_amd import ops, functional as RF
from oracle import ops as oo, model as om

CL = torch.channels_last
dev = "cuda"
g = torch.Generator().manual_seed(0)

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max())

# A. stride-1 dgrad through the forward kernel, with and without accumulation
for (n, c, k, h) in ((2, 256, 256, 64), (2, 256, 256, 32), (2, 256, 36, 64), (2, 64, 256, 64)):
    dy = torch.randn(n, k, h, h, generator=g)
    w = torch.randn(k, c, 3, 3, generator=g) * 0.05
    base = torch.randn(n, c, h, h, generator=g)
    ref = F.conv_transpose2d(dy, w, padding=1)
    dyc, wc = ops.to_nhwc(dy.to(dev)), ops.to_nhwc(w.to(dev))
    out = ops.conv_dgrad(dyc, wc, (n, c, h, h), 1, (1, 1))
    acc = ops.to_nhwc(base.to(dev)).clone(memory_format=CL)
    ops.conv_dgrad(dyc, wc, (n, c, h, h), 1, (1, 1), out=acc, accumulate=True)
    print("A dgrad n%d c%d k%d h%d: plain %.2e  accumulate %.2e" % (n, c, k, h, rel(out, ref), rel(acc, ref + base)))

# B. roi_align backward vs oracle autograd
feat = torch.randn(2, 256, 64, 64, generator=g)
R = 200
xy = torch.rand(R, 2, generator=g) * 58
wh = 1.0 + torch.rand(R, 2, generator=g) * 5
rois = torch.cat((torch.randint(0, 2, (R, 1), generator=g).float(), xy, xy + wh), 1)
dout = torch.randn(R, 256, 3, 3, generator=g)
f = feat.clone().requires_grad_()
o = oo.roi_align(f, rois, (3, 3))
o.backward(dout)
got_f = ops.roi_align_fwd(ops.to_nhwc(feat.to(dev)), rois.to(dev), (3, 3))
got = ops.roi_align_bwd(ops.to_nhwc(dout.to(dev)), rois.to(dev), (2, 256, 64, 64), (3, 3))
print("B roi_align fwd %.2e bwd %.2e" % (rel(got_f, o), rel(got, f.grad)))

# C. whole model: which loss term carries the deviation
from test_configs_gpu import _cfg, _matched_batch, _HG104_GRAD_KEYS
from rrnet_amd.datasets.synthetic import synth_batch
from rrnet_amd.models.rrnet import RRNet
k = 100
keys = ["backbone.hgs.1.up1.1.conv2.weight", "backbone.convs.1.conv.weight", "backbone.hgs.1.low3.1.conv2.weight",
        "hm.detect_layer.1.0.conv.weight", "wh.detect_conv_layer.1.0.conv.weight", "offset_reg.detect_layer.1.0.conv.weight",
        "head_detector.top_layer.conv1.weight"]
torch.manual_seed(219)
model = RRNet(_cfg("hourglass"))
for i in range(2):
    model.wh.detect_H_layer[i][0].conv.bias.data.fill_(3.0)
    model.wh.detect_W_layer[i][0].conv.bias.data.fill_(3.0)
sd0 = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
batch = _matched_batch(sd0, synth_batch(2, 256, 256, boxes_per_image=4, seed=219)[0], k)
model = model.cuda().to(memory_format=CL).train()

def oracle(weights):
    sd = {kk: (v.clone().double() if v.is_floating_point() else v.clone()) for kk, v in sd0.items()}
    for kk in keys: sd[kk].requires_grad_()
    imgs, annos, hms, whs, inds, offs, masks = [t.double() for t in batch]
    P = om.Params(sd, True)
    outs = om.rrnet_forward(P, imgs, k=k)
    L = oo.criterion(outs, (hms, whs, inds, offs, masks, annos.clone()))
    sum(w * l for w, l in zip(weights, L)).backward()
    return {kk: sd[kk].grad for kk in keys}

def mine(weights):
    model.zero_grad()
    imgs, annos, hms, whs, inds, offs, masks = [t.cuda() for t in batch]
    outs = model(imgs, k=k)
    hm_l = sum(RF.focal_loss_hm_from_logits(outs[0][i], hms) / 2 for i in range(2))
    wh_l = sum(RF.reg_l1_loss(outs[1][i], masks, inds, whs) / 2 for i in range(2))
    off_l = sum(RF.reg_l1_loss(outs[2][i], masks, inds, offs) / 2 for i in range(2))
    a = annos.clone(); a[:, :, 2:4] += a[:, :, 0:2]
    s2_l = RF.stage2_reg_loss(outs[3], outs[4], a, 4.0)
    sum(w * l for w, l in zip(weights, (hm_l, wh_l, off_l, s2_l))).backward()
    return {kk: dict(model.named_parameters())[kk].grad for kk in keys}

for name, wts in (("hm only", (1, 0, 0, 0)), ("wh only", (0, 1, 0, 0)), ("off only", (0, 0, 1, 0)), ("s2 only", (0, 0, 0, 1))):
    t, m = oracle(wts), mine(wts)
    print("C", name, " ".join("%s=%.1e" % (kk.split("backbone.")[-1][:22], rel(m[kk], t[kk])) if t[kk].abs().max() > 0 else "%s=zero" % kk[:10] for kk in keys))
