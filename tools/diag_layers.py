"""Diagnostic (builder tool): every distinct conv / BN kernel call of one hourglass-104 train step at a small
frame size is re-computed with torch on the host and compared."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.nn.functional as F
from rrnet_amd import ops, functional as RF

CL = torch.channels_last
seen = {}
bad = []

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))

_fprop, _dgrad, _wgrad = ops.conv_fprop, ops.conv_dgrad, ops.conv_wgrad
_bn_apply, _bn_bwd_reduce, _bn_bwd_apply = ops.bn_apply, ops.bn_bwd_reduce, ops.bn_bwd_apply

def note(kind, sig, err, tol=2e-4):
    key = (kind,) + sig
    if key in seen and err <= seen[key]:
        return
    seen[key] = err
    if err > tol:
        bad.append((key, err))
        print("MISMATCH", key, "%.2e" % err, flush=True)

def fprop(x, w, bias=None, stride=1, pad=(0, 0), relu=False, want_stats=False):
    out = _fprop(x, w, bias, stride, pad, relu, want_stats)
    y = out[0] if want_stats else out
    sig = (tuple(x.shape), tuple(w.shape), stride, tuple(pad), bias is not None, relu, want_stats)
    if ("fprop",) + sig not in seen:
        ref = F.conv2d(x.cpu().double(), w.cpu().double(), None if bias is None else bias.cpu().double(), stride, pad)
        if relu: ref = ref.relu()
        note("fprop", sig, rel(y, ref))
        if want_stats:
            k = w.shape[0]
            sums = ops.bn_reduce_slab(out[1], k)
            r0 = ref.sum((0, 2, 3)); r1 = (ref * ref).sum((0, 2, 3))
            note("fprop_stats", sig, max(rel(sums[:k], r0) if r0.abs().max() > 1e-6 * ref.numel() ** .5 else 0, rel(sums[k:2 * k], r1)))
    return out

def dgrad(dy, w, x_shape, stride=1, pad=(0, 0), out=None, accumulate=False):
    base = out.clone() if (out is not None and accumulate) else None
    res = _dgrad(dy, w, x_shape, stride, pad, out, accumulate)
    sig = (tuple(dy.shape), tuple(w.shape), tuple(x_shape), stride, tuple(pad), bool(accumulate))
    if ("dgrad",) + sig not in seen:
        n, c, h, wd = x_shape
        ref = torch.nn.grad.conv2d_input((n, c, h, wd), w.cpu().double(), dy.cpu().double(), stride, pad)
        if base is not None: ref = ref + base.cpu().double()
        note("dgrad", sig, rel(res, ref))
    return res

def wgrad(x, dy, dw, stride=1, pad=(0, 0), explicit_out=False):
    base = dw.clone()
    res = _wgrad(x, dy, dw, stride, pad, explicit_out)
    sig = (tuple(x.shape), tuple(dy.shape), tuple(dw.shape), stride, tuple(pad), explicit_out)
    if ("wgrad",) + sig not in seen and not explicit_out:
        ref = torch.nn.grad.conv2d_weight(x.cpu().double(), tuple(dw.shape), dy.cpu().double(), stride, pad)
        note("wgrad", sig, rel(res.cpu().double() - base.cpu().double(), ref), tol=1e-3)
    return res

ops.conv_fprop, ops.conv_dgrad, ops.conv_wgrad = fprop, dgrad, wgrad

def bn_apply(y, scale, shift, residual=None, relu=False, res_scale=None, res_shift=None):
    out = _bn_apply(y, scale, shift, residual, relu, res_scale, res_shift)
    sig = (tuple(y.shape), residual is not None, relu)
    if ("bn_apply",) + sig not in seen:
        ref = y.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        if residual is not None: ref = ref + residual.double()
        if relu: ref = ref.relu()
        note("bn_apply", sig, rel(out, ref))
    return out

def bn_bwd_reduce(dz, z, y, mean, invstd, extra=0, mask_scale=None, mask_shift=None):
    out = _bn_bwd_reduce(dz, z, y, mean, invstd, extra, mask_scale, mask_shift)
    sig = (tuple(y.shape), z is not None, mask_scale is not None)
    if ("bn_bwd_reduce",) + sig not in seen:
        d = dz.double()
        if z is not None: d = d * (z > 0)
        elif mask_scale is not None: d = d * ((y.double() * mask_scale.double().view(1, -1, 1, 1) + mask_shift.double().view(1, -1, 1, 1)) > 0)
        xh = (y.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
        c = y.shape[1]
        note("bn_bwd_reduce", sig, max(rel(out[:c], d.sum((0, 2, 3))), rel(out[c:2 * c], (d * xh).sum((0, 2, 3)))), tol=1e-3)
    return out

def bn_bwd_apply(dz, z, y, mean, invstd, gamma, sums, count, want_g=False, dgamma=None, dbeta=None, count_dev=None,
                 mask_scale=None, mask_shift=None):
    out = _bn_bwd_apply(dz, z, y, mean, invstd, gamma, sums, count, want_g, dgamma, dbeta, count_dev, mask_scale, mask_shift)
    sig = (tuple(y.shape), z is not None, mask_scale is not None, want_g)
    if ("bn_bwd_apply",) + sig not in seen:
        d = dz.double()
        if z is not None: d = d * (z > 0)
        elif mask_scale is not None: d = d * ((y.double() * mask_scale.double().view(1, -1, 1, 1) + mask_shift.double().view(1, -1, 1, 1)) > 0)
        V = lambda t: t.double().view(1, -1, 1, 1)
        c = y.shape[1]
        xh = (y.double() - V(mean)) * V(invstd)
        ref = V(gamma) * V(invstd) * (d - V(sums[:c]) / count - xh * V(sums[c:2 * c]) / count)
        e = rel(out[0], ref)
        if want_g: e = max(e, rel(out[1], d))
        note("bn_bwd_apply", sig, e)
    return out

ops.bn_apply, ops.bn_bwd_reduce, ops.bn_bwd_apply = bn_apply, bn_bwd_reduce, bn_bwd_apply

from test_configs_gpu import _cfg
from rrnet_amd.datasets.synthetic import synth_batch
from rrnet_amd.models.rrnet import RRNet
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(219)
model = RRNet(_cfg("hourglass")).cuda().to(memory_format=CL).train()
for i in range(2):
    model.wh.detect_H_layer[i][0].conv.bias.data.fill_(3.0)
    model.wh.detect_W_layer[i][0].conv.bias.data.fill_(3.0)
imgs, annos, hms, whs, inds, offs, masks, _ = [t.cuda() if torch.is_tensor(t) else t for t in synth_batch(2, size, size, boxes_per_image=12, seed=219)]
outs = model(imgs, k=100)
hm_l = sum(RF.focal_loss_hm_from_logits(outs[0][i], hms) / 2 for i in range(2))
wh_l = sum(RF.reg_l1_loss(outs[1][i], masks, inds, whs) / 2 for i in range(2))
off_l = sum(RF.reg_l1_loss(outs[2][i], masks, inds, offs) / 2 for i in range(2))
a = annos.clone(); a[:, :, 2:4] += a[:, :, 0:2]
s2_l = RF.stage2_reg_loss(outs[3], outs[4], a, 4.0)
(hm_l + 0.1 * wh_l + off_l + s2_l).backward()
torch.cuda.synchronize()
print("checked %d distinct kernel calls, %d mismatches" % (len(seen), len(bad)))
worst = sorted(seen.items(), key=lambda kv: -kv[1])[:12]
for k, v in worst:
    print("%.2e" % v, k)
