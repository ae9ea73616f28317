"""Diagnostic (builder tool): stage-1 heads forward/backward (shared gradient accumulator) vs an fp64 host
computation, once with the host's own ReLU masks and once with the masks of the HIP forward (a pre-activation within
rounding of zero may land on either side: a 'flip' changes the gradient discretely without being an error)."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rrnet_amd import ops, functional as RF
from oracle import model as om
from test_configs_gpu import _cfg
from rrnet_amd.models.rrnet import RRNet
CL = torch.channels_last
torch.manual_seed(3)
model = RRNet(_cfg("hourglass_tiny"))
sd = {k: v.detach().clone().double() if v.is_floating_point() else v.clone() for k, v in model.state_dict().items()}
model = model.cuda().to(memory_format=CL).train()
g = torch.Generator().manual_seed(1)


def host_heads(P, feats, masks=None):
    """om.stage1 with optional forced masks: masks[(head, i)] and masks[('feat', i)] boolean tensors."""
    outs = ([], [], [])
    for i, f in enumerate(feats):
        fr = f * masks[("feat", i)] if masks else torch.relu(f)
        for h, (name, kind) in enumerate((("hm", "ct"), ("wh", "wh"), ("offset_reg", "ct"))):
            key = ("%s.detect_layer.%d.0.conv" if kind == "ct" else "%s.detect_conv_layer.%d.0.conv") % (name, i)
            hid = om.conv(P, key, fr, 1, 1)
            act = hid * masks[(name, i)] if masks else torch.relu(hid)
            if kind == "ct":
                o = om.conv(P, "%s.detect_layer.%d.1" % (name, i), act)
            else:
                hh = om.conv(P, "%s.detect_H_layer.%d.0.conv" % (name, i), act, 1, (8, 0))
                ww = om.conv(P, "%s.detect_W_layer.%d.0.conv" % (name, i), act, 1, (0, 8))
                o = torch.cat((ww, hh), 1)
            outs[h].append(o)
    return outs


for hw in (32, 64, 96):
    feats = [torch.randn(2, 256, hw, hw, generator=g) for _ in range(2)]
    dset = [torch.randn(2, c, hw, hw, generator=g) for c in (10, 2, 2)]
    # mine
    fm = [f.clone().cuda().contiguous(memory_format=CL).requires_grad_() for f in feats]
    mh, mw, mo = model.forward_stage1(fm)
    loss = sum((o * d.cuda()).sum() for i in range(2) for o, d in zip((mh[i], mw[i], mo[i]), dset))
    loss.backward()
    masks = {}
    with torch.no_grad():
        for i in range(2):
            masks[("feat", i)] = (feats[i] > 0).double()
            x = RF.relu(fm[i].detach())
            masks[("hm", i)] = (model.hm.detect_layer[i][0](x) > 0).cpu().double()
            masks[("wh", i)] = (model.wh.detect_conv_layer[i][0](x) > 0).cpu().double()
            masks[("offset_reg", i)] = (model.offset_reg.detect_layer[i][0](x) > 0).cpu().double()
    for label, mk in (("host masks", None), ("HIP masks ", masks)):
        fo = [f.clone().double().requires_grad_() for f in feats]
        P = om.Params(sd, True)
        hms, whs, offs = host_heads(P, fo, mk)
        loss = sum((o * d.double()).sum() for i in range(2) for o, d in zip((hms[i], whs[i], offs[i]), dset))
        loss.backward()
        for i in range(2):
            a, b = fm[i].grad.cpu().double(), fo[i].grad
            print("hw %d %s stack %d: d feat max %.2e l2 %.2e" % (
                hw, label, i, float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())))
