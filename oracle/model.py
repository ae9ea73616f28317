"""ORACLE (test infrastructure only) — functional torch-CPU restatement of the reference's
network graph, written over a *state_dict with the reference's key names* so that the same
weights drive the oracle and the HIP path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this file.

Restates:
  /root/reference/backbones/hourglass.py:12-40   ResidualBlock
  /root/reference/backbones/hourglass.py:43-61   ConvBNRelu
  /root/reference/backbones/hourglass.py:64-124  Hourglass (recursive; max1 is identity,
                                                 down-sampling = stride-2 first block of low1)
  /root/reference/backbones/hourglass.py:127-199 HourglassNet
  /root/reference/detectors/centernet_detector.py:6-93   CenterNetDetector / CenterNetWHDetector
  /root/reference/detectors/fasterrcnn_detector.py:6-18  FasterRCNNDetector
  /root/reference/backbones/resnet.py:17-53      Bottleneck
  /root/reference/models/rrnet.py:25-54,140-157  RRNet.forward / forward_stage1 / forward_stage2
  /root/reference/models/centernet.py:18-32      CenterNet.forward
torch (conv / batch_norm / interpolate / topk) is the de-facto spec of those layers: torch-CPU
fp32 is the oracle, tolerance 1e-3 (north_star).
"""
import re

import torch
import torch.nn.functional as F

from oracle import ops

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Params:
    """A reference-keyed state_dict plus the train/eval switch.  In training mode batch_norm
    uses batch statistics and updates running_mean/var/num_batches_tracked in place, like
    nn.BatchNorm2d."""

    def __init__(self, sd, training=True, bf16=False):
        self.sd = sd
        self.training = training
        # builder-defined precision of BASELINE configs[3] (the reference is fp32-only): both operands of every
        # convolution rounded to bf16 (nearest even), products and sums in the tensors' own dtype (fp32 / fp64) —
        # the contract of rrnet_amd/csrc/conv_bf16.hip, restated so that a bf16 model run has an oracle
        self.bf16 = bf16
        self.dcn_bf16 = True       # cfg.Model.dcn_bf16 of the builder's config 4 (False: DCN products on unrounded operands)

    def has(self, key):
        return key in self.sd

    def count(self, prefix):
        """Number of children `prefix.<i>.` of a Sequential/ModuleList."""
        pat = re.compile(re.escape(prefix) + r"\.(\d+)\.")
        idx = {int(m.group(1)) for k in self.sd for m in [pat.match(k)] if m}
        return (max(idx) + 1) if idx else 0


def _bf16_round(t):
    return t.detach().to(torch.bfloat16).to(t.dtype)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class _ConvBf16(torch.autograd.Function):
    """The bf16 contract of the builder's convolution kernels (rrnet_amd/csrc/conv_bf16.hip, conv16.hip), forward AND
    backward: both operands of every matrix product rounded to bf16 (nearest even), products and sums in the tensors' own
    dtype — y = conv(round(x), round(w)); dx = conv_transpose(round(dy), round(w)); dw = correlation(round(x), round(dy)).
    (A plain `.to(bfloat16).to(dtype)` would instead round the finished GRADIENTS on their way back through the cast.)"""

    @staticmethod
    def forward(ctx, x, w, stride, padding):
        xq, wq = _bf16_round(x), _bf16_round(w)
        ctx.save_for_backward(xq, wq)
        ctx.cfg = (_pair(stride), _pair(padding))
        return F.conv2d(xq, wq, None, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, dy):
        xq, wq = ctx.saved_tensors
        stride, padding = ctx.cfg
        dq = _bf16_round(dy)
        dx = torch.nn.grad.conv2d_input(xq.shape, wq, dq, stride, padding) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(xq, wq.shape, dq, stride, padding) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


def conv_bf16(x, w, bias, stride, padding):
    y = _ConvBf16.apply(x, w, stride, padding)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def conv(P, key, x, stride=1, padding=0):
    w = P.sd[key + ".weight"]
    if P.bf16:
        return conv_bf16(x, w, P.sd.get(key + ".bias"), stride, padding)
    return F.conv2d(x, w, P.sd.get(key + ".bias"), stride=stride, padding=padding)


def bn(P, key, x):
    rm, rv = P.sd[key + ".running_mean"], P.sd[key + ".running_var"]
    y = F.batch_norm(x, rm, rv, P.sd[key + ".weight"], P.sd[key + ".bias"], P.training, BN_MOMENTUM, BN_EPS)
    if P.training and (key + ".num_batches_tracked") in P.sd:
        P.sd[key + ".num_batches_tracked"] += 1
    return y


def residual_block(P, p, x, stride=1):
    """hourglass.py:12-40."""
    out = F.relu(bn(P, p + ".bn1", conv(P, p + ".conv1", x, stride, 1)))
    out = bn(P, p + ".bn2", conv(P, p + ".conv2", out, 1, 1))
    if P.has(p + ".skip_connection.0.weight"):
        skip = bn(P, p + ".skip_connection.1", conv(P, p + ".skip_connection.0", x, stride, 0))
    else:
        skip = x
    return F.relu(out + skip)


def residual_seq(P, p, x, first_stride=1):
    for i in range(P.count(p)):
        x = residual_block(P, "%s.%d" % (p, i), x, first_stride if i == 0 else 1)
    return x


def hourglass(P, p, x):
    """hourglass.py:115-124.  low2 is a nested Hourglass iff it has an `up1` child."""
    up1 = residual_seq(P, p + ".up1", x)
    low1 = residual_seq(P, p + ".low1", x, first_stride=2)
    if P.has(p + ".low2.up1.0.conv1.weight"):
        low2 = hourglass(P, p + ".low2", low1)
    else:
        low2 = residual_seq(P, p + ".low2", low1)
    low3 = residual_seq(P, p + ".low3", low2)
    up2 = F.interpolate(low3, scale_factor=2)                                   # nn.Upsample(scale_factor=2): nearest
    up2 = F.interpolate(up2, size=(up1.size(2), up1.size(3)), mode='bilinear', align_corners=True)
    return up1 + up2


def hourglass_net(P, x, p="backbone"):
    """hourglass.py:179-199 -> list of num_stacks pre-ReLU feature maps."""
    pre = F.relu(bn(P, p + ".pre_layer.1", conv(P, p + ".pre_layer.0", x, 2, 3)))
    pre = residual_block(P, p + ".pre_layer.3", pre, stride=2)
    n_stacks = P.count(p + ".hgs")
    outs = []
    for i in range(n_stacks):
        feat = hourglass(P, "%s.hgs.%d" % (p, i), pre)
        feat = bn(P, "%s.convs.%d.bn" % (p, i), conv(P, "%s.convs.%d.conv" % (p, i), feat, 1, 1))
        outs.append(feat)
        feat = torch.relu(feat)
        if i < n_stacks - 1:
            a = bn(P, "%s.inter_.%d.1" % (p, i), conv(P, "%s.inter_.%d.0" % (p, i), pre))
            b = bn(P, "%s.conv_.%d.1" % (p, i), conv(P, "%s.conv_.%d.0" % (p, i), feat))
            pre = residual_block(P, "%s.residual.%d" % (p, i), F.relu(a + b))
    return outs


def head_conv3x3(P, key, x):
    """The heads' first layer: 3x3 conv(+bias), or — builder-defined DCN heads (BASELINE configs[3]) — ext/dcn's
    `DCN` in its place when the state_dict carries `<key>.conv_offset_mask.*` (oracle/dcn.py:dcn_forward)."""
    if P.has(key + ".conv_offset_mask.weight"):
        from oracle import dcn as odcn
        # (P.bf16: the offset / mask layer is an ordinary convolution of the bf16 model; the deformable products take the
        # same contract — oracle/dcn.py:_ContractBf16 — on the fp32 input the kernel samples)
        return odcn.dcn_forward(x, P.sd[key + ".weight"], P.sd[key + ".bias"], P.sd[key + ".conv_offset_mask.weight"],
                                P.sd[key + ".conv_offset_mask.bias"], 1, 1, 1, 1,
                                conv=conv_bf16 if P.bf16 else None, bf16=P.bf16 and P.dcn_bf16)
    return conv(P, key, x, 1, 1)


def ctdet_head(P, p, x, i):
    """centernet_detector.py:6-23: 3x3 conv(+bias)+ReLU -> 1x1 conv(+bias)."""
    q = "%s.detect_layer.%d" % (p, i)
    return conv(P, q + ".1", F.relu(head_conv3x3(P, q + ".0.conv", x)))


def wh_head(P, p, x, i):
    """centernet_detector.py:26-55: 3x3 conv+ReLU -> 17x1 (H) and 1x17 (W) convs, interleaved [W,H]."""
    c = F.relu(head_conv3x3(P, "%s.detect_conv_layer.%d.0.conv" % (p, i), x))
    h = conv(P, "%s.detect_H_layer.%d.0.conv" % (p, i), c, 1, (8, 0))
    w = conv(P, "%s.detect_W_layer.%d.0.conv" % (p, i), c, 1, (0, 8))
    h = h.view(h.size(0), -1, 1, h.size(2), h.size(3))
    w = w.view(w.size(0), -1, 1, w.size(2), w.size(3))
    return torch.cat((w, h), dim=2).view(h.size(0), -1, h.size(3), h.size(4))


def stage1(P, feats, off_name="offset_reg"):
    """rrnet.py:140-153 / centernet.py:18-32."""
    hms, whs, offs = [], [], []
    for i, f in enumerate(feats):
        f = torch.relu(f)
        hms.append(ctdet_head(P, "hm", f, i))
        whs.append(wh_head(P, "wh", f, i))
        offs.append(ctdet_head(P, off_name, f, i))
    return hms, whs, offs


def stage2_head(P, roi_feat, p="head_detector"):
    """fasterrcnn_detector.py:13-18 + resnet.py:33-53 (Bottleneck 256->64->64->256)."""
    t = p + ".top_layer"
    out = F.relu(bn(P, t + ".bn1", conv(P, t + ".conv1", roi_feat)))
    out = F.relu(bn(P, t + ".bn2", conv(P, t + ".conv2", out, 1, 1)))
    out = bn(P, t + ".bn3", conv(P, t + ".conv3", out))
    out = F.relu(out + roi_feat)
    out = F.adaptive_avg_pool2d(out, 1)
    reg = conv(P, p + ".regressor", out)
    return reg.view(reg.size(0), reg.size(1))


def rrnet_forward(P, x, k=1500, nms_type='nms', nms_per_class=True):
    """models/rrnet.py:25-54 -> (hms, whs, offsets, stage2_reg, bxyxys, scores, clses)."""
    feats = hourglass_net(P, x)
    hms, whs, offs = stage1(P, feats)
    bboxs = ops.transform_bbox(hms[-1], whs[-1], offs[-1], k)
    rois, scores, clses = [], [], []
    for b in range(bboxs.size(0)):
        kept = ops.stage1_nms(bboxs[b], nms_type, nms_per_class)
        scores.append(kept[:, 4])
        clses.append(kept[:, 5])
        rois.append(torch.cat((torch.ones((kept.size(0), 1)) * b, kept[:, :4]), dim=1))
    rois = torch.cat(rois, dim=0)
    scores = torch.cat(scores, dim=0)
    clses = torch.cat(clses, dim=0)
    roi_feat = ops.roi_align(torch.relu(feats[-1]), rois, (3, 3))
    reg = stage2_head(P, roi_feat)
    return hms, whs, offs, reg, rois, scores, clses


def centernet_forward(P, x):
    """models/centernet.py:18-32 (third head is named `reg` there)."""
    return stage1(P, hourglass_net(P, x), off_name="reg")
