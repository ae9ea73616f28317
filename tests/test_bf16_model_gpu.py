"""GPU: BASELINE configs[3] as a bf16 MODEL (cfg.Model.bf16): every convolution of the backbone and the heads takes bf16
matrix operands with fp32 accumulation (csrc/conv_bf16.hip); activations, weights, BatchNorm, losses, optimizer stay fp32.

The reference is fp32-only (/root/reference/backbones/hourglass.py:12-61, ext/dcn/src/cuda/dcn_v2_cuda.cu:58), so the
precision is builder-defined and its parity statement has two parts:
  1. per kernel call, tight: every conv launch of a full-size train step equals an fp64 recomputation on bf16-ROUNDED
     operands from the inputs the call received (tests/kernel_audit.py, 2e-5 / 2e-4) — nothing else in the step changes;
  2. end to end, derived: the oracle (oracle/model.py) is run in fp64 twice, plain and with both operands of every
     convolution rounded to bf16.  Their distance E = max |H_rounded - H_plain| is what the precision itself does to the
     heat-maps (measured: 6.5e-4 on logits of scale 2.3 in eval mode, 1.5e-2 on scale 2.9 with batch statistics over
     2 x 32 x 32 samples).  The HIP bf16 model must stay within 2 E of the plain oracle AND of the rounded oracle
     (VERDICT r3 task 3: "state the bound; do not guess one" — E is computed in the test and printed).  It cannot track
     the rounded oracle more closely than ~E: an activation that sits on a bf16 rounding boundary rounds the other way
     when the fp32 sum in front of it differs in its last bit, which is a perturbation of the size of the rounding
     itself (measured: |H_hip - H_rounded| = 0.9-1.2 E);
  3. training works: the tiny RRNet over-fits one batch in bf16 exactly as the fp32 model does."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _cfg(backbone, bf16):
    return SimpleNamespace(num_classes=10, Model=SimpleNamespace(num_stacks=2, backbone=backbone, nms_type_for_stage1="nms",
                           nms_per_class_for_stage1=True, bf16=bf16), Train=SimpleNamespace(scale_factor=4))


def _maxdiff(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


@pytest.mark.parametrize("backbone,size,bs,training", [("hourglass_tiny", 128, 2, True), ("hourglass_tiny", 160, 2, False),
                                                        ("hourglass", 256, 1, False)])
def test_bf16_model_heatmaps_within_the_derived_bound(backbone, size, bs, training):
    from oracle import model as om
    from helpers import host_synth_batch as synth_batch
    from rrnet_amd.models.rrnet import RRNet
    torch.manual_seed(219)
    model = RRNet(_cfg(backbone, True))
    assert model.bf16
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    imgs = synth_batch(bs, size, size, boxes_per_image=8, seed=219)[0]

    def oracle(bf16):
        P = om.Params({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, training=training,
                      bf16=bf16)
        with torch.no_grad():
            feats = om.hourglass_net(P, imgs.double())
            return om.stage1(P, feats)

    plain, rounded = oracle(False), oracle(True)
    model = model.cuda().to(memory_format=CL)
    model.train(training)
    with torch.no_grad():
        outs = model(imgs.cuda(), k=50)
    for name, idx in (("heat-map logits", 0), ("wh", 1), ("offset", 2)):
        for s in range(2):
            e = _maxdiff(rounded[idx][s], plain[idx][s])
            scale = float(plain[idx][s].abs().max())
            d_model = _maxdiff(outs[idx][s], rounded[idx][s])
            d_fp32 = _maxdiff(outs[idx][s], plain[idx][s])
            print("%s stack %d (%s %dx%d %s): E = |bf16-rounded oracle - plain oracle| = %.3e (scale %.3g); HIP bf16 vs "
                  "rounded oracle %.3e; HIP bf16 vs plain oracle %.3e" % (name, s, backbone, size, size,
                                                                          "train" if training else "eval", e, scale, d_model, d_fp32))
            assert e > 1e-5 * scale                         # the rounding is visible at all
            assert d_fp32 > 1e-6 * scale                    # ... and the HIP model really ran in bf16
            assert d_model <= 2.0 * e, (name, s, d_model, e)
            assert d_fp32 <= 2.0 * e, (name, s, d_fp32, e)


def test_bf16_block_gradients_match_fp32_at_bf16_level():
    """A slice of the model — ResidualBlock (conv-bn-relu-conv-bn + projection skip, train-mode BN over 4 x 32 x 32
    samples) followed by a head convolution with bias + ReLU — forward and backward in both precisions from the same
    weights and a fixed random cotangent.  The output differs by ~4e-3 of its scale (2^-9 per operand).  The gradients
    differ by more than that, and must: ~0.3 % of the ReLU inputs lie within the forward's bf16 noise of zero and flip
    their mask, each flip an O(1) change of one element, i.e. ~sqrt(0.003) = 5 % in a sum over pixels (measured: median
    6.5e-2 of the parameter's max).  So the statement is directional: every parameter gradient (and the input gradient)
    keeps cosine similarity >= 0.99 and a norm within 5 % of the fp32 one — a wrong operand / scaling / tap order in a
    bf16 backward kernel gives cos ~ 0 — and the two are not identical (the bf16 kernels really ran).  The kernels
    themselves are pinned exactly (2e-5) in tests/test_conv_bf16_gpu.py and by the full-size audit below."""
    import torch.nn as nn
    from rrnet_amd import functional as RF, ops
    from rrnet_amd.backbones.hourglass import ResidualBlock
    from rrnet_amd.flat import FlatParams
    res = {}
    x0 = torch.randn(4, 128, 64, 64, generator=torch.Generator().manual_seed(3))
    g0 = torch.randn(4, 256, 32, 32, generator=torch.Generator().manual_seed(4)).cuda().to(memory_format=CL)
    for bf16 in (False, True):
        torch.manual_seed(5)
        blk = ResidualBlock(128, 256, stride=2).cuda().to(memory_format=CL).train()
        head = nn.Conv2d(256, 256, 3, padding=1).cuda().to(memory_format=CL)
        mods = nn.ModuleList([blk, head])
        fp = FlatParams(mods)
        fp.zero_grad()
        x = x0.cuda().to(memory_format=CL).requires_grad_()
        with ops.bf16_scope(bf16):
            y = RF.conv_bias(blk(x), head, relu=True)
        (y * g0).mean().backward()
        torch.cuda.synchronize()
        res[bf16] = (y.detach().clone(), x.grad.clone(), fp.grad.clone(),
                     [((p._rr_grad.data_ptr() - fp.grad.data_ptr()) // 4, p.numel()) for p in fp.params])
    y32, ybf = res[False][0], res[True][0]
    dy = float((ybf - y32).abs().max() / y32.abs().max())

    def cos_ratio(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return float((a @ b) / (a.norm() * b.norm())), float(a.norm() / b.norm())
    cx = cos_ratio(res[True][1], res[False][1])
    g32, gbf, sl = res[False][2], res[True][2], res[False][3]
    cr = [cos_ratio(gbf[o:o + n], g32[o:o + n]) for o, n in sl]
    print("ResidualBlock + head, bf16 vs fp32: output %.2e of its scale; input gradient cos %.5f norm ratio %.4f; parameter "
          "gradients: worst cos %.5f, norm ratio in [%.4f, %.4f]" % (dy, cx[0], cx[1], min(c for c, _ in cr),
                                                                     min(r for _, r in cr), max(r for _, r in cr)))
    assert 1e-4 < dy < 2e-2, dy
    assert cx[0] >= 0.99 and abs(cx[1] - 1) < 0.05, cx
    assert all(c >= 0.99 and abs(r - 1) < 0.05 for c, r in cr), cr
    assert not torch.equal(g32, gbf)


def test_bf16_rrnet_tiny_overfits_one_batch():
    """tests/test_train_gpu.py::test_rrnet_tiny_overfits_one_batch with cfg.Model.bf16: the whole training loop
    (forward, the four losses, backward, fused Adam, BN statistics) on one fixed batch — the loss must fall as it does in
    fp32.  A sign / scaling error in a bf16 backward kernel shows up here."""
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    saved = (cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, getattr(cfg.Model, "bf16", False))
    cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, cfg.Model.bf16 = 2, (256, 256), "hourglass_tiny", True
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    try:
        torch.manual_seed(219)
        op = RRNetOperator(cfg)
        assert op.model.module.bf16
        op.model.train()
        b = op.training_loader.get_batch()
        hist = []
        for step in range(60):
            _, losses = op.train_step(step, (b[0], b[1].clone()) + tuple(b[2:]))
            hist.append([float(v.detach()) for v in losses])
    finally:
        cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, cfg.Model.bf16 = saved
    hist = np.array(hist)
    assert np.isfinite(hist).all()
    first, last = hist[:5, 0].mean(), hist[-5:, 0].mean()
    print("bf16 overfit: loss %.3f -> %.3f, heat-map focal %.3f -> %.3f" % (first, last, hist[:5, 1].mean(), hist[-5:, 1].mean()))
    assert last < 0.6 * first, (first, last)
    assert hist[-5:, 1].mean() < hist[:5, 1].mean()


def test_config4_bf16_full_size_train_step_every_kernel_call_sampled():
    """The config-4 launches themselves: one train step at B=8, 1024x1024, hourglass-104 with cfg.Model.bf16, every
    distinct kernel call audited against a host fp64 recomputation from the inputs the call received — the convolutions
    on bf16-ROUNDED operands (the kernels' contract), everything else as in the fp32 audit
    (tests/test_configs_gpu.py::test_config2_full_size_train_step_every_kernel_call_sampled).  Tolerances unchanged:
    2e-5 (wgrad 2e-4) of the output scale: what is left after rounding the operands is summation order."""
    import time
    from kernel_audit import audit
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    saved = (cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, getattr(cfg.Model, "bf16", False))
    cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, cfg.Model.bf16 = 8, (1024, 1024), "hourglass", True
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    try:
        torch.manual_seed(cfg.seed)
        op = RRNetOperator(cfg)
        op.model.train()
        b = op.training_loader.get_batch()
        t0 = time.perf_counter()
        with audit(sample=True, ref_device="cuda") as rec:
            _, losses = op.train_step(0, b)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, cfg.Model.bf16 = saved
    assert all(np.isfinite(float(v.detach())) for v in losses)
    kinds, nbf = {}, {}
    for key, err in rec.seen.items():
        kinds.setdefault(key[0], []).append(err)
        if key[-1] == "bf16":
            nbf[key[0]] = nbf.get(key[0], 0) + 1
    print("bf16 audit %.0f s: " % dt + "  ".join("%s:%d (max %.1e)" % (kk, len(v), max(v)) for kk, v in sorted(kinds.items())))
    print("bf16-operand signatures: %s; sampled: %d of %d" % (nbf, len(rec.sampled), len(rec.seen)))
    assert not rec.bad, rec.bad[:10]
    assert len(rec.seen) > 150
    # the bulk of the convolutions ran on the bf16 kernels
    assert nbf.get("fprop", 0) >= 30 and nbf.get("dgrad", 0) >= 25 and nbf.get("wgrad", 0) >= 25, nbf
    big = (8, 256, 256, 256)
    keys = set(rec.seen)
    for w_ in (("fprop", big, (256, 256, 3, 3), 1), ("dgrad", big, (256, 256, 3, 3), big, 1), ("wgrad", big, big, (256, 256, 3, 3), 1)):
        assert any(k[:len(w_)] == w_ and k[-1] == "bf16" for k in keys), w_
