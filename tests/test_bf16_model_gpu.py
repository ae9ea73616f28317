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


def _config4_operator(backbone, batch, size):
    """RRNetOperator built exactly as bench.py / tools/bench_config4.py build the timed config-4 step: cfg.Model.bf16 +
    dcn_heads + dcn_bf16 TOGETHER (flat parameter buffer, per-step bf16 filter copies, shared fan-in buffers), with the
    DCN offset / mask convolutions moved off their zero initialisation so that the deformation is real."""
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = batch, (size, size), backbone
    cfg.Model.bf16, cfg.Model.dcn_heads, cfg.Model.dcn_bf16 = True, True, True
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    torch.manual_seed(cfg.seed)
    op = RRNetOperator(cfg)
    op.model.train()
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for name, p in op.model.module.named_parameters():
            if "conv_offset_mask.weight" in name:
                p.copy_(torch.empty(p.shape).normal_(0, 0.02, generator=g).to(p.device))
            if "conv_offset_mask.bias" in name:
                p.copy_(torch.empty(p.shape).normal_(0, 0.3, generator=g).to(p.device))
    op.optimizer.fp.invalidate_wt()
    return op, cfg


class _config4_cfg:
    """Restores the shared Config object the operator tests mutate."""

    def __enter__(self):
        from rrnet_amd.configs.rrnet_config import Config as cfg
        self.cfg = cfg
        self.saved = (cfg.Train.batch_size, cfg.Train.crop_size, dict(cfg.Model))
        return self

    def __exit__(self, *exc):
        cfg = self.cfg
        cfg.Train.batch_size, cfg.Train.crop_size = self.saved[0], self.saved[1]
        for k in list(cfg.Model):
            if k not in self.saved[2]:
                del cfg.Model[k]
        cfg.Model.update(self.saved[2])


def test_config4_tiny_bf16_with_dcn_heads_all_gradients_vs_oracle_composition():
    """BASELINE configs[3] AS bench.py TIMES IT — cfg.Model.bf16 and cfg.Model.dcn_heads (+ dcn_bf16) TOGETHER — on the
    tiny backbone at 2 x 256 x 256 (the heads see 2 x 64 x 64 = 8192 pixels of 256 channels: the size from which the bf16
    launches go to csrc/conv16.hip and the DCN gradients read bf16 images), train mode, non-degenerate offsets.
    Reference: /root/reference/ext/dcn/dcn_v2.py:105-122 composed with /root/reference/detectors/centernet_detector.py:12-14
    under the builder's bf16 arithmetic, restated in fp64 by oracle/model.py (_ConvBf16) + oracle/dcn.py (_ContractBf16):
    both operands of every matrix product rounded to bf16, forward and backward.

    Forward: the six stage-1 maps within 2 E of the rounded AND of the plain oracle, E = |rounded - plain| (the bound
    test_bf16_model_heatmaps_within_the_derived_bound derives; measured |HIP - rounded| = 0.8-1.2 E).
    Backward, two statements about the gradient of a dense smooth functional of the head outputs for EVERY parameter:
      (i)  per kernel call, tight: the whole forward + backward runs under tests/kernel_audit.py — every distinct launch (the
           DCN forward / data gradient incl. `out=` accumulation and image-fed dY / weight gradient, the 28-filter offset
           convolution, every bf16 convolution, BatchNorm, fan-in) equals its fp64 recomputation on bf16-rounded operands
           from the inputs the call received (2e-5; DCN bf16 calls 1e-4; weight gradients 2e-4);
      (ii) end to end, derived like E: E_g(p) = ||G_rounded - G_plain|| / ||G_plain|| is what the precision itself does to the
           gradient of parameter p.  It is LARGE in this network (measured: median 0.47): ~0.3 % of the ReLU inputs of each of
           the ~45 layers lie within bf16 noise of zero and flip their mask, each flip an O(1) change of one gradient path.
           The HIP model is a third sample of the same noise (its own flips are independent of the rounded oracle's: measured
           ||HIP - rounded||^2 = ||HIP - plain||^2 + ||rounded - plain||^2 to 1 %), so the bounds are: ||HIP - plain|| <=
           1.5 E_g + 0.02 and ||HIP - rounded|| <= 2 E_g + 0.02 (norms relative to ||G_plain||) for every parameter — a
           missing / doubled / sign-flipped contribution to a parameter's gradient is an error of >= 1 — and the HIP
           gradient differs from the plain oracle's by more than fp32 rounding (the bf16 kernels really ran)."""
    from oracle import model as om
    from helpers import host_synth_batch as synth_batch
    with _config4_cfg():
        op, cfg = _config4_operator("hourglass_tiny", 2, 256)
        model = op.model.module
        assert model.bf16
        dcns = [m for m in model.modules() if type(m).__name__ == "DCN"]
        assert len(dcns) == 6 and all(m.bf16 is True for m in dcns)
        fp = op.optimizer.fp
        names = {id(p): n for n, p in model.named_parameters()}
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        imgs = synth_batch(2, 256, 256, boxes_per_image=10, seed=3)[0]
        g = torch.Generator().manual_seed(13)
        proj = [torch.randn(2, c, 64, 64, generator=g) for c in (10, 2, 2)]

        def oracle(bf16):
            sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
            leaves = {k: v.requires_grad_() for k, v in sd64.items() if v.is_floating_point() and "running_" not in k}
            P = om.Params(sd64, training=True, bf16=bf16)
            hm, wh, off = om.stage1(P, om.hourglass_net(P, imgs.double()))
            sum((o * w.double()).sum() for i in range(2) for o, w in zip((hm[i], wh[i], off[i]), proj)).backward()
            return [[t.detach() for t in hm], [t.detach() for t in wh], [t.detach() for t in off]], \
                   {k: v.grad for k, v in leaves.items()}

        plain, g_plain = oracle(False)
        rounded, g_rounded = oracle(True)
        from kernel_audit import audit
        op.optimizer.zero_grad()
        with audit(ref_device="cuda") as rec:
            outs = model(imgs.cuda().contiguous(memory_format=CL), k=50)
            for name, idx in (("heat-map logits", 0), ("wh", 1), ("offset", 2)):
                for s in range(2):
                    e = _maxdiff(rounded[idx][s], plain[idx][s])
                    d_model, d_fp32 = _maxdiff(outs[idx][s], rounded[idx][s]), _maxdiff(outs[idx][s], plain[idx][s])
                    scale = float(plain[idx][s].abs().max())
                    print("%s stack %d: E %.3e (scale %.3g); HIP vs rounded oracle %.3e; vs plain %.3e" % (name, s, e, scale, d_model, d_fp32))
                    assert e > 1e-5 * scale and d_fp32 > 1e-6 * scale
                    assert d_model <= 2.0 * e and d_fp32 <= 2.0 * e, (name, s, d_model, d_fp32, e)
            sum((o * w.cuda()).sum() for i in range(2) for o, w in zip((outs[0][i], outs[1][i], outs[2][i]), proj)).backward()
            torch.cuda.synchronize()
        kinds = {}
        for key, err in rec.seen.items():
            kinds.setdefault(key[0], []).append(err)
        print("per-call audit: " + "  ".join("%s:%d (max %.1e)" % (kk, len(v), max(v)) for kk, v in sorted(kinds.items())))
        assert not rec.bad, rec.bad[:10]
        for kind in ("dcn_fwd", "dcn_dgrad", "dcn_dgrad_doffset", "dcn_dgrad_dmask", "dcn_wgrad", "fprop", "dgrad", "wgrad", "bn_bwd_apply"):
            assert kind in kinds, (kind, sorted(kinds))
        assert any(k[0] == "dcn_dgrad" and k[7] is True and k[8] is True for k in rec.seen), "no DCN data gradient added into a fan-in buffer"
        rows = []
        for p in fp.params:
            key = names[id(p)]
            if key.startswith("head_detector"):          # stage 2 is not part of this functional
                continue
            got = p._rr_grad.detach().double().cpu()
            gr, gp = g_rounded[key], g_plain[key]
            npl = float(gp.norm())
            assert npl > 0, key
            e_g = float((gr - gp).norm()) / npl
            d_r = float((got - gr).norm()) / npl
            d_p = float((got - gp).norm()) / npl
            cos = float((got.flatten() @ gr.flatten()) / (got.norm() * gr.norm()).clamp_min(1e-300))
            rows.append((key, e_g, d_r, d_p, cos))
        rows.sort(key=lambda r: -r[2] / max(r[1], 1e-12))
        for key, e_g, d_r, d_p, cos in rows[:12]:
            print("  %-58s E_g %.2e  HIP-rounded %.2e  HIP-plain %.2e  cos %.5f" % (key, e_g, d_r, d_p, cos))
        eg = np.array([r[1] for r in rows]); dr = np.array([r[2] for r in rows]); dp = np.array([r[3] for r in rows])
        print("%d parameters: E_g median %.2e max %.2e; HIP vs rounded oracle median %.2e max %.2e; vs plain median %.2e; worst "
              "cos %.5f" % (len(rows), np.median(eg), eg.max(), np.median(dr), dr.max(), np.median(dp), min(r[4] for r in rows)))
        assert len(rows) > 100
        dcn_keys = [r for r in rows if "conv_offset_mask" in r[0] or ".0.conv.weight" in r[0]]
        assert len(dcn_keys) >= 18, len(dcn_keys)
        for key, e_g, d_r, d_p, cos in rows:
            assert d_r <= 2.0 * e_g + 2e-2, (key, d_r, e_g)
            assert d_p <= 1.5 * e_g + 2e-2, (key, d_p, e_g)
        # the three gradients behave as the truth + two independent samples of the precision's noise
        tri = np.sqrt(dp ** 2 + eg ** 2)
        big = eg > 0.05
        print("||HIP - rounded|| / sqrt(||HIP - plain||^2 + E_g^2): median %.3f over the %d parameters with E_g > 0.05" %
              (np.median(dr[big] / tri[big]), int(big.sum())))
        assert 0.7 <= np.median(dr[big] / tri[big]) <= 1.3
        assert np.median(dp) > 1e-5            # the bf16 kernels really ran


def test_config4_bf16_dcn_heads_full_size_train_step_every_kernel_call_sampled():
    """The config-4 step bench.py times (`config4.with_dcn_heads`): B=8, 1024x1024, hourglass-104, cfg.Model.bf16 +
    dcn_heads + dcn_bf16, offsets off their zero initialisation — one train step with every distinct kernel call audited
    against an fp64 recomputation from the inputs the call received.  On top of the plain bf16 audit above this covers
    the launch graph only this composition has: the DCN forward on the window kernel, its data gradient ADDING into the
    pre-filled shared fan-in buffer of relu(feature) (`out=`), dY fed from its producer's bf16 image to both DCN
    gradients, the 28-filter offset / mask convolution's three launches on the bf16 kernels (tests/kernel_audit.py:
    dcn_fwd / dcn_dgrad / dcn_wgrad hooks; reference oracle/dcn.py, /root/reference/ext/dcn/dcn_v2.py:105-122,
    ext/dcn/src/cuda/dcn_v2_im2col_cuda.cu:125-327)."""
    import time
    from kernel_audit import audit
    with _config4_cfg():
        op, cfg = _config4_operator("hourglass", 8, 1024)
        b = op.training_loader.get_batch()
        t0 = time.perf_counter()
        with audit(sample=True, ref_device="cuda") as rec:
            _, losses = op.train_step(0, b)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert all(np.isfinite(float(v.detach())) for v in losses)
    kinds = {}
    for key, err in rec.seen.items():
        kinds.setdefault(key[0], []).append(err)
    print("config-4 (bf16 + DCN heads) audit %.0f s: " % dt + "  ".join("%s:%d (max %.1e)" % (kk, len(v), max(v)) for kk, v in sorted(kinds.items())))
    assert not rec.bad, rec.bad[:10]
    feat = (8, 256, 256, 256)
    keys = set(rec.seen)
    dcn = lambda kind: [k for k in keys if k[0] == kind and k[1] == feat]
    assert any(k[7] is True for k in dcn("dcn_fwd")), dcn("dcn_fwd")                         # bf16 operands
    dg = dcn("dcn_dgrad")
    assert any(k[7] is True and k[8] is True for k in dg), dg                                # ... adding into a pre-filled fan-in buffer
    assert any(k[7] is True and k[9] is True for k in dg), dg                                # ... dY from its producer's bf16 image
    assert any(k[7] is True and k[8] is True for k in dcn("dcn_wgrad")), dcn("dcn_wgrad")    # image-fed weight gradient
    assert any(k[0] == "dcn_dy_image" for k in keys)
    # the offset / mask convolution (27 -> 28 filters) ran on the bf16 kernels, all three launches
    om = (28, 256, 3, 3)
    assert any(k[0] == "fprop" and k[2] == om and k[-1] == "bf16" for k in keys)
    assert any(k[0] == "dgrad" and k[2] == om and k[-1] == "bf16" for k in keys)
    assert any(k[0] == "wgrad" and k[3] == om and k[-1] == "bf16" for k in keys)
