"""Parity at the configurations BASELINE.json names (the golden-vector tests run the same code at fixture sizes):
  configs[0]  CenterNet + hourglass-tiny on 2 synthetic 512x512 frames: outputs and the three losses against the
              CPU oracle on the same seeded frames and default-initialised weights;
  configs[1]  RRNet hourglass-104 (the full 191 M-parameter model): heat-maps / wh / offset of both stacks and the
              decoded boxes against the CPU oracle, eval-mode BN, on a reduced 256x256 frame so that the oracle
              finishes in seconds (the 1024x1024 batch itself is covered by size-independent properties:
              conv linearity at the 256x256 layer, decode sortedness, NMS idempotence, and the train step below)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _cfg(backbone):
    return SimpleNamespace(num_classes=10, Model=SimpleNamespace(num_stacks=2, backbone=backbone, nms_type_for_stage1="nms",
                           nms_per_class_for_stage1=True), Train=SimpleNamespace(scale_factor=4))


def _close(a, b, atol=1e-3, rtol=1e-3):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), atol=atol, rtol=rtol)


def test_config1_centernet_tiny_512():
    from oracle import model as om, ops as oo
    from rrnet_amd import functional as RF
    from helpers import host_synth_batch as synth_batch
    from rrnet_amd.models.centernet import CenterNet
    torch.manual_seed(219)
    model = CenterNet(_cfg("hourglass_tiny"))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    imgs, annos, hms, whs, inds, offs, masks, _ = synth_batch(2, 512, 512, boxes_per_image=100, seed=219)
    P = om.Params(sd, training=True)
    with torch.no_grad():
        r_hm, r_wh, r_reg = om.centernet_forward(P, imgs)
        r_hm_l = sum(oo.hm_loss_from_logits(r_hm[i], hms) / 2 for i in range(2))
        r_wh_l = sum(oo.reg_l1_loss(r_wh[i], masks, inds, whs) / 2 for i in range(2))
        r_off_l = sum(oo.reg_l1_loss(r_reg[i], masks, inds, offs) / 2 for i in range(2))
    model = model.cuda().to(memory_format=CL).train()
    g_hm, g_wh, g_reg = model(imgs.cuda())
    for i in range(2):
        _close(g_hm[i], r_hm[i]); _close(g_wh[i], r_wh[i]); _close(g_reg[i], r_reg[i])
    gt = [t.cuda() for t in (hms, whs, inds, offs, masks)]
    hm_l = sum(RF.focal_loss_hm_from_logits(g_hm[i], gt[0]) / 2 for i in range(2))
    wh_l = sum(RF.reg_l1_loss(g_wh[i], gt[4], gt[2], gt[1]) / 2 for i in range(2))
    off_l = sum(RF.reg_l1_loss(g_reg[i], gt[4], gt[2], gt[3]) / 2 for i in range(2))
    _close(hm_l, r_hm_l); _close(wh_l, r_wh_l); _close(off_l, r_off_l)
    (hm_l + 0.1 * wh_l + off_l).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_config2_hourglass104_forward_eval_256():
    from oracle import model as om, ops as oo
    from helpers import host_synth_batch as synth_batch
    from rrnet_amd.models.rrnet import RRNet
    torch.manual_seed(219)
    model = RRNet(_cfg("hourglass"))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    imgs = synth_batch(1, 256, 256, boxes_per_image=20, seed=219)[0]
    P = om.Params(sd, training=False)
    with torch.no_grad():
        feats = om.hourglass_net(P, imgs)
        r_hm, r_wh, r_off = om.stage1(P, feats)
        r_boxes = oo.transform_bbox(r_hm[-1], r_wh[-1], r_off[-1], 100)
    model = model.cuda().to(memory_format=CL).eval()
    with torch.no_grad():
        g_hm, g_wh, g_off, reg, rois, scores, clses = model(imgs.cuda(), k=100)
        g_boxes = model.transform_bbox(g_hm[-1], g_wh[-1], g_off[-1], k=100)
    for i in range(2):
        _close(g_hm[i], r_hm[i]); _close(g_wh[i], r_wh[i]); _close(g_off[i], r_off[i])
    # decoded boxes: scores within 1e-4; rows may swap only between near-tied scores
    gs, rs = g_boxes[0, :, 4].cpu(), r_boxes[0, :, 4]
    np.testing.assert_allclose(gs.numpy(), rs.numpy(), atol=1e-4)
    same = (g_boxes[0, :, 5].cpu() == r_boxes[0, :, 5])
    gap = torch.minimum((rs - torch.roll(rs, 1)).abs(), (rs - torch.roll(rs, -1)).abs())
    assert torch.all(gap[~same] < 1e-4)
    np.testing.assert_allclose(g_boxes[0, same, :4].cpu().numpy(), r_boxes[0, same, :4].numpy(), atol=2e-3, rtol=1e-3)
    assert torch.isfinite(reg).all() and rois.shape[0] == scores.shape[0] == clses.shape[0] > 0


def test_config2_full_size_train_step_properties():
    """One real train step at BASELINE configs[1] (B=8, 1024x1024, hourglass-104): finite losses, every parameter
    receives a finite gradient and moves, BN running statistics update — size-independent sanity of the bench path."""
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    cfg.Train.batch_size = 8
    cfg.Train.crop_size = (1024, 1024)
    cfg.Model.backbone = "hourglass"
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    torch.manual_seed(cfg.seed)
    op = RRNetOperator(cfg)
    op.model.train()
    flat = op.model.flat
    before = flat.flat.clone()
    rm0 = op.model.module.backbone.pre_layer[1].running_mean.clone()
    b = op.training_loader.get_batch()
    _, losses = op.train_step(0, (b[0], b[1].clone()) + tuple(b[2:]))
    vals = [float(v.detach()) for v in losses]
    assert all(np.isfinite(v) for v in vals), vals
    assert torch.isfinite(flat.flat).all() and torch.isfinite(flat.grad).all()
    moved = (flat.flat != before).float().mean().item()
    # Adam moves every parameter that received a gradient: all but the stage-2 head (its loss is gated off before
    # step 2000, rrnet_operator.py:131) and the alignment padding of the flat buffer
    assert moved > 0.99, moved
    assert not torch.equal(op.model.module.backbone.pre_layer[1].running_mean, rm0)


# --- configs[1] in TRAIN mode (the bench path): batch-statistic BN through all 163 layers, the four losses, the
# --- running statistics and selected gradients of the 191 M-parameter model against the oracle ------------------
_HG104_GRAD_KEYS = (
    "backbone.pre_layer.0.weight",                                   # stem 7x7 (the far end of backward)
    "backbone.pre_layer.1.weight",                                   # its BN gamma
    "backbone.hgs.0.low2.low2.low2.low2.low2.1.conv1.weight",        # an innermost 512-channel block, stack 1
    "backbone.hgs.1.low2.low2.low2.low2.low2.3.conv2.weight",        # ... and of stack 2
    "backbone.hgs.1.up1.1.conv2.weight",                             # 256-channel block at the full 1/4 resolution
    "backbone.convs.1.conv.weight",
    "hm.detect_layer.1.1.weight", "hm.detect_layer.1.1.bias",        # last head convolutions
    "wh.detect_H_layer.1.0.conv.weight", "offset_reg.detect_layer.0.0.conv.weight",
    "head_detector.regressor.weight",
)


def test_config2_full_size_train_step_every_kernel_call_sampled():
    """The headline's OWN launches: one train step at BASELINE configs[1] (B=8, 1024x1024, hourglass-104 — the
    configuration bench.py times) with every distinct conv / BatchNorm / element-wise kernel call audited against a host
    recomputation from the inputs the call received (tests/kernel_audit.py, sample=True).  The variants that only exist
    at this size — wgrad's split counts at 8x256x256 pixels, the stem's space-to-depth wgrad at 512x512, stride-2 dgrad
    512->256, the two-kernel BatchNorm statistics of > 512 pixel tiles, capped reduce grids — are checked in the exact
    launch configuration the step selects: fprop / dgrad on three 16-row bands of three images (fp64), wgrad on 16-24
    filters over ALL pixels (fp64) plus the full dw of the largest layers against torch-CPU fp32, reductions over the
    whole tensors.  Tolerances as in the small audit: 2e-5 (wgrad 2e-4)."""
    import time
    from kernel_audit import audit
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    cfg.Train.batch_size = 8
    cfg.Train.crop_size = (1024, 1024)
    cfg.Model.backbone = "hourglass"
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    torch.manual_seed(cfg.seed)
    op = RRNetOperator(cfg)
    op.model.train()
    b = op.training_loader.get_batch()
    t0 = time.perf_counter()
    with audit(sample=True, ref_device="cuda") as rec:      # fp64 reference arithmetic by torch's own device kernels (kernel_audit.REF)
        _, losses = op.train_step(0, b)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(np.isfinite(float(v.detach())) for v in losses)
    kinds = {}
    for key, err in rec.seen.items():
        kinds.setdefault(key[0], []).append(err)
    print("audit %.0f s: " % dt + "  ".join("%s:%d (max %.1e)" % (kk, len(v), max(v)) for kk, v in sorted(kinds.items())))
    print("sampled signatures: %d of %d" % (len(rec.sampled), len(rec.seen)))
    assert not rec.bad, rec.bad[:10]
    assert len(rec.seen) > 150
    assert {"fprop", "dgrad", "wgrad", "wgrad_full_fp32", "fprop_packed", "wgrad_packed", "dgrad_bnsum", "fprop_stats", "bn_apply", "bn_bwd_apply",
            "bn_bwd_reduce", "bn_finalize", "bn_stats_finalize", "sum_n", "upsample_add_fwd", "upsample_add_bwd",
            "bias_relu_bwd"} <= set(kinds), sorted(kinds)
    # the signatures that exist only at this size were seen AND sampled
    big = (8, 256, 256, 256)
    want = [("fprop", big, (256, 256, 3, 3), 1), ("dgrad", big, (256, 256, 3, 3), big, 1), ("wgrad", big, big, (256, 256, 3, 3), 1),
            ("fprop_packed", (8, 3, 1024, 1024), (128, 3, 7, 7), 2), ("fprop", (8, 128, 512, 512), (256, 128, 3, 3), 2),
            ("fprop", (8, 128, 512, 512), (256, 128, 1, 1), 2), ("dgrad", big, (256, 128, 3, 3), (8, 128, 512, 512), 2)]
    keys = set(rec.seen)
    for w_ in want:
        assert any(k[:len(w_)] == w_ for k in keys), (w_, sorted(k for k in keys if k[0] == w_[0])[:5])
    assert any(k[0] == "wgrad_packed" and k[1] == (8, 160, 512, 512) for k in keys)      # the stem: 147 taps padded to 160


def _hg104_oracle_train(sd, batch, dtype, keys, k, perturb=0.0):
    """One oracle train step (forward, criterion, backward) -> outputs, losses, grads, BN buffers."""
    from oracle import model as om, ops as oo
    sd = {kk: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for kk, v in sd.items()}
    for kk in keys:
        sd[kk].requires_grad_()
    imgs, annos, hms, whs, inds, offs, masks = [t.to(dtype) for t in batch]
    if perturb:
        g = torch.Generator().manual_seed(5)
        imgs = imgs * (1 + perturb * torch.randn(imgs.shape, dtype=dtype, generator=g))
    P = om.Params(sd, training=True)
    outs = om.rrnet_forward(P, imgs, k=k)
    losses = oo.criterion(outs, (hms, whs, inds, offs, masks, annos.clone()))
    (losses[0] + 0.1 * losses[1] + losses[2] + losses[3]).backward()
    grads = {kk: sd[kk].grad.double().numpy() for kk in keys}
    return outs, [float(l.detach()) for l in losses], grads, sd


def _matched_batch(sd, imgs, k, per_image=12):
    """Ground truth that the stage-2 criterion can match (otherwise a randomly initialised model has no RoI with
    IoU > 0.5 and the stage-2 loss and its gradients are identically zero): a first oracle forward decodes the
    proposals, `per_image` of them — shifted and rescaled by a few percent, image coordinates — become the
    annotations, and the stage-1 targets are built from those annotations by the oracle's host contract
    (oracle/targets.py: to_heatmap + collate_fn_ctnet)."""
    from oracle import model as om
    from oracle.targets import collate_ctnet, to_heatmap
    with torch.no_grad():
        P = om.Params({kk: v.clone() for kk, v in sd.items()}, training=True)
        rois = om.rrnet_forward(P, imgs, k=k)[4]
    rng = np.random.default_rng(7)
    samples = []
    for b in range(imgs.size(0)):
        r = rois[rois[:, 0] == b][:, 1:].numpy() * 4.0
        r = r[(r[:, 2] - r[:, 0] > 4) & (r[:, 3] - r[:, 1] > 4) & (r[:, 0] > 0) & (r[:, 1] > 0)
              & (r[:, 2] < imgs.size(3) - 1) & (r[:, 3] < imgs.size(2) - 1)]
        assert r.shape[0] >= per_image, r.shape
        r = r[rng.choice(r.shape[0], per_image, replace=False)]
        w, h = r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]
        x = r[:, 0] + rng.uniform(-0.08, 0.08, per_image) * w
        y = r[:, 1] + rng.uniform(-0.08, 0.08, per_image) * h
        w, h = w * rng.uniform(0.9, 1.1, per_image), h * rng.uniform(0.9, 1.1, per_image)
        a = np.stack([x, y, w, h, np.ones(per_image), rng.integers(1, 11, per_image), np.zeros(per_image),
                      np.zeros(per_image)], 1).astype(np.float32)
        a = torch.from_numpy(a)
        _, a, hm, wh, ind, off, mask = to_heatmap((imgs[b], a), 4, 10)
        samples.append((imgs[b], a, hm, wh, ind, off, mask, "f%d" % b))
    return collate_ctnet(samples)[:7]


@pytest.mark.parametrize("size,bs", [(256, 2)])
def test_config2_hourglass104_train_mode_vs_oracle(size, bs):
    """operators/rrnet_operator.py:42-84,128-138 on the full-depth model in train mode."""
    from rrnet_amd import functional as RF
    from helpers import host_synth_batch as synth_batch
    from rrnet_amd.models.rrnet import RRNet
    k = 100
    torch.manual_seed(219)
    model = RRNet(_cfg("hourglass"))
    for i in range(2):      # proposals of a plausible size (12 px) instead of the ~0-pixel boxes of a fresh wh head
        model.wh.detect_H_layer[i][0].conv.bias.data.fill_(3.0)
        model.wh.detect_W_layer[i][0].conv.bias.data.fill_(3.0)
    sd0 = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    batch = _matched_batch(sd0, synth_batch(bs, size, size, boxes_per_image=4, seed=219)[0], k)
    keys = list(_HG104_GRAD_KEYS)
    r_outs, r_losses, r_grads, r_sd = _hg104_oracle_train(sd0, batch, torch.float32, keys, k)
    assert r_losses[3] > 1e-3, r_losses          # the stage-2 loss has positives

    model = model.cuda().to(memory_format=CL).train()
    imgs, annos, hms, whs, inds, offs, masks = [t.cuda() for t in batch]
    outs = model(imgs, k=k)
    for i in range(2):
        _close(outs[0][i], r_outs[0][i]); _close(outs[1][i], r_outs[1][i]); _close(outs[2][i], r_outs[2][i])
    hm_l = sum(RF.focal_loss_hm_from_logits(outs[0][i], hms) / 2 for i in range(2))
    wh_l = sum(RF.reg_l1_loss(outs[1][i], masks, inds, whs) / 2 for i in range(2))
    off_l = sum(RF.reg_l1_loss(outs[2][i], masks, inds, offs) / 2 for i in range(2))
    a = annos.clone()
    a[:, :, 2:4] += a[:, :, 0:2]
    s2_l = RF.stage2_reg_loss(outs[3], outs[4], a, 4.0)
    got_losses = [float(v.detach()) for v in (hm_l, wh_l, off_l, s2_l)]
    np.testing.assert_allclose(got_losses, r_losses, rtol=1e-3, atol=1e-3)
    # proposals: the same set of rows; inside an (image, class) segment two rows may trade places when their scores
    # are an ulp or two apart (sigmoid / conv rounding), so rows are matched by position before comparing
    assert outs[4].shape == r_outs[4].shape, (outs[4].shape, r_outs[4].shape)
    mine = torch.cat((outs[4].detach().cpu(), outs[6].detach().cpu().view(-1, 1)), 1).numpy()
    ref = torch.cat((r_outs[4].detach(), r_outs[6].detach().view(-1, 1)), 1).numpy()
    perm = np.full(mine.shape[0], -1)
    for r in range(mine.shape[0]):
        cand = np.where((ref[:, 0] == mine[r, 0]) & (ref[:, 5] == mine[r, 5]))[0]
        perm[r] = cand[np.abs(ref[cand, 1:5] - mine[r, 1:5]).sum(1).argmin()]
    assert len(set(perm.tolist())) == mine.shape[0]                       # a bijection
    moved = np.where(perm != np.arange(mine.shape[0]))[0]
    sc_m, sc_r = outs[5].detach().cpu().numpy(), r_outs[5].detach().numpy()
    assert len(moved) <= 0.05 * mine.shape[0], len(moved)
    assert np.all(np.abs(sc_m[moved] - sc_r[moved]) < 1e-5)                # swaps only between near-tied scores
    np.testing.assert_allclose(mine[:, 1:5], ref[perm, 1:5], atol=2e-3, rtol=1e-3)
    np.testing.assert_allclose(sc_m, sc_r[perm], atol=1e-4)
    np.testing.assert_allclose(outs[3].detach().cpu().numpy(), r_outs[3].detach().numpy()[perm], atol=2e-3, rtol=1e-3)
    (hm_l + 0.1 * wh_l + off_l + s2_l).backward()
    # BN running statistics of the first and the last BN layer of the backbone + the stage-2 head's
    for key in ("backbone.pre_layer.1", "backbone.convs.1.bn", "head_detector.top_layer.bn3"):
        mod = model.get_submodule(key)
        _close(mod.running_mean, r_sd[key + ".running_mean"], atol=1e-4)
        _close(mod.running_var, r_sd[key + ".running_var"], atol=1e-4)
        assert int(mod.num_batches_tracked) == int(r_sd[key + ".num_batches_tracked"]) == 1
    # gradients under a bound derived from fp64 runs of the oracle (see test_model_gpu.py::
    # test_centernet_tiny_vs_reference_golden): truth = fp64 oracle; e_ref = the fp32 oracle's own distance from it;
    # noise = movement of the fp64 gradient when the INPUT is disturbed by one fp32 rounding (6e-8 relative).  A third
    # term covers ReLU decisions: any fp32 forward lands ~1e-4 away from the fp64 activations after ~100 train-mode
    # BN layers, so a few hundred pre-activations within that distance of zero fall on the other side ("flips");
    # each flip changes the gradient discretely without being an error (tests/test_model_gpu.py::
    # test_stage1_heads_backward_exact_given_masks shows the backward is exact once the masks are given).  Its size
    # is measured, not guessed: the input disturbance is scaled until the fp64 forward moves as far from the truth
    # as the HIP forward is (at the last stack's heat-map), and the gradient movement of THAT run (`e_flip`) bounds
    # what a correct implementation with this forward accuracy can show.
    t_outs, _, truth, _ = _hg104_oracle_train(sd0, batch, torch.float64, keys, k)
    p0 = 6e-8
    m_outs, _, moved, _ = _hg104_oracle_train(sd0, batch, torch.float64, keys, k, perturb=p0)
    hm_t = t_outs[0][1].detach()
    d_mine = float((outs[0][1].detach().cpu().double() - hm_t).norm() / hm_t.norm())
    d_ref = float((r_outs[0][1].detach().double() - hm_t).norm() / hm_t.norm())
    d_p0 = float((m_outs[0][1].detach() - hm_t).norm() / hm_t.norm())
    assert d_mine <= 4 * d_ref + 1e-6, (d_mine, d_ref)              # forward accuracy on par with the fp32 oracle
    p1 = min(p0 * max(d_mine, d_ref) / max(d_p0, 1e-30), 1e-3)
    _, _, flipped, _ = _hg104_oracle_train(sd0, batch, torch.float64, keys, k, perturb=p1)
    named = dict(model.named_parameters())
    report, bad = [], []
    for kk in keys:
        t = truth[kk]
        scale = max(np.abs(t).max(), 1e-9)
        tn = np.linalg.norm(t)
        g = named[kk].grad.detach().cpu().numpy().astype(np.float64)
        e = {n: np.abs(v - t).max() for n, v in (("mine", g), ("ref", r_grads[kk]), ("noise", moved[kk]), ("flip", flipped[kk]))}
        d = {n: np.linalg.norm(v.ravel() - t.ravel()) / tn for n, v in (("mine", g), ("ref", r_grads[kk]), ("flip", flipped[kk]))}
        report.append((kk, e["mine"] / scale, e["ref"] / scale, e["noise"] / scale, e["flip"] / scale, d["mine"], d["ref"], d["flip"]))
        # floor: 1e-4 of the gradient's scale — the head parameters (short backward path) are pinned to that
        if not e["mine"] <= max(8 * e["ref"], 16 * e["noise"], 4 * e["flip"], 1e-4 * scale):
            bad.append(("max", kk, e, scale))
        if not d["mine"] <= max(4 * d["ref"], 4 * d["flip"], 1e-4):      # whole tensor: relative L2 deviation
            bad.append(("l2", kk, d))
    print("forward deviation at hm[1]: mine %.1e ref %.1e; input disturbance matched to it: %.1e" % (d_mine, d_ref, p1))
    print("\n".join("%-56s max: mine %.1e ref %.1e noise %.1e flip %.1e | l2: mine %.1e ref %.1e flip %.1e" % r for r in report))
    assert not bad, bad


@pytest.mark.parametrize("h,w,bs", [(256, 256, 2), (192, 320, 1)])
def test_config2_hourglass104_train_step_every_kernel_call(h, w, bs):
    """The tight check of the bench path's arithmetic: one full train step (forward, criterion, backward) of RRNet
    hourglass-104 in train mode with EVERY distinct conv / BatchNorm / elementwise kernel call recomputed on the
    host in fp64 from the inputs the call actually received (tests/kernel_audit.py) — ~220 distinct (kernel,
    shape, stride, flags) combinations per size, covering the large-layer variants (pipelined tiles, wgrad's
    uniform row walk at Q % 32 == 0 and the generic walk at 320/4 = 80, stride-1 dgrad through the forward kernel,
    accumulate epilogues, split-K, parity-decomposed stride-2 dgrad) in the exact configuration the step selects."""
    from kernel_audit import audit
    from rrnet_amd import functional as RF
    from rrnet_amd.datasets.synthetic import synth_batch
    from rrnet_amd.models.rrnet import RRNet
    torch.manual_seed(219)
    model = RRNet(_cfg("hourglass")).cuda().to(memory_format=CL).train()
    for i in range(2):
        model.wh.detect_H_layer[i][0].conv.bias.data.fill_(3.0)
        model.wh.detect_W_layer[i][0].conv.bias.data.fill_(3.0)
    imgs, annos, hms, whs, inds, offs, masks, _ = [t.cuda() if torch.is_tensor(t) else t
                                                   for t in synth_batch(bs, h, w, boxes_per_image=12, seed=219)]
    with audit() as rec:
        outs = model(imgs, k=100)
        hm_l = sum(RF.focal_loss_hm_from_logits(outs[0][i], hms) / 2 for i in range(2))
        wh_l = sum(RF.reg_l1_loss(outs[1][i], masks, inds, whs) / 2 for i in range(2))
        off_l = sum(RF.reg_l1_loss(outs[2][i], masks, inds, offs) / 2 for i in range(2))
        a = annos.clone()
        a[:, :, 2:4] += a[:, :, 0:2]
        s2_l = RF.stage2_reg_loss(outs[3], outs[4], a, 4.0)
        (hm_l + 0.1 * wh_l + off_l + s2_l).backward()
        torch.cuda.synchronize()
    kinds = {}
    for key, err in rec.seen.items():
        kinds.setdefault(key[0], []).append(err)
    print("  ".join("%s:%d (max %.1e)" % (kk, len(v), max(v)) for kk, v in sorted(kinds.items())))
    assert len(rec.seen) > 150 and {"fprop", "dgrad", "wgrad", "bn_apply", "bn_bwd_apply", "bn_bwd_reduce"} <= set(kinds)
    assert not rec.bad, rec.bad[:10]
