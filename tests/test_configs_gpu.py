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
    from rrnet_amd.datasets.synthetic import synth_batch
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
    from rrnet_amd.datasets.synthetic import synth_batch
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
