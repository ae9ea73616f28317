"""GPU parity of the API-completion layer against goldens produced by the reference's own methods
(tests/golden/helpers.npz, tools/gen_goldens.py:g11_helpers): CenterNetOperator.transform_bbox / _ctnet_nms /
save_result (operators/centernet_operator.py:152-249) and RRNet._topk / _gather_feat / _transpose_and_gather_feat /
nms (models/rrnet.py:56-115); plus the CenterNet operator's train step (BASELINE configs[0] through the operator)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _z(golden_dir):
    return np.load(os.path.join(golden_dir, "helpers.npz"), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.array(a))


def test_centernet_operator_decode_peak_save_vs_reference_golden(golden_dir, tmp_path):
    from rrnet_amd.operators.centernet_operator import CenterNetOperator
    z = _z(golden_dir)
    op = CenterNetOperator.__new__(CenterNetOperator)
    hm, wh, off, k = T(z["hm"]).cuda(), T(z["wh"]).cuda(), T(z["offset"]).cuda(), int(z["k"])
    for pred, key in ((op.transform_bbox(hm, wh, off, k=k, scale_factor=4), "ct_pred"),
                      (op.transform_bbox(hm, wh, None, k=k, scale_factor=4), "ct_pred_nooff")):
        ref = z[key]
        assert tuple(pred.shape) == ref.shape
        np.testing.assert_array_equal(pred[:, 5].cpu().numpy(), ref[:, 5])              # classes (+1), tie-free input
        np.testing.assert_allclose(pred[:, 4].cpu().numpy(), ref[:, 4], atol=1e-6)
        np.testing.assert_allclose(pred[:, :4].cpu().numpy(), ref[:, :4], atol=1e-4, rtol=1e-5)
        assert (ref[:, 2] < 0).any()                                                     # negative widths survive: no clamp
    # `_ctnet_nms` on ready scores: no transcendental in the way -> bit-exact map, plateaus and borders included
    got = op._ctnet_nms(T(z["heat"]).cuda()).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), z["heat_nms"].view(np.uint32))
    path = str(tmp_path / "r.txt")
    op.save_result(path, T(z["save_rows"]).clone())
    assert open(path, "rb").read() == z["save_text"].tobytes()


def test_rrnet_helper_methods_vs_reference_golden(golden_dir):
    from rrnet_amd.models.rrnet import RRNet
    z = _z(golden_dir)
    net = RRNet.__new__(RRNet)
    torch.nn.Module.__init__(net)
    hm, wh, k = T(z["hm"]).cuda(), T(z["wh"]).cuda(), int(z["k"])
    sc, inds, cl, ys, xs = net._topk(torch.sigmoid(hm), k)
    np.testing.assert_array_equal(inds.cpu().numpy(), z["topk_inds"])
    np.testing.assert_array_equal(cl.cpu().numpy(), z["topk_clses"])
    np.testing.assert_array_equal(ys.cpu().numpy(), z["topk_ys"])
    np.testing.assert_array_equal(xs.cpu().numpy(), z["topk_xs"])
    np.testing.assert_allclose(sc.cpu().numpy(), z["topk_score"], atol=1e-6)
    got = net._transpose_and_gather_feat(wh, inds)
    np.testing.assert_array_equal(got.cpu().numpy(), z["tg_feat"])
    bbox = T(z["nms_in"]).cuda()
    net.nms_per_class, net.nms_type = True, 'nms'
    np.testing.assert_array_equal(net.nms(bbox).cpu().numpy(), z["nms_out"])
    net.nms_type = 'soft_nms'
    out = net.nms(bbox).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), z["softnms_out"].view(np.uint32))         # Soft-NMS rows bit-exact
    net.nms_per_class = False
    from oracle import ops as oo
    np.testing.assert_array_equal(net.nms(bbox).cpu().numpy(), oo.stage1_nms(bbox.cpu(), 'soft_nms', False).numpy())


def test_centernet_operator_train_step_vs_oracle():
    """BASELINE configs[0] through the operator: CenterNet + hourglass-tiny, criterion values against the oracle on
    the operator's own batch, then the full train step (losses finite, parameters move)."""
    from oracle import model as om, ops as oo
    from rrnet_amd.configs.centernet_config import Config as cfg
    from rrnet_amd.operators.centernet_operator import CenterNetOperator
    cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = 2, (128, 128), "hourglass_tiny"
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    torch.manual_seed(cfg.seed)
    op = CenterNetOperator(cfg)
    op.model.train()
    batch = op.training_loader.get_batch()
    imgs, _annos, hms, whs, inds, offs, masks, _ = batch
    sd = {k: v.detach().cpu().clone() for k, v in op.model.module.state_dict().items()}
    with torch.no_grad():
        r = om.centernet_forward(om.Params(sd, True), imgs.cpu())
        ref = [sum(oo.hm_loss_from_logits(r[0][i], hms.cpu()) / 2 for i in range(2)),
               sum(oo.reg_l1_loss(r[1][i], masks.cpu(), inds.cpu(), whs.cpu()) / 2 for i in range(2)),
               sum(oo.reg_l1_loss(r[2][i], masks.cpu(), inds.cpu(), offs.cpu()) / 2 for i in range(2))]
    before = op.model.flat.flat.clone()
    outs, losses = op.train_step(0, batch)
    got = [float(v.detach()) for v in losses[1:]]
    np.testing.assert_allclose(got, [float(v) for v in ref], rtol=1e-3, atol=1e-3)
    assert (op.model.flat.flat != before).float().mean().item() > 0.9
    pred = op.transform_bbox(outs[0][1], outs[1][1], outs[2][1], scale_factor=4)
    assert pred.dim() == 2 and pred.size(1) == 6
