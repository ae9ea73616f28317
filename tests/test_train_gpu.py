"""GPU: flat parameter buffer + fused Adam vs torch.optim.Adam (CPU) on the tiny CenterNet, and
gradient equivalence of the in-place (flat-buffer) accumulation path with the autograd-returned
path."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import det_fill

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _cfg():
    return SimpleNamespace(num_classes=10, Model=SimpleNamespace(num_stacks=1, backbone="hourglass_tiny",
                           nms_type_for_stage1="nms", nms_per_class_for_stage1=True))


def _model(seed=11):
    from rrnet_amd.models.centernet import CenterNet
    m = CenterNet(_cfg())
    m.load_state_dict(det_fill({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
    return m.cuda().to(memory_format=CL).train()


def _loss(model, x):
    hms, whs, regs = model(x)
    return (hms[0] ** 2).mean() + whs[0].abs().mean() + (regs[0] ** 2).mean()


def test_flat_grads_match_autograd_grads():
    from rrnet_amd.flat import FlatParams
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0)).cuda()
    m1 = _model()
    _loss(m1, x).backward()
    ref = {k: p.grad.clone() for k, p in m1.named_parameters()}
    m2 = _model()
    fp = FlatParams(m2)
    fp.zero_grad()
    _loss(m2, x).backward()
    for k, p in m2.named_parameters():
        assert p.grad.data_ptr() == p._rr_grad.data_ptr()
        a, b = p.grad.cpu().numpy(), ref[k].cpu().numpy()
        np.testing.assert_allclose(a, b, atol=1e-5 + 1e-3 * np.abs(b).max(), rtol=0)
    # state_dict round trip through the flat views
    sd = {k: v.clone() for k, v in m2.state_dict().items()}
    m3 = _model(seed=12)
    FlatParams(m3)
    m3.load_state_dict(sd)
    for k, v in m3.state_dict().items():
        assert torch.equal(v, sd[k])


def test_fused_adam_matches_torch_adam():
    from rrnet_amd.flat import FlatAdam
    m = _model()
    opt = FlatAdam(m, lr=2.5e-4)
    ref_p = [p.detach().cpu().clone().requires_grad_() for p in m.parameters()]
    ref_opt = torch.optim.Adam(ref_p, lr=2.5e-4)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    for step in range(3):
        opt.zero_grad()
        _loss(m, x).backward()
        for rp, p in zip(ref_p, m.parameters()):
            rp.grad = p.grad.detach().cpu().clone()
        opt.step()
        ref_opt.step()
    for rp, p in zip(ref_p, m.parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), rp.detach().numpy(), atol=2e-6, rtol=1e-5)


def test_rrnet_tiny_overfits_one_batch():
    """The whole training loop end to end (operators/rrnet_operator.py:104-186: forward, the four losses, backward,
    fused Adam, lr schedule, BN statistics) on one fixed synthetic batch: the loss must fall steadily.  A
    sign / scaling error anywhere in the backward kernels shows up here even when it hides inside a tolerance."""
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    cfg.Train.batch_size = 2
    cfg.Train.crop_size = (256, 256)
    cfg.Model.backbone = "hourglass_tiny"
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    torch.manual_seed(219)
    op = RRNetOperator(cfg)
    op.model.train()
    b = op.training_loader.get_batch()
    hist = []
    for step in range(60):
        _, losses = op.train_step(step, (b[0], b[1].clone()) + tuple(b[2:]))
        hist.append([float(v.detach()) for v in losses])
    hist = np.array(hist)
    assert np.isfinite(hist).all()
    first, last = hist[:5, 0].mean(), hist[-5:, 0].mean()
    assert last < 0.6 * first, (first, last)
    assert hist[-5:, 1].mean() < hist[:5, 1].mean()            # the heat-map focal loss itself goes down


def test_training_process_entry_point(tmp_path, monkeypatch, capsys):
    """The reference's entry point itself (operators/rrnet_operator.py:104-186 `training_process`): the step loop, the
    periodic report with generate_bbox + _ext_nms on rank 0, and the checkpoint with reference state_dict keys."""
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    monkeypatch.chdir(tmp_path)
    cfg.Train.batch_size = 2
    cfg.Train.crop_size = (256, 256)
    cfg.Train.iter_num = 4
    cfg.Train.print_interval = 2
    cfg.Train.checkpoint_interval = 5000
    cfg.Model.backbone = "hourglass_tiny"
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    torch.manual_seed(219)
    op = RRNetOperator(cfg)
    op.training_process()
    out = capsys.readouterr().out
    assert "step 1 " in out and "step 3 " in out
    ckp = tmp_path / "log" / cfg.log_prefix / "ckp-3.pth"
    assert ckp.exists()
    sd = torch.load(str(ckp), map_location="cpu")
    assert "backbone.pre_layer.0.weight" in sd and "head_detector.regressor.weight" in sd
    assert sd["backbone.pre_layer.0.weight"].shape == (16, 3, 7, 7)
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    # round trip into a fresh model of the same architecture
    from rrnet_amd.models.rrnet import RRNet
    RRNet(cfg).load_state_dict(sd, strict=True)


def test_flipped_filter_cache_follows_the_parameters():
    """FlatParams keeps flipped / transposed copies of every filter (fp32 and, on demand, bf16) for the stride-1 data gradients,
    refilled by ONE launch per optimizer step (rr_weight_flip_transpose_batch[_bf16]).  The cache must equal the per-layer
    kernel's output, follow a fused Adam step (raw-pointer update) and follow parameters written some other way
    (load_state_dict: the flat buffer's version counter)."""
    from rrnet_amd import _C, ops
    from rrnet_amd.flat import FlatAdam
    m = _model()
    opt = FlatAdam(m, lr=1e-2)
    fp = opt.fp

    def reference(p):
        k, c, r, s = p.shape
        wt = torch.empty(k * c * r * s, dtype=torch.float32, device=p.device)
        _C.check(_C.fn("rr_weight_flip_transpose")(_C.ptr(ops.to_nhwc(p.detach())), _C.ptr(wt), k, c, r, s, _C.stream()), "flip")
        return wt

    def check(tag):
        n = 0
        for p in fp.params:
            if p.dim() != 4:
                assert fp.wt_view(p) is None
                continue
            ref = reference(p)
            assert torch.equal(fp.wt_view(p), ref), (tag, tuple(p.shape))
            w16, wt16 = fp.w16_views(p)
            assert torch.equal(wt16.float(), ref.to(torch.bfloat16).float()), (tag, "wt16", tuple(p.shape))
            assert torch.equal(w16.float(), p.detach().permute(0, 2, 3, 1).reshape(-1).to(torch.bfloat16).float()), (tag, "w16")
            n += 1
        assert n > 10
    check("initial")
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    opt.zero_grad()
    _loss(m, x).backward()
    before = fp.flat.clone()
    opt.step()
    assert not torch.equal(before, fp.flat)
    check("after the fused Adam step")
    sd = {k: v.clone() * 1.5 for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    check("after load_state_dict")


def test_host_fed_loader_hands_over_the_same_batches():
    """The reference's loader hands HOST batches to the step (operators/rrnet_operator.py:121).  HostFedDronesDET (pinned
    pool, copy stream one batch ahead, event hand-over, two device slots) must deliver, bit for bit, the batches the
    device-resident SyntheticDronesDET holds — also while the compute stream is busy and the slots are being recycled."""
    from types import SimpleNamespace
    from rrnet_amd.datasets.synthetic import HostFedDronesDET, synth_batch
    cfg = SimpleNamespace(seed=219, num_classes=10, Train=SimpleNamespace(scale_factor=4))
    ld = HostFedDronesDET(cfg, 2, 256, 320, boxes_per_image=40, pool=3)
    ref = [synth_batch(2, 256, 320, 40, 219 + 1000 * i, 0, 4, 10, "cuda") for i in range(3)]
    keep = []
    for step in range(7):                               # wraps around the pool and around the two slots several times
        b = ld.get_batch()
        r = ref[step % 3]
        # work on the compute stream that READS the batch late: a spin, then a reduction of the frames
        torch.cuda._sleep(2_000_000)
        keep.append((b[0].double().sum(), r[0].double().sum()))
        for got, want in zip(b[:7], r[:7]):
            assert got.shape == want.shape and got.dtype == want.dtype
            assert torch.equal(got, want), step
        assert b[0].is_contiguous(memory_format=torch.channels_last)
        assert b[7] == r[7]
    torch.cuda.synchronize()
    for a, c in keep:
        assert float(a) == float(c)
