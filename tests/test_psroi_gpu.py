"""GPU parity of the deformable PS-RoI pooling (rr_dcn_psroi_fwd / _bwd through ext.dcn.dcn_v2.dcn_v2_pooling) against
the oracle's line-by-line restatement of the reference kernels (oracle/psroi.py; parity unpinned: the reference op is
CUDA-only), forward, count, data gradient and trans gradient; plus the DCNPooling module."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last

# no_trans, classes, out_dim, group, pooled, part, spp, trans_std, scale
CASES = [
    (True, 1, 8, 1, 3, 3, 2, 0.0, 1.0),
    (False, 1, 8, 1, 3, 3, 4, 0.1, 1.0),
    (False, 2, 8, 1, 7, 7, 2, 0.2, 0.5),
    (False, 4, 4, 2, 4, 2, 3, 0.1, 0.25),       # group_size 2 (16 input channels), part_size != pooled_size
]


@pytest.mark.parametrize("cfg", CASES)
def test_psroi_forward_backward_vs_oracle(cfg):
    from oracle import psroi as op
    from rrnet_amd.ext.dcn.dcn_v2 import dcn_v2_pooling
    no_trans, classes, out_dim, gs, P, part, spp, tstd, scale = cfg
    rng = np.random.default_rng(sum(int(v * 10) for v in cfg[1:]))
    H, W = 19, 23
    x = rng.normal(0, 1, (2, out_dim * gs * gs, H, W)).astype(np.float32)
    sc = 1.0 / scale
    rois = np.array([[0, 2.2 * sc, 3.1 * sc, 9.7 * sc, 10.2 * sc], [1, 4.0 * sc, 1.0 * sc, 20.6 * sc, 16.4 * sc],
                     [1, 0.3 * sc, 0.2 * sc, 5.5 * sc, 4.9 * sc], [0, 15 * sc, 12 * sc, 30 * sc, 25 * sc]], np.float32)
    trans = rng.normal(0, 1, (rois.shape[0], 2 * classes, part, part)).astype(np.float32)
    kw = dict(no_trans=no_trans, scale=scale, out_dim=out_dim, gs=gs, P=P, part=part, spp=spp, trans_std=tstd)
    r_out, r_cnt = op.psroi_forward(x, rois, trans, **kw)
    dout = rng.normal(0, 1, r_out.shape).astype(np.float32)
    r_dx, r_dt = op.psroi_backward(dout, x, rois, trans, r_cnt, **kw)
    xd = torch.from_numpy(x).cuda().contiguous(memory_format=CL).requires_grad_()
    td = torch.from_numpy(trans).cuda().requires_grad_()
    out = dcn_v2_pooling(xd, torch.from_numpy(rois).cuda(), td if not no_trans else xd.new(), scale, P, out_dim, no_trans, gs,
                         part, spp, tstd)
    np.testing.assert_allclose(out.detach().cpu().numpy(), r_out, atol=1e-5, rtol=1e-5)
    out.backward(torch.from_numpy(dout).cuda())
    np.testing.assert_allclose(xd.grad.cpu().numpy(), r_dx, atol=1e-5, rtol=1e-4)
    if not no_trans:
        np.testing.assert_allclose(td.grad.cpu().numpy(), r_dt, atol=2e-5, rtol=1e-4)


def test_dcn_pooling_module():
    """DCNPooling (dcn_v2.py:222-300): with the zero-initialised last FC layer the offsets are 0 and the mask 0.5, so
    the module returns half the plain pooling; with a trained-like last layer the result equals the explicit composition."""
    from rrnet_amd.ext.dcn.dcn_v2 import DCNPooling, DCNv2Pooling, dcn_v2_pooling
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 16, 20, 20, generator=g).cuda().contiguous(memory_format=CL)
    rois = torch.tensor([[0, 1.0, 2.0, 11.0, 12.0], [1, 3.0, 3.0, 17.0, 15.0]]).cuda()
    plain = DCNv2Pooling(1.0, 3, 16, True).cuda()(x, rois, x.new())
    m = DCNPooling(1.0, 3, 16, False, trans_std=0.1, deform_fc_dim=64).cuda()
    out = m(x, rois)
    np.testing.assert_allclose(out.detach().cpu().numpy(), 0.5 * plain.cpu().numpy(), atol=1e-6)
    m.offset_mask_fc[4].weight.data.normal_(0, 0.05, generator=torch.Generator(device="cuda").manual_seed(1))
    out = m(x, rois)
    om = m.offset_mask_fc(plain.reshape(2, -1)).view(2, 3, 3, 3)
    o1, o2, mk = torch.chunk(om, 3, dim=1)
    exp = dcn_v2_pooling(x, rois, torch.cat((o1, o2), 1), 1.0, 3, 16, False, 1, 3, 4, 0.1) * torch.sigmoid(mk)
    np.testing.assert_allclose(out.detach().cpu().numpy(), exp.detach().cpu().numpy(), atol=1e-6)
    out.sum().backward()
    assert m.offset_mask_fc[0].weight.grad.abs().sum().item() > 0
