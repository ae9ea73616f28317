"""GPU parity of the fused inference tail of the re-regression head (rr_conv1x1_bn_res_relu_avgpool: conv3 1x1 + folded
bn3 + residual + ReLU + global average pool, backbones/resnet.py:46-53 + detectors/fasterrcnn_detector.py:15 of the
reference) against the same composition in torch on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (RoIs, K, N, ph, pw): RoI counts around the 32-RoI workgroup and the 32-row sub-tile, both K variants, narrow N
CASES = [(1, 64, 256, 3, 3), (31, 64, 256, 3, 3), (32, 64, 256, 3, 3), (33, 64, 256, 3, 3), (1000, 64, 256, 3, 3),
         (77, 32, 128, 3, 3), (50, 64, 12, 3, 3), (65, 64, 256, 1, 1), (45, 32, 64, 2, 2), (7, 64, 200, 5, 7)]


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "r%dk%dn%dp%dx%d" % c)
def test_fused_tail_vs_torch(cfg):
    from rrnet_amd import ops
    r, k, n, ph, pw = cfg
    rng = np.random.default_rng(r * 7 + k + n)
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    h, res = mk(r, k, ph, pw), mk(r, n, ph, pw)
    w = mk(n, k, 1, 1) / np.sqrt(k)
    scale, shift = mk(n) * 0.3 + 1.0, mk(n) * 0.2
    ref = F.relu(F.conv2d(h, w) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res).mean(dim=(2, 3), keepdim=True)
    out = ops.conv1x1_bn_res_relu_avgpool(ops.to_nhwc(h.cuda()), ops.to_nhwc(w.cuda()), scale.cuda(), shift.cuda(),
                                          ops.to_nhwc(res.cuda()))
    assert tuple(out.shape) == (r, n, 1, 1)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5 * np.sqrt(k), rtol=1e-4)


def test_fused_tail_rejects_what_it_cannot_take():
    from rrnet_amd import _C, ops
    h = ops.to_nhwc(torch.zeros(4, 48, 3, 3, device="cuda"))
    w = ops.to_nhwc(torch.zeros(256, 48, 1, 1, device="cuda"))
    with pytest.raises(RuntimeError):
        ops.conv1x1_bn_res_relu_avgpool(h, w, torch.ones(256, device="cuda"), torch.zeros(256, device="cuda"),
                                        ops.to_nhwc(torch.zeros(4, 256, 3, 3, device="cuda")))


# (rows, K, N, bias, relu): around the 64-row tile and the persistent grid (512 workgroups = 32768 rows per sweep)
ROWS = [(1, 256, 64, True, True), (63, 256, 64, False, False), (65, 128, 64, True, False), (1000, 256, 40, True, True),
        (32768 + 77, 256, 64, True, True), (70001, 128, 12, False, True)]


@pytest.mark.parametrize("cfg", ROWS, ids=lambda c: "m%dk%dn%d_b%d_r%d" % c)
def test_rows_gemm_vs_torch(cfg):
    """rr_conv1x1_rows (the stage-2 head's conv1 at inference: weights in registers, rows streamed by persistent
    workgroups) against x @ w.T (+ bias, ReLU) in torch on the CPU."""
    from rrnet_amd import _C
    m, k, n, use_bias, relu = cfg
    rng = np.random.default_rng(m + k + n)
    x = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(n).astype(np.float32)) if use_bias else None
    ref = x.double() @ w.double().t()
    if use_bias:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp_min(0)
    xd, wd = x.cuda(), w.cuda()
    bd = b.cuda() if use_bias else None
    y = torch.full((m, n), float("nan"), device="cuda")
    _C.check(_C.fn("rr_conv1x1_rows")(_C.ptr(xd), _C.ptr(wd), _C.ptr(bd), _C.ptr(y), m, k, n, int(relu), _C.stream()),
             "rr_conv1x1_rows")
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().numpy(), atol=2e-5 * np.sqrt(k), rtol=1e-4)


def test_conv_fprop_takes_the_rows_kernel_for_the_head_shape():
    """rr_conv_fprop routes a 1x1 256 -> 64 convolution on >= 65536 rows without statistics to the rows kernel: same
    values as the implicit-GEMM path (taken when statistics are requested) up to summation order."""
    from rrnet_amd import ops
    g = torch.Generator().manual_seed(9)
    x = ops.to_nhwc(torch.randn(7300, 256, 3, 3, generator=g).cuda())
    w = ops.to_nhwc((torch.randn(64, 256, 1, 1, generator=g) / 16).cuda())
    b = torch.randn(64, generator=g).cuda()
    y = ops.conv_fprop(x, w, b, 1, (0, 0), True)
    y2, _ = ops.conv_fprop(x, w, None, 1, (0, 0), False, want_stats=True)
    ref = torch.relu(y2 + b.view(1, -1, 1, 1))
    assert (y - ref).abs().max().item() <= 1e-4
