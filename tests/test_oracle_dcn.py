"""Pins the oracle's DCNv2 restatement (oracle/dcn.py).  The reference op is CUDA-only (no CPU path, not
buildable here), so it is pinned by the reference's own analytic checks (ext/dcn/test.py) and by
independent torch formulations instead of reference outputs."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import dcn


def test_reference_zero_offset_identity_check():
    """ext/dcn/test.py:32-67: zero offsets, mask = sigmoid(0), identity weight  =>  2 * out == in."""
    N, C, H, W, kh, kw = 2, 2, 4, 4, 3, 3
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, C, H, W, generator=g)
    weight = torch.zeros(C, C, kh, kw)
    for p in range(C):
        weight[p, p, kh // 2, kw // 2] = 1.0          # conv_identify (test.py:18-29)
    offset = torch.zeros(N, 2 * kh * kw, H, W)
    mask = torch.sigmoid(torch.zeros(N, kh * kw, H, W))
    out = dcn.dcn_v2_conv(x, offset, mask, weight, torch.zeros(C), 1, 1, 1, 1)
    assert (x - 2 * out).abs().max() < 1e-10


def test_zero_offset_unit_mask_is_conv2d():
    g = torch.Generator().manual_seed(1)
    for (stride, pad, dil, H, W) in [(1, 1, 1, 7, 9), (2, 1, 1, 8, 8), (1, 2, 2, 9, 7), (2, 0, 1, 9, 9)]:
        x = torch.randn(2, 4, H, W, generator=g)
        w = torch.randn(6, 4, 3, 3, generator=g)
        b = torch.randn(6, generator=g)
        ref = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
        P, Q = ref.shape[2:]
        out = dcn.dcn_v2_conv(x, torch.zeros(2, 18, P, Q), torch.ones(2, 9, P, Q), w, b, stride, pad, dil, 1)
        np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=1e-5)


def test_fractional_offsets_match_grid_sample():
    """One tap, fractional offsets: the sampled value equals F.grid_sample(bilinear, zeros padding,
    align_corners=True) wherever the reference's window rule (-1 < h < H) and grid_sample agree (interior)."""
    g = torch.Generator().manual_seed(2)
    H = W = 8
    x = torch.randn(1, 3, H, W, generator=g)
    off = (torch.rand(1, 2, H, W, generator=g) - 0.5) * 3
    w = torch.ones(1, 3, 1, 1)
    out = dcn.dcn_v2_conv(x, off, torch.ones(1, 1, H, W), w, None, 1, 0, 1, 1)          # sum over channels
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    hy, wx = ys + off[0, 0], xs + off[0, 1]
    grid = torch.stack((wx / (W - 1) * 2 - 1, hy / (H - 1) * 2 - 1), dim=-1).unsqueeze(0)
    ref = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True).sum(1, keepdim=True)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=1e-5)


def test_reference_gradcheck_configuration():
    """ext/dcn/test.py:69-97: gradcheck with N,C,H,W = 2,2,4,4, 3x3, eps 1e-3, atol 1e-4, rtol 1e-2 (float64)."""
    g = torch.Generator().manual_seed(3)
    N, C, K, H, W = 2, 2, 2, 4, 4
    x = (torch.rand(N, C, H, W, generator=g, dtype=torch.float64) * 0.01).requires_grad_()
    offset = (torch.randn(N, 18, H, W, generator=g, dtype=torch.float64) * 2)
    offset = (offset + 0.37 * (offset.round() == offset)).requires_grad_()      # keep away from integer kinks
    mask = torch.sigmoid(torch.rand(N, 9, H, W, generator=g, dtype=torch.float64)).detach().requires_grad_()
    weight = torch.randn(K, C, 3, 3, generator=g, dtype=torch.float64).requires_grad_()
    bias = torch.rand(K, generator=g, dtype=torch.float64).requires_grad_()
    assert torch.autograd.gradcheck(lambda a, o, m, w, b: dcn.dcn_v2_conv(a, o, m, w, b, 1, 1, 1, 1),
                                    (x, offset, mask, weight, bias), eps=1e-3, atol=1e-4, rtol=1e-2)


def test_dcn_module_starts_as_half_conv():
    """DCN.init_offset zeroes conv_offset_mask (dcn_v2.py:108-112): offsets 0, mask sigmoid(0) = 0.5."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 4, 6, 6, generator=g)
    w = torch.randn(5, 4, 3, 3, generator=g)
    out = dcn.dcn_forward(x, w, None, torch.zeros(27, 4, 3, 3), torch.zeros(27), 1, 1, 1, 1)
    np.testing.assert_allclose(out.numpy(), 0.5 * F.conv2d(x, w, None, padding=1).numpy(), atol=1e-5)
