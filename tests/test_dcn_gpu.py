"""GPU parity: DCNv2 gather-GEMM forward and backward (through the C ABI) against the oracle
(oracle/dcn.py, pinned by the reference's own analytic checks) on seeded inputs, plus the reference's
zero-offset identity check (ext/dcn/test.py:32-67) run on the HIP op itself.  fp32, |err| <= 1e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last

# N, C, H, W, K, ksize, stride, pad, dil, dg
CASES = [
    (2, 8, 9, 11, 12, 3, 1, 1, 1, 1),
    (1, 64, 12, 12, 64, 3, 1, 1, 1, 1),
    (2, 64, 10, 10, 40, 3, 2, 1, 1, 2),        # two deformable groups (32 channels each), stride 2
    (1, 16, 8, 8, 8, 3, 1, 2, 2, 1),           # dilation 2
    (1, 256, 6, 6, 256, 3, 1, 1, 1, 1),
]


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "n%dc%dh%dw%dk%d_k%ds%dp%dd%dg%d" % c)
def test_dcn_forward_backward_vs_oracle(cfg):
    from oracle import dcn as odcn
    from rrnet_amd.functional import dcn_v2_conv
    n, c, h, w, k, ks, stride, pad, dil, dg = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    p = (h + 2 * pad - (dil * (ks - 1) + 1)) // stride + 1
    q = (w + 2 * pad - (dil * (ks - 1) + 1)) // stride + 1
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, 2 * dg * ks * ks, p, q, generator=g) * 1.5
    off = off + 0.31 * (off.round() == off)
    mask = torch.sigmoid(torch.randn(n, dg * ks * ks, p, q, generator=g))
    wt = torch.randn(k, c, ks, ks, generator=g) / np.sqrt(c * ks * ks)
    b = torch.randn(k, generator=g)
    ref_in = [t.clone().requires_grad_() for t in (x, off, mask, wt, b)]
    ref = odcn.dcn_v2_conv(*ref_in, stride, pad, dil, dg)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    dev_in = [x.cuda().contiguous(memory_format=CL).requires_grad_(), off.cuda().contiguous(memory_format=CL).requires_grad_(),
              mask.cuda().contiguous(memory_format=CL).requires_grad_(), wt.cuda().contiguous(memory_format=CL).requires_grad_(),
              b.cuda().requires_grad_()]
    out = dcn_v2_conv(*dev_in, stride, pad, dil, dg)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=1e-3, rtol=1e-3)
    out.backward(gy.cuda())
    for name, a, r in zip(("dx", "doffset", "dmask", "dw", "db"), dev_in, ref_in):
        ra = r.grad.numpy()
        tol = 1e-3 * max(1.0, np.abs(ra).max())
        np.testing.assert_allclose(a.grad.cpu().numpy(), ra, atol=tol, rtol=1e-3, err_msg=name)


def test_reference_zero_offset_identity_on_hip():
    from rrnet_amd.ext.dcn.dcn_v2 import DCNv2
    N, C, H, W = 2, 8, 6, 6
    m = DCNv2(C, C, (3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    m.weight.data.zero_()
    for p in range(C):
        m.weight.data[p, p, 1, 1] = 1.0
    m.bias.data.zero_()
    m = m.cuda().to(memory_format=CL)
    x = torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(0)).cuda()
    offset = torch.zeros(N, 18, H, W, device="cuda")
    mask = torch.sigmoid(torch.zeros(N, 9, H, W, device="cuda"))
    out = m(x, offset, mask) * 2
    assert (x - out).abs().max().item() < 1e-6


def test_dcn_module_matches_oracle_and_trains():
    from oracle import dcn as odcn
    from rrnet_amd.ext.dcn.dcn_v2 import DCN
    g = torch.Generator().manual_seed(5)
    m = DCN(32, 32, 3, stride=1, padding=1)
    m.conv_offset_mask.weight.data.normal_(0, 0.05, generator=g)
    m.conv_offset_mask.bias.data.normal_(0, 0.1, generator=g)
    x = torch.randn(2, 32, 10, 10, generator=g)
    ref = odcn.dcn_forward(x, m.weight.data, m.bias.data, m.conv_offset_mask.weight.data, m.conv_offset_mask.bias.data)
    md = m.cuda().to(memory_format=CL)
    xd = x.cuda().contiguous(memory_format=CL).requires_grad_()
    out = md(xd)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.numpy(), atol=1e-3, rtol=1e-3)
    out.square().mean().backward()
    assert xd.grad is not None and md.conv_offset_mask.weight.grad.abs().sum().item() > 0


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "n%dc%dh%dw%dk%d_k%ds%dp%dd%dg%d" % c)
def test_dcn_forward_bf16_operands(cfg):
    """BASELINE config 4: bf16 matrix operands, fp32 accumulation.  Against the fp32 oracle evaluated on
    bf16-ROUNDED weights (exact for that operand) the remaining error is the rounding of the sampled columns:
    2^-9 relative per term, far below 2e-2 of the output scale; the zero-offset identity stays exact."""
    from oracle import dcn as odcn
    from rrnet_amd.functional import dcn_v2_conv
    n, c, h, w, k, ks, stride, pad, dil, dg = cfg
    g = torch.Generator().manual_seed(sum(cfg) + 1)
    p = (h + 2 * pad - (dil * (ks - 1) + 1)) // stride + 1
    q = (w + 2 * pad - (dil * (ks - 1) + 1)) // stride + 1
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, 2 * dg * ks * ks, p, q, generator=g) * 1.5
    mask = torch.sigmoid(torch.randn(n, dg * ks * ks, p, q, generator=g))
    wt = (torch.randn(k, c, ks, ks, generator=g) / np.sqrt(c * ks * ks)).bfloat16().float()
    b = torch.randn(k, generator=g)
    ref = odcn.dcn_v2_conv(x, off, mask, wt, b, stride, pad, dil, dg)
    dev = [t.cuda().contiguous(memory_format=CL) for t in (x, off, mask, wt)]
    out = dcn_v2_conv(*dev, b.cuda(), stride, pad, dil, dg, bf16=True)
    err = (out.cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-2 * scale, (err, scale)
    # and it really is a different (lower precision) path than the fp32 kernel, not an alias of it
    out32 = dcn_v2_conv(*dev, b.cuda(), stride, pad, dil, dg, bf16=False)
    assert (out32.cpu() - ref).abs().max().item() < err or err == 0.0
    # backward of the bf16 forward is the fp32 backward
    xg = dev[0].clone().requires_grad_()
    dcn_v2_conv(xg, dev[1], dev[2], dev[3], b.cuda(), stride, pad, dil, dg, bf16=True).sum().backward()
    assert torch.isfinite(xg.grad).all()
