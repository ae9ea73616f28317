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
    (2, 256, 20, 24, 256, 3, 1, 1, 1, 1),      # several pixel tiles x 2 channel tiles x 2 filter tiles (fused backward)
    (1, 128, 17, 15, 132, 3, 2, 1, 1, 1),      # ragged pixel / filter tiles, stride 2
    (1, 256, 9, 9, 64, 3, 1, 1, 1, 2),         # two deformable groups of 128 channels: fused backward, per-group geometry
    (1, 32, 12, 20, 64, 3, 1, 2, 2, 1),        # window kernels with dilation 2 (wider window, smaller margin fits)
    (1, 64, 11, 9, 64, 5, 1, 2, 1, 1),         # 5x5: forward on the window kernel, gradients on the L2-gather kernels
    (2, 32, 17, 33, 36, 3, 1, 1, 1, 1),        # ragged 8x16 blocks, K = 36 (one partial 32-filter slab)
    (1, 64, 10, 18, 48, 3, 1, 1, 1, 2),        # two deformable groups of 32 channels: window kernels only (fused backward)
    (1, 128, 9, 17, 32, 3, 1, 1, 1, 4),        # four groups of 32 channels
    (1, 128, 14, 14, 32, 3, 1, 3, 3, 2),       # groups of 64 channels, dilation 3: the dgrad window does not fit -> column path
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


def test_dcn_fused_backward_equals_column_path():
    """The fused backward (columns in registers / LDS inside the two GEMM kernels) against the reference-structured
    column path (im2col -> GEMM, GEMM -> col2im) of the same library on a multi-tile layer: same gradients up to the
    summation order."""
    from rrnet_amd import functional as RF
    g = torch.Generator().manual_seed(21)
    n, c, h, w, k = 2, 256, 24, 40, 256
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, 18, h, w, generator=g) * 2.0
    mask = torch.sigmoid(torch.randn(n, 9, h, w, generator=g))
    wt = torch.randn(k, c, 3, 3, generator=g) / 48.0
    gy = torch.randn(n, k, h, w, generator=g).cuda()
    grads = {}
    saved = RF.DCN_FUSED_BWD
    try:
        for fused in (True, False):
            RF.DCN_FUSED_BWD = fused
            ins = [t.cuda().contiguous(memory_format=CL).requires_grad_() for t in (x, off, mask, wt)]
            RF.dcn_v2_conv(*ins, None, 1, 1, 1, 1).backward(gy)
            grads[fused] = [t.grad.cpu() for t in ins]
    finally:
        RF.DCN_FUSED_BWD = saved
    for name, a, b in zip(("dx", "doffset", "dmask", "dw"), grads[True], grads[False]):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-4 * scale, (name, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("shape", [(2, 256, 24, 40, 256, 1), (1, 64, 19, 21, 96, 1), (1, 256, 16, 16, 64, 2)])
def test_dcn_bf16_dgrad_equals_fp32_dgrad_on_rounded_operands(shape):
    """rr_dcn_dgrad_bf16 (bf16 matrix operands, d input pre-summed in an LDS window) computes exactly what the fp32
    kernel computes when dY and W are rounded to bf16 beforehand: same products, fp32 accumulation — only the summation
    order differs.  Covers ragged 8x16 pixel blocks, several blocks per image, offsets beyond the window margin
    (sigma 2.5: a few percent of the corners take the direct global path) and two deformable groups."""
    from rrnet_amd import ops
    n, c, h, w, k, dg = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, 18 * dg, h, w, generator=g) * 2.5
    mask = torch.sigmoid(torch.randn(n, 9 * dg, h, w, generator=g))
    wt = (torch.randn(k, c, 3, 3, generator=g) / 48.0).bfloat16().float()
    dy = torch.randn(n, k, h, w, generator=g).bfloat16().float()
    dev = [ops.to_nhwc(t.cuda()) for t in (x, off, mask, wt, dy)]
    ref = ops.dcn_dgrad(*dev, 1, (1, 1), 1, dg, bf16=False)
    got = ops.dcn_dgrad(*dev, 1, (1, 1), 1, dg, bf16=True)          # rr_dcn_dgrad_bf16_packed: the LDS-DMA sweep (K % 32 == 0)
    for name, a, b in zip(("dx", "doffset", "dmask"), got, ref):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 5e-5 * scale, (name, (a - b).abs().max().item(), scale)
    # the register-staged sweep (rr_dcn_dgrad_bf16_ws: what layers with other K take) agrees with it
    saved = ops._DCN_WPACK
    ops._DCN_WPACK = False
    try:
        old = ops.dcn_dgrad(*dev, 1, (1, 1), 1, dg, bf16=True)
    finally:
        ops._DCN_WPACK = saved
    for name, a, b in zip(("dx", "doffset", "dmask"), got, old):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 5e-5 * scale, (name, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("shape", [(2, 256, 24, 40, 256, 1, 1.0), (1, 64, 19, 21, 96, 1, 2.5), (1, 256, 16, 16, 64, 2, 2.5),
                                   (1, 32, 40, 33, 260, 1, 1.0)])
def test_dcn_bf16_wgrad_equals_gemm_of_rounded_operands(shape):
    """rr_dcn_wgrad_bf16 = dY^T x columns with BOTH operands rounded to bf16 (the columns after the fp32 bilinear blend and
    mask, as in rr_dcn_fwd_bf16) and fp32 accumulation: checked against a float64 matmul of the rounded operands (columns
    from rr_dcn_im2col).  Covers ragged pixel blocks, several splits, offsets beyond the window margin, two deformable
    groups, K not a multiple of 32 / above one 256-filter tile, accumulation into a non-zero dw."""
    import os
    from rrnet_amd import ops
    if int(os.environ.get("RR_DCN_WINDOW", "3")) <= 0:
        pytest.skip("without the LDS window rr_dcn_wgrad_bf16 runs the fp32-operand kernel")
    n, c, h, w, k, dg, sigma = shape
    g = torch.Generator().manual_seed(sum(int(v) for v in shape))
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, 18 * dg, h, w, generator=g) * sigma
    mask = torch.sigmoid(torch.randn(n, 9 * dg, h, w, generator=g))
    dy = torch.randn(n, k, h, w, generator=g)
    xd, od, md, dyd = [ops.to_nhwc(t.cuda()) for t in (x, off, mask, dy)]
    col = ops.dcn_im2col(xd, od, md, 3, 3, 1, (1, 1), 1, dg)                     # [1, 9c, M, 1] logical, (tap, c) minor
    colr = col.permute(0, 2, 3, 1).reshape(-1, 9 * c).bfloat16().double()
    dyr = dyd.permute(0, 2, 3, 1).reshape(-1, k).bfloat16().double()
    ref = (dyr.t() @ colr).reshape(k, 3, 3, c).permute(0, 3, 1, 2)               # logical [K, C, R, S]
    base = torch.randn(k, c, 3, 3, generator=g) * 0.1
    dw = ops.to_nhwc(base.cuda())
    ops.dcn_wgrad(xd, od, md, dyd, dw, 1, (1, 1), 1, dg, bf16=True)
    got = dw.double() - base.cuda().double()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 3e-5 * scale, ((got - ref).abs().max().item(), scale)
    # the same through rr_dcn_wgrad_bf16_img: dY's bf16 image feeds the operand by LDS-DMA (K % 32 == 0; other K fall back inside)
    dwi = ops.to_nhwc(base.cuda())
    ops.dcn_wgrad(xd, od, md, dyd, dwi, 1, (1, 1), 1, dg, bf16=True, dy_img=ops.bf16_of(dyd))
    goti = dwi.double() - base.cuda().double()
    assert (goti - ref).abs().max().item() <= 3e-5 * scale, ((goti - ref).abs().max().item(), scale)
    # and it is the fp32 kernel's result up to the operand rounding
    dw32 = ops.zeros_nhwc(k, c, 3, 3, device="cuda")
    ops.dcn_wgrad(xd, od, md, dyd, dw32, 1, (1, 1), 1, dg)
    assert (got - dw32.double()).abs().max().item() <= 2e-2 * scale


def test_dcn_bf16_dgrad_fixed_point_window_keeps_local_precision():
    """The LDS window of rr_dcn_dgrad_bf16 accumulates in fixed point with one power-of-two scale per (8x16 pixel block,
    channel), derived from that block's largest |dcol| x |mask| of the channel: a block with huge output gradients must not
    cost precision anywhere else.  dY is 1e6 x larger in the first block of frame 0 than in the rest of the batch;
    rows far from that block are compared at THEIR scale; masks > 1 (allowed by dcn_v2_conv) enter the bound too."""
    from rrnet_amd import ops
    n, c, h, w, k = 2, 64, 48, 48, 64
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, 18, h, w, generator=g)
    mask = torch.rand(n, 9, h, w, generator=g) * 3.0
    wt = (torch.randn(k, c, 3, 3, generator=g) / 24.0).bfloat16().float()
    dy = (torch.randn(n, k, h, w, generator=g) * 1e-3)
    dy[0, :, :8, :16] *= 1e6
    dy = dy.bfloat16().float()
    dev = [ops.to_nhwc(t.cuda()) for t in (x, off, mask, wt, dy)]
    ref = ops.dcn_dgrad(*dev, 1, (1, 1), 1, 1, bf16=False)
    got = ops.dcn_dgrad(*dev, 1, (1, 1), 1, 1, bf16=True)
    for name, a, b in zip(("dx", "doffset", "dmask"), got, ref):
        far_a, far_b = a[1], b[1]                                  # the other frame: untouched by the loud block
        scale = far_b.abs().max().item()
        assert 0 < scale < 1.0
        assert (far_a - far_b).abs().max().item() <= 5e-5 * scale, (name, (far_a - far_b).abs().max().item(), scale)
        loud = b[0].abs().max().item()
        assert loud > 100 * scale
        assert (a[0] - b[0]).abs().max().item() <= 5e-5 * loud, (name, (a[0] - b[0]).abs().max().item(), loud)
    # the fp32 kernel keeps d input in the same fixed-point window: pin it against the column path (im2col-free GEMM +
    # rr_dcn_col2im, plain float atomics) of the library on the quiet frame, at the quiet frame's scale
    from rrnet_amd import functional as RF
    saved = RF.DCN_FUSED_BWD
    try:
        RF.DCN_FUSED_BWD = False
        ins = [t.clone().requires_grad_() for t in dev[:4]]
        RF.dcn_v2_conv(*ins, None, 1, 1, 1, 1).backward(dev[4])
    finally:
        RF.DCN_FUSED_BWD = saved
    for name, a, b in zip(("dx", "doffset", "dmask"), ref, [t.grad for t in ins[:3]]):
        scale = b[1].abs().max().item()
        assert (a[1] - b[1]).abs().max().item() <= 5e-5 * scale, (name, (a[1] - b[1]).abs().max().item(), scale)


@pytest.mark.parametrize("bf16", [False, True])
def test_dcn_dgrad_fixed_point_scale_is_per_channel(bf16):
    """One input channel with 1e7 x larger weights (hence column gradients) must not cost its neighbours in the same
    32-channel chunk any precision: the window's fixed-point scale is per channel.  Reference = the column path
    (float atomics, every channel with its own exponent), on bf16-rounded operands for the bf16 kernel."""
    from rrnet_amd import functional as RF
    from rrnet_amd import ops
    n, c, h, w, k = 1, 64, 24, 32, 64
    g = torch.Generator().manual_seed(123)
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, 18, h, w, generator=g)
    mask = torch.sigmoid(torch.randn(n, 9, h, w, generator=g))
    wt = torch.randn(k, c, 3, 3, generator=g) / 24.0
    wt[:, 5] *= 1e7                                              # channel 5 is loud, channels 0..31 share its chunk
    dy = torch.randn(n, k, h, w, generator=g)
    if bf16:
        wt, dy = wt.bfloat16().float(), dy.bfloat16().float()
    dev = [ops.to_nhwc(t.cuda()) for t in (x, off, mask, wt, dy)]
    dx = ops.dcn_dgrad(*dev, 1, (1, 1), 1, 1, bf16=bf16)[0]
    saved = RF.DCN_FUSED_BWD
    try:
        RF.DCN_FUSED_BWD = False
        xin = dev[0].clone().requires_grad_()
        RF.dcn_v2_conv(xin, dev[1], dev[2], dev[3], None, 1, 1, 1, 1).backward(dev[4])
    finally:
        RF.DCN_FUSED_BWD = saved
    ref = xin.grad
    quiet = [i for i in range(32) if i != 5]
    scale_q = ref[:, quiet].abs().max().item()
    assert ref[:, 5].abs().max().item() > 1e5 * scale_q
    assert (dx[:, quiet] - ref[:, quiet]).abs().max().item() <= 5e-5 * scale_q
    assert (dx[:, 5] - ref[:, 5]).abs().max().item() <= 5e-5 * ref[:, 5].abs().max().item()


@pytest.mark.parametrize("bf16", [False, True])
def test_dcn_dgrad_all_samples_on_one_pixel(bf16):
    """Offsets that send EVERY sample of the image to the same fractional position pile 128 x 9 contributions per block
    onto four input pixels, all with the same sign (dY > 0, W > 0): the fixed-point window must size its scale by the
    number of hits it counted, not by the ~36 of well-behaved offsets (an int32 would wrap around otherwise)."""
    from rrnet_amd import functional as RF
    from rrnet_amd import ops
    n, c, h, w, k = 1, 32, 16, 32, 32
    g = torch.Generator().manual_seed(31)
    x = torch.randn(n, c, h, w, generator=g)
    ty, tx = 7.3, 9.6
    off = torch.zeros(n, 18, h, w)
    for tap in range(9):
        ti, tj = tap // 3, tap % 3
        pp = torch.arange(h).view(h, 1).float() - 1 + ti
        qq = torch.arange(w).view(1, w).float() - 1 + tj
        off[0, 2 * tap] = ty - pp
        off[0, 2 * tap + 1] = tx - qq
    mask = torch.rand(n, 9, h, w, generator=g) * 0.5 + 0.5
    wt = torch.rand(k, c, 3, 3, generator=g) + 0.5
    dy = torch.rand(n, k, h, w, generator=g) + 0.5
    if bf16:
        wt, dy = wt.bfloat16().float(), dy.bfloat16().float()
    dev = [ops.to_nhwc(t.cuda()) for t in (x, off, mask, wt, dy)]
    dx, doff, dmask = ops.dcn_dgrad(*dev, 1, (1, 1), 1, 1, bf16=bf16)
    saved = RF.DCN_FUSED_BWD
    try:
        RF.DCN_FUSED_BWD = False
        ins = [t.clone().requires_grad_() for t in dev[:3]]
        RF.dcn_v2_conv(*ins, dev[3], None, 1, 1, 1, 1).backward(dev[4])
    finally:
        RF.DCN_FUSED_BWD = saved
    for name, a, b in zip(("dx", "doffset", "dmask"), (dx, doff, dmask), [t.grad for t in ins]):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-4 * scale, (name, (a - b).abs().max().item(), scale)
    assert dx[0, :, 7, 9].min().item() > 0 and (dx[0, :, :5] == 0).all()      # everything landed on rows 7..8, cols 9..10


@pytest.mark.parametrize("bf16", [False, True])
def test_dcn_dgrad_nonfinite_gradients_stay_visible(bf16):
    """A NaN / inf in dY (diverged training) must not be laundered into finite numbers by the fixed-point window: the
    input gradient around the affected block comes out NaN, d offset / d mask carry it too, and blocks that do not
    see the bad value stay finite."""
    from rrnet_amd import ops
    n, c, h, w, k = 1, 64, 40, 48, 64
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, 18, h, w, generator=g) * 0.5
    mask = torch.sigmoid(torch.randn(n, 9, h, w, generator=g))
    wt = torch.randn(k, c, 3, 3, generator=g) / 24.0
    for bad in (float("nan"), float("inf")):
        dy = torch.randn(n, k, h, w, generator=g)
        dy[0, 5, 3, 4] = bad
        dev = [ops.to_nhwc(t.cuda()) for t in (x, off, mask, wt, dy)]
        dx, doff, dmask = ops.dcn_dgrad(*dev, 1, (1, 1), 1, 1, bf16=bf16)
        assert not torch.isfinite(dx[0, :, 3, 4]).all()
        assert not torch.isfinite(dmask[0, :, 3, 4]).all()
        assert torch.isfinite(dx[0, :, 24:, 32:]).all() and torch.isfinite(dmask[0, :, 24:, 32:]).all()


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
    # tight: the kernel IS the GEMM of the bf16-rounded columns (blended and masked in fp32, as rr_dcn_im2col writes them)
    # with the bf16 weights, accumulated in fp32 — against a float64 matmul of those rounded operands
    from rrnet_amd import ops
    col = ops.dcn_im2col(dev[0], dev[1], dev[2], ks, ks, stride, (pad, pad), dil, dg)
    colr = col.permute(0, 2, 3, 1).reshape(-1, ks * ks * c).bfloat16().double()
    wr = dev[3].permute(0, 2, 3, 1).reshape(k, -1).double()                     # OHWI rows, already bf16 values
    tight = (colr @ wr.t() + b.cuda().double()).reshape(n, p, q, k).permute(0, 3, 1, 2)
    assert (out.double() - tight).abs().max().item() <= 3e-5 * scale, ((out.double() - tight).abs().max().item(), scale)
    # the backward of a bf16 forward runs the bf16-operand gradient kernels where the window kernels take the layer
    xg = dev[0].clone().requires_grad_()
    dcn_v2_conv(xg, dev[1], dev[2], dev[3], b.cuda(), stride, pad, dil, dg, bf16=True).sum().backward()
    assert torch.isfinite(xg.grad).all()


def _dcn_cfg():
    from types import SimpleNamespace
    return SimpleNamespace(num_classes=10, Model=SimpleNamespace(num_stacks=2, backbone="hourglass_tiny",
                           nms_type_for_stage1="nms", nms_per_class_for_stage1=True, dcn_heads=True, dcn_bf16=False),
                           Train=SimpleNamespace(scale_factor=4))


def test_rrnet_with_dcn_heads_vs_oracle_composition():
    """BASELINE configs[3] at model level (builder-defined option cfg.Model.dcn_heads): RRNet on the tiny backbone with
    the three heads' 3x3 convolutions replaced by DCN layers whose offset convolutions are NOT zero (real deformation),
    train mode: stage-1 maps of both stacks, the three stage-1 losses and the gradients of the DCN parameters against
    the oracle's composition (oracle/model.py:head_conv3x3 -> oracle/dcn.py)."""
    from oracle import model as om, ops as oo
    from rrnet_amd import functional as RF
    from helpers import host_synth_batch as synth_batch
    from rrnet_amd.models.rrnet import RRNet
    torch.manual_seed(11)
    model = RRNet(_dcn_cfg())
    g = torch.Generator().manual_seed(12)
    for name, p in model.named_parameters():
        if "conv_offset_mask.weight" in name:
            p.data.normal_(0, 0.02, generator=g)
        if "conv_offset_mask.bias" in name:
            p.data.normal_(0, 0.3, generator=g)
    for i in range(2):
        model.wh.detect_H_layer[i][0].conv.bias.data.fill_(3.0)
        model.wh.detect_W_layer[i][0].conv.bias.data.fill_(3.0)
    keys = [k for k, _ in model.named_parameters()
            if ".0.conv." in k and ("detect_layer." in k or "detect_conv_layer." in k)]      # 6 DCN layers x 4 tensors
    assert any("conv_offset_mask" in k for k in keys) and len(keys) == 24
    # The gradient check must not depend on discrete decisions that fp32 rounding can flip (a hidden ReLU whose
    # pre-activation is within rounding of zero, the sign() of the L1 losses): the DCN biases are raised so that every
    # hidden unit is active, and the gradients are taken of a dense smooth functional of the head outputs.
    for name, p in model.named_parameters():
        if name in keys and name.endswith(".0.conv.bias"):
            p.data.fill_(5.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k in keys:
        sd[k].requires_grad_()
    imgs, annos, hms, whs, inds, offs, masks, _ = synth_batch(2, 128, 128, boxes_per_image=10, seed=3)
    proj = [torch.randn(2, c, 32, 32, generator=g) for c in (10, 2, 2)]
    P = om.Params(sd, True)
    r_hm, r_wh, r_off = om.stage1(P, om.hourglass_net(P, imgs))
    r_losses = [sum(oo.hm_loss_from_logits(r_hm[i], hms) / 2 for i in range(2)),
                sum(oo.reg_l1_loss(r_wh[i], masks, inds, whs) / 2 for i in range(2)),
                sum(oo.reg_l1_loss(r_off[i], masks, inds, offs) / 2 for i in range(2))]
    sum((o * w).sum() for i in range(2) for o, w in zip((r_hm[i], r_wh[i], r_off[i]), proj)).backward()
    model = model.cuda().to(memory_format=CL).train()
    outs = model(imgs.cuda(), k=50)
    for i in range(2):
        for a, b in ((outs[0][i], r_hm[i]), (outs[1][i], r_wh[i]), (outs[2][i], r_off[i])):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), atol=1e-3, rtol=1e-3)
    gt = [t.cuda() for t in (hms, whs, inds, offs, masks)]
    losses = [sum(RF.focal_loss_hm_from_logits(outs[0][i], gt[0]) / 2 for i in range(2)),
              sum(RF.reg_l1_loss(outs[1][i], gt[4], gt[2], gt[1]) / 2 for i in range(2)),
              sum(RF.reg_l1_loss(outs[2][i], gt[4], gt[2], gt[3]) / 2 for i in range(2))]
    np.testing.assert_allclose([float(l.detach()) for l in losses], [float(l.detach()) for l in r_losses], rtol=1e-3, atol=1e-3)
    sum((o * w.cuda()).sum() for i in range(2) for o, w in zip((outs[0][i], outs[1][i], outs[2][i]), proj)).backward()
    named = dict(model.named_parameters())
    for k in keys:
        ref = sd[k].grad.numpy()
        got = named[k].grad.detach().cpu().numpy()
        scale = max(np.abs(ref).max(), 1e-8)
        assert np.abs(got - ref).max() <= 2e-3 * scale + 1e-7, (k, np.abs(got - ref).max(), scale)


def test_rrnet_with_dcn_heads_train_step_bf16():
    """Train-step property check of the config-4 model (DCN heads, bf16 matrix operands in the deformable forward):
    finite losses, every parameter — the DCN weights and their zero-initialised offset convolutions included —
    receives a finite gradient and moves."""
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    saved = dict(cfg.Model)
    try:
        cfg.Train.batch_size, cfg.Train.crop_size = 2, (128, 128)
        cfg.Model.backbone, cfg.Model.dcn_heads, cfg.Model.dcn_bf16 = "hourglass_tiny", True, True
        cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
        torch.manual_seed(cfg.seed)
        op = RRNetOperator(cfg)
        op.model.train()
        flat = op.model.flat
        before = flat.flat.clone()
        b = op.training_loader.get_batch()
        _, losses = op.train_step(2000, (b[0], b[1].clone()) + tuple(b[2:]))
        assert all(np.isfinite(float(v.detach())) for v in losses)
        assert torch.isfinite(flat.grad).all() and torch.isfinite(flat.flat).all()
        names = dict(op.model.module.named_parameters())
        dcn = op.model.module.hm.detect_layer[1][0].conv
        assert dcn.bf16 is True
        for p in (dcn.weight, dcn.conv_offset_mask.weight, dcn.conv_offset_mask.bias):
            assert p.grad.abs().sum().item() > 0
        assert (flat.flat != before).float().mean().item() > 0.97
    finally:
        for k in list(cfg.Model):
            if k not in saved:
                del cfg.Model[k]
        cfg.Model.update(saved)


def test_dcn_bench_layer_full_size_vs_column_path_and_cropped_oracle():
    """The layer bench.py's `config4` times — 256->256 3x3 DCNv2 on B=8 x 256x256 — at its own launch configuration:
      (i)  fused forward / backward (fp32 operands) against the reference-structured column path of the same library on
           the full tensors (all four gradients);
      (ii) against the oracle (oracle/dcn.py) on three cropped windows (top-left corner, an interior window straddling
           8x16 pixel blocks, bottom-right corner) of three images: the op is local for bounded offsets, so a crop with a
           margin of 2 x (pad + max|offset| + 1) reproduces out, d offset, d mask and d input of the window's core exactly;
      (iii) the bf16-operand kernels (BASELINE config 4) against the fp32 result within bf16's rounding."""
    from oracle import dcn as odcn
    from rrnet_amd import functional as RF
    n, c, h, w, k = 8, 256, 256, 256, 256
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(n, c, h, w, device="cuda", generator=g).contiguous(memory_format=CL)
    off = torch.randn(n, 18, h, w, device="cuda", generator=g).clamp(-3, 3).contiguous(memory_format=CL)
    mask = torch.sigmoid(torch.randn(n, 9, h, w, device="cuda", generator=g)).contiguous(memory_format=CL)
    wt = (torch.randn(k, c, 3, 3, device="cuda", generator=g) / 48.0).contiguous(memory_format=CL)
    bias = torch.randn(k, device="cuda", generator=g)
    gy = torch.randn(n, k, h, w, device="cuda", generator=g).contiguous(memory_format=CL)
    res = {}
    saved = RF.DCN_FUSED_BWD
    try:
        for tag, fused, bf in (("fused", True, False), ("column", False, False), ("bf16", True, True)):
            RF.DCN_FUSED_BWD = fused
            ins = [t.clone().requires_grad_() for t in (x, off, mask, wt, bias)]
            out = RF.dcn_v2_conv(*ins, 1, 1, 1, 1, bf16=bf)
            out.backward(gy)
            torch.cuda.synchronize()
            res[tag] = [out.detach()] + [t.grad for t in ins]
            del out, ins
    finally:
        RF.DCN_FUSED_BWD = saved
    names = ("out", "dx", "doffset", "dmask", "dw", "db")
    for name, a, b in zip(names, res["fused"], res["column"]):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-4 * scale, ("fused vs column", name, (a - b).abs().max().item(), scale)
    for name, a, b in zip(names, res["bf16"], res["fused"]):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 3e-2 * scale, ("bf16 vs fp32", name, (a - b).abs().max().item(), scale)
    m = 2 * (1 + 3 + 1)
    for n0, (r0, r1, c0, c1) in ((0, (0, 20, 0, 28)), (3, (117, 141, 104, 140)), (7, (236, 256, 228, 256))):
        ra, rb, ca, cb = max(0, r0 - m), min(h, r1 + m), max(0, c0 - m), min(w, c1 + m)
        crop = [t[n0:n0 + 1, :, ra:rb, ca:cb].detach().cpu().contiguous().requires_grad_() for t in (x, off, mask)]
        ref = odcn.dcn_v2_conv(crop[0], crop[1], crop[2], wt.cpu().contiguous(), bias.cpu(), 1, 1, 1, 1)
        ref.backward(gy[n0:n0 + 1, :, ra:rb, ca:cb].cpu().contiguous())
        core = (slice(None), slice(None), slice(r0 - ra, r1 - ra), slice(c0 - ca, c1 - ca))
        full = (slice(n0, n0 + 1), slice(None), slice(r0, r1), slice(c0, c1))
        for name, got, exp in (("out", res["fused"][0], ref.detach()), ("dx", res["fused"][1], crop[0].grad),
                               ("doffset", res["fused"][2], crop[1].grad), ("dmask", res["fused"][3], crop[2].grad)):
            e = exp[core]
            tol = 1e-3 * max(1.0, e.abs().max().item())
            np.testing.assert_allclose(got[full].cpu().numpy(), e.numpy(), atol=tol, rtol=1e-3,
                                       err_msg="%s vs oracle, image %d window %s" % (name, n0, (r0, r1, c0, c1)))


def test_dcn_dgrad_fixed_point_scale_with_vanishing_gradients():
    """column gradients around 1e-35 (bound below 2^-96) must not overflow the power-of-two scale of the
    fixed-point window: d input equals the fp32 column path's result scaled down (the op is linear in dy)."""
    from rrnet_amd import ops
    g = torch.Generator().manual_seed(77)
    n, c, h, w, k = 1, 64, 16, 24, 64
    x = ops.to_nhwc(torch.randn(n, c, h, w, generator=g).cuda())
    off = ops.to_nhwc((torch.randn(n, 18, h, w, generator=g) * 0.7).cuda())
    mask = ops.to_nhwc(torch.sigmoid(torch.randn(n, 9, h, w, generator=g)).cuda())
    wt = ops.to_nhwc((torch.randn(k, c, 3, 3, generator=g) / 24.0).cuda())
    dy = ops.to_nhwc(torch.randn(n, k, h, w, generator=g).cuda())
    ref = ops.dcn_dgrad(x, off, mask, wt, dy, 1, (1, 1), 1, 1)[0]
    tiny = ops.dcn_dgrad(x, off, mask, wt, dy * 1e-34, 1, (1, 1), 1, 1)[0]
    assert torch.isfinite(tiny).all()
    scale = ref.abs().max().item()
    assert ((tiny * 1e34) - ref).abs().max().item() <= 1e-3 * scale      # fp32 denormal range starts near 1e-38
