"""GPU parity: NHWC fp32 MFMA convolutions (fprop / dgrad / wgrad, through the C ABI) against
torch-CPU conv2d and its autograd — torch is the de-facto spec of these layers.
Tolerance: fp32 accumulation in a different order, |err| <= 1e-4 + 1e-4*|ref| scaled by sqrt(Kg)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, C, H, W, K, R, S, stride, pad_h, pad_w, bias, relu)
SHAPES = [
    (2, 32, 16, 16, 32, 3, 3, 1, 1, 1, False, False),
    (2, 256, 16, 16, 256, 3, 3, 1, 1, 1, False, False),     # the dominant shape, small spatial
    (1, 64, 9, 13, 128, 3, 3, 1, 1, 1, False, False),       # ragged M / N tiles
    (2, 128, 16, 16, 256, 3, 3, 2, 1, 1, False, False),     # stride-2 3x3
    (2, 128, 15, 17, 256, 1, 1, 2, 0, 0, False, False),     # stride-2 1x1 skip, odd size
    (2, 256, 8, 8, 384, 1, 1, 1, 0, 0, False, False),
    (2, 3, 32, 32, 16, 7, 7, 2, 3, 3, False, False),        # stem (scalar gather path)
    (2, 3, 30, 34, 128, 7, 7, 2, 3, 3, False, False),
    (1, 256, 12, 12, 10, 1, 1, 1, 0, 0, True, False),       # hm head 1x1 (BN=32 tile, N masked)
    (1, 256, 12, 12, 2, 1, 1, 1, 0, 0, True, False),
    (1, 256, 20, 12, 1, 17, 1, 1, 8, 0, True, False),       # HCov
    (1, 256, 12, 20, 1, 1, 17, 1, 0, 8, True, False),       # WCov
    (1, 256, 10, 10, 256, 3, 3, 1, 1, 1, True, True),       # head 3x3 + bias + ReLU
    (7, 256, 3, 3, 64, 1, 1, 1, 0, 0, False, False),        # stage-2 bottleneck on RoIs
    (7, 64, 3, 3, 64, 3, 3, 1, 1, 1, False, False),
    (7, 256, 1, 1, 4, 1, 1, 1, 0, 0, True, False),          # regressor
    (2, 48, 8, 8, 24, 3, 3, 1, 1, 1, False, False),         # channels not a multiple of 32
    (2, 12, 6, 6, 8, 3, 3, 2, 1, 1, False, False),
    (1, 512, 4, 4, 512, 3, 3, 1, 1, 1, False, False),
    (2, 16, 7, 9, 24, 3, 3, 2, 1, 1, False, False),         # stride-2 dgrad parity classes, odd H and W
    (1, 64, 11, 11, 128, 3, 3, 2, 1, 1, False, False),
    (3, 40, 5, 7, 40, 1, 1, 2, 0, 0, False, False),
    # ---- the kernel variants only LARGE layers select (software-pipelined 128x128 tiles, wgrad's uniform row walk
    # ---- for Q % 32 == 0, stride-1 dgrad through the forward kernel at >= 4096 pixels, mid/small tile choices)
    (2, 256, 64, 64, 256, 3, 3, 1, 1, 1, False, False),     # dominant layer type at 64x64 (Q = 2 K-steps per row)
    (1, 256, 96, 96, 256, 3, 3, 1, 1, 1, True, True),       # Q = 96, head conv with bias + ReLU
    (2, 128, 128, 128, 256, 3, 3, 2, 1, 1, False, False),   # stem ResidualBlock conv1: stride 2, Q = 64
    (2, 128, 128, 128, 256, 1, 1, 2, 0, 0, False, False),   # its 1x1 stride-2 skip
    (2, 256, 64, 64, 256, 3, 3, 2, 1, 1, False, False),     # low1 stride-2 3x3, Q = 32
    (2, 256, 64, 64, 384, 3, 3, 2, 1, 1, False, False),
    (1, 384, 32, 32, 384, 3, 3, 1, 1, 1, False, False),     # 128x64 / 128x32 tile territory, split-K
    (1, 384, 16, 16, 512, 3, 3, 2, 1, 1, False, False),
    (2, 256, 64, 64, 36, 1, 1, 1, 0, 0, False, False),      # WH head's fused 1x1 (36 tap products)
    (2, 256, 64, 64, 10, 1, 1, 1, 0, 0, True, False),       # hm head 1x1 at 64x64
    (1, 256, 64, 160, 256, 3, 3, 1, 1, 1, False, False),    # non-square, Q = 160
    (1, 256, 66, 70, 256, 3, 3, 1, 1, 1, False, False),     # Q % 32 != 0 at a pipelined size
]


def _mk(shape, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


@pytest.mark.parametrize("cfg", SHAPES, ids=lambda c: "n%dc%dh%dw%dk%dr%ds%d_s%d" % c[:8])
def test_conv_fprop_dgrad_wgrad(cfg):
    from rrnet_amd import ops
    n, c, h, w, k, r, s, stride, ph, pw, use_bias, relu = cfg
    x = _mk((n, c, h, w), 1)
    wt = _mk((k, c, r, s), 2) * (1.0 / np.sqrt(c * r * s))
    b = _mk((k,), 3) if use_bias else None
    xr = x.clone().requires_grad_()
    wr = wt.clone().requires_grad_()
    y_ref = F.conv2d(xr, wr, b, stride=stride, padding=(ph, pw))
    if relu:
        y_ref = F.relu(y_ref)
    gy = _mk(tuple(y_ref.shape), 4)
    y_ref.backward(gy)

    xd = ops.to_nhwc(x.cuda())
    wd = ops.to_nhwc(wt.cuda())
    bd = b.cuda() if use_bias else None
    y, slab = ops.conv_fprop(xd, wd, bd, stride, (ph, pw), relu, want_stats=True)
    tol = 2e-5 * np.sqrt(c * r * s)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.detach().numpy(), atol=tol, rtol=1e-4)
    # fused BatchNorm partial sums: column sums / sums of squares of y
    p, q = y.shape[2], y.shape[3]
    st = slab.view(-1, 2, k).sum(0).cpu().numpy()
    yr = y_ref.detach().double().permute(0, 2, 3, 1).reshape(-1, k).numpy()
    np.testing.assert_allclose(st[0], yr.sum(0), atol=1e-3 * np.sqrt(n * p * q), rtol=1e-4)
    np.testing.assert_allclose(st[1], (yr * yr).sum(0), atol=1e-3 * np.sqrt(n * p * q), rtol=1e-4)

    gy_eff = gy.clone()
    if relu:
        gy_eff = gy_eff * (y_ref.detach() > 0).float()
    gyd = ops.to_nhwc(gy_eff.cuda())
    tol = 2e-5 * np.sqrt(k * r * s)
    base = _mk((n, c, h, w), 5)
    saved = ops._DGRAD_VIA_FPROP, ops._DGRAD_VIA_FPROP_MIN_PIXELS
    ops._DGRAD_VIA_FPROP_MIN_PIXELS = 0
    try:
        # both data-gradient routes: the dgrad kernel, and (stride 1) the forward kernel on flipped weights
        for via_fprop in (False, True):
            ops._DGRAD_VIA_FPROP = via_fprop
            dx = ops.conv_dgrad(gyd, wd, (n, c, h, w), stride, (ph, pw))
            np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), atol=tol, rtol=1e-4)
            # accumulate form
            acc = ops.to_nhwc(base.cuda()).clone(memory_format=torch.channels_last)
            ops.conv_dgrad(gyd, wd, (n, c, h, w), stride, (ph, pw), out=acc, accumulate=True)
            np.testing.assert_allclose(acc.cpu().numpy(), (xr.grad + base).numpy(), atol=tol, rtol=1e-4)
    finally:
        ops._DGRAD_VIA_FPROP, ops._DGRAD_VIA_FPROP_MIN_PIXELS = saved

    dw = torch.zeros((k, r, s, c), device="cuda").permute(0, 3, 1, 2)
    ops.conv_wgrad(xd, gyd, dw, stride, (ph, pw))
    tol = 3e-5 * np.sqrt(n * p * q)
    np.testing.assert_allclose(dw.cpu().numpy(), wr.grad.numpy(), atol=tol, rtol=2e-4)


# (N, C, H, W, K, R, S, pad_h, pad_w, bias, relu): tiny maps, many images — the stage-2 head's geometry
POS_MAJOR = [
    (2500, 64, 3, 3, 64, 3, 3, 1, 1, True, True),       # Bottleneck.conv2 on RoI maps (N not a multiple of 128)
    (2048, 32, 2, 4, 48, 3, 5, 1, 2, False, False),     # non-square map and filter, K = 48 (masked N tile)
    (2100, 64, 4, 4, 128, 3, 3, 1, 1, True, False),     # 16 pixels, two N tiles at BN = 64
    (2304, 16, 1, 3, 32, 3, 3, 1, 1, False, True),      # one row: only the middle filter row is real
    (2050, 64, 3, 3, 64, 5, 5, 2, 2, False, False),     # filter wider than the map
]


@pytest.mark.parametrize("cfg", POS_MAJOR, ids=lambda c: "n%dc%dh%dw%dk%dr%ds%d" % c[:7])
def test_conv_position_major_tiles_skip_padding_taps(cfg):
    """Padded filters on tiny maps with >= 2048 images take the position-major tiling (ConvArgs::pos_major: one output
    pixel per M tile, padding taps skipped): forward (no statistics requested) and the stride-1 data gradient through the
    forward kernel, plain and accumulating, against torch-CPU conv2d / autograd."""
    from rrnet_amd import ops
    n, c, h, w, k, r, s, ph, pw, use_bias, relu = cfg
    x = _mk((n, c, h, w), 11)
    wt = _mk((k, c, r, s), 12) * (1.0 / np.sqrt(c * r * s))
    b = _mk((k,), 13) if use_bias else None
    xr = x.clone().requires_grad_()
    y_ref = F.conv2d(xr, wt, b, stride=1, padding=(ph, pw))
    if relu:
        y_ref = F.relu(y_ref)
    gy = _mk(tuple(y_ref.shape), 14)
    y_ref.backward(gy * (y_ref.detach() > 0).float() if relu else gy)
    xd, wd = ops.to_nhwc(x.cuda()), ops.to_nhwc(wt.cuda())
    y = ops.conv_fprop(xd, wd, b.cuda() if use_bias else None, 1, (ph, pw), relu)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.detach().numpy(), atol=2e-5 * np.sqrt(c * r * s), rtol=1e-4)
    # with statistics the ordinary tiling runs: same values up to summation order
    y2, _ = ops.conv_fprop(xd, wd, None, 1, (ph, pw), False, want_stats=True)
    y3 = ops.conv_fprop(xd, wd, None, 1, (ph, pw), False)
    np.testing.assert_allclose(y3.cpu().numpy(), y2.cpu().numpy(), atol=2e-5 * np.sqrt(c * r * s), rtol=1e-4)
    if 2 * ph == r - 1 and 2 * pw == s - 1:
        gyd = ops.to_nhwc((gy * (y_ref.detach() > 0).float() if relu else gy).cuda())
        saved = ops._DGRAD_VIA_FPROP, ops._DGRAD_VIA_FPROP_MIN_PIXELS
        ops._DGRAD_VIA_FPROP, ops._DGRAD_VIA_FPROP_MIN_PIXELS = True, 0
        try:
            tol = 2e-5 * np.sqrt(k * r * s)
            dx = ops.conv_dgrad(gyd, wd, (n, c, h, w), 1, (ph, pw))
            np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), atol=tol, rtol=1e-4)
            base = _mk((n, c, h, w), 15)
            acc = ops.to_nhwc(base.cuda()).clone(memory_format=torch.channels_last)
            ops.conv_dgrad(gyd, wd, (n, c, h, w), 1, (ph, pw), out=acc, accumulate=True)
            np.testing.assert_allclose(acc.cpu().numpy(), (xr.grad + base).numpy(), atol=tol, rtol=1e-4)
        finally:
            ops._DGRAD_VIA_FPROP, ops._DGRAD_VIA_FPROP_MIN_PIXELS = saved


def test_conv_linearity_large():
    """Size-independent property at a BASELINE-sized layer (256->256 3x3 on 256x256, B=1):
    conv(a*x1 + x2) == a*conv(x1) + conv(x2) within fp32 rounding, and a spot check of one
    output row against torch-CPU."""
    from rrnet_amd import ops
    torch.manual_seed(0)
    x1 = ops.to_nhwc(torch.randn(1, 256, 256, 256, device="cuda"))
    x2 = ops.to_nhwc(torch.randn(1, 256, 256, 256, device="cuda"))
    w = ops.to_nhwc(torch.randn(256, 256, 3, 3, device="cuda") / 48.0)
    y1 = ops.conv_fprop(x1, w, None, 1, (1, 1))
    y2 = ops.conv_fprop(x2, w, None, 1, (1, 1))
    y12 = ops.conv_fprop(ops.to_nhwc(0.5 * x1 + x2), w, None, 1, (1, 1))
    err = (y12 - (0.5 * y1 + y2)).abs().max().item()
    assert err < 5e-4, err
    rows = slice(100, 104)
    ref = F.conv2d(x1[:, :, 99:105, :].cpu(), w.cpu(), None, 1, (0, 1))
    np.testing.assert_allclose(y1[:, :, rows, :].cpu().numpy(), ref.numpy(), atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("npix,c,relu", [(70001, 256, True), (4097, 64, True), (513, 4, False), (1000, 10, True),
                                         (333, 12, True), (65536, 256, False), (7, 256, True)])
def test_bias_relu_bwd_column_sums(npix, c, relu):
    """rr_bias_relu_bwd (head convs: conv + bias [+ ReLU]): dy * (z > 0) bit-exact, column sums added to dbias within
    fp32 summation noise of a float64 sum.  Wide heads (C % 4 == 0, C/4 | 256) take the 16-byte kernel, the rest the
    scalar one; ragged pixel counts exercise both tails."""
    from rrnet_amd import ops
    g = torch.Generator().manual_seed(npix + c)
    dy = torch.randn(npix, c, generator=g).cuda()
    z = torch.randn(npix, c, generator=g).cuda() if relu else None
    db0 = torch.randn(c, generator=g).cuda()
    db = db0.clone()
    out = ops.bias_relu_bwd(dy, z, db)
    want = dy * (z > 0) if relu else dy
    assert torch.equal(out, want)
    ref = want.double().sum(0) + db0.double()
    tol = 1e-6 * want.abs().double().sum(0).max().item() + 1e-6
    assert (db.double() - ref).abs().max().item() <= tol


# (N, C = producer's channels, H, W, K = dy channels, R, pad, mask source, accumulate)
BNSUM_CASES = [
    (2, 256, 96, 96, 256, 3, 1, "scale", False),      # conv1 -> bn1 -> relu -> conv2: the common case, 128x128 tiles
    (2, 256, 96, 96, 256, 3, 1, "z", True),           # block output (residual): mask from z, last contributor of a fan-in
    (2, 256, 96, 96, 384, 1, 0, "z", True),           # projection skip 1x1 as the last contributor
    (2, 64, 128, 128, 128, 3, 1, "scale", False),     # 64 producer channels: 128x64 tiles
    (2, 32, 128, 128, 64, 3, 1, "none", False),       # 32 channels: 128x32 tiles; producer without ReLU
    (1, 384, 64, 64, 384, 3, 1, "scale", False),      # under-filled -> split-K: the entry falls back to a reduce pass
    (3, 256, 81, 83, 256, 3, 1, "z", False),          # ragged: M not a multiple of 128
    (2, 256, 64, 64, 256, 3, 1, "z", True),           # split-K with accumulation
]


@pytest.mark.parametrize("cfg", BNSUM_CASES, ids=lambda c: "n%dc%dh%dw%dk%dr%dp%d_%s_acc%d" % c)
def test_dgrad_with_producer_bn_backward_sums(cfg):
    """rr_conv_dgrad_s1_bnsum: the stride-1 data gradient whose epilogue also reduces the producer's BatchNorm-backward
    sums == rr_conv_dgrad_s1 followed by rr_bn_bwd_reduce on its output (and both against an fp64 host evaluation)."""
    from rrnet_amd import ops
    n, c, h, w, k, r, pad, mask, acc = cfg
    g = torch.Generator().manual_seed(sum(cfg[:7]))
    dy = ops.to_nhwc(torch.randn(n, k, h, w, generator=g).cuda())
    wt = ops.to_nhwc((torch.randn(k, c, r, r, generator=g) / np.sqrt(k * r * r)).cuda())
    y = ops.to_nhwc(torch.randn(n, c, h, w, generator=g).cuda())
    res = ops.to_nhwc(torch.randn(n, c, h, w, generator=g).cuda())
    mean, var = torch.randn(c, generator=g).cuda() * 0.1, (torch.rand(c, generator=g).cuda() + 0.5)
    invstd = 1.0 / torch.sqrt(var)
    scale, shift = (torch.rand(c, generator=g).cuda() + 0.5) * invstd, torch.randn(c, generator=g).cuda() * 0.3
    link = ops.BnLink()
    link.y, link.mean, link.invstd = y, mean, invstd
    z = None
    if mask == "z":
        link.use_z = True
        z = torch.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res).contiguous(memory_format=torch.channels_last)
    elif mask == "scale":
        link.msc, link.msh = scale, shift
    base = ops.to_nhwc(torch.randn(n, c, h, w, generator=g).cuda()) if acc else None
    out_a = base.clone() if acc else None
    out_b = base.clone() if acc else None
    dx_ref = ops.conv_dgrad(dy, wt, (n, c, h, w), 1, (pad, pad), out=out_a, accumulate=acc)
    sums_ref = ops.bn_bwd_reduce(dx_ref, z, y, mean, invstd, mask_scale=link.msc, mask_shift=link.msh)
    dx = ops.conv_dgrad(dy, wt, (n, c, h, w), 1, (pad, pad), out=out_b, accumulate=acc, bnsum=link, bnsum_z=z)
    assert link.sums is not None and link.dz is dx
    mtiles, ntiles = -(-n * h * w // 128), -(-c // 128)
    if mtiles * ntiles >= 256:
        assert torch.equal(dx, dx_ref)                                 # same kernel arithmetic, same stores
    else:                                                              # split-K: float atomics, order varies run to run
        assert (dx - dx_ref).abs().max().item() <= 1e-5 * dx_ref.abs().max().item()
    d = dx.double()
    if mask == "z":
        d = d * (z > 0)
    elif mask == "scale":
        d = d * (torch.addcmul(shift.view(1, -1, 1, 1), y, scale.view(1, -1, 1, 1)) > 0)
    xh = (y.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
    exp = torch.cat([d.sum((0, 2, 3)), (d * xh).sum((0, 2, 3))])
    mag = torch.cat([d.abs().sum((0, 2, 3)), (d * xh).abs().sum((0, 2, 3))]).max().item()
    assert (link.sums[:2 * c] - exp).abs().max().item() <= 2e-6 * mag, (link.sums[:2 * c] - exp).abs().max().item() / mag
    assert (sums_ref[:2 * c] - exp).abs().max().item() <= 2e-6 * mag


def test_train_step_gradients_equal_with_and_without_fused_bn_sums():
    """A tiny hourglass train step with the fused producer sums (default) and with RR_DGRAD_BNSUM / RR_BN_G_INTO off:
    same losses, same flat gradient up to summation order; the fused path really ran (bn_bwd_reduce launches drop)."""
    from types import SimpleNamespace
    from rrnet_amd import functional as RF, ops
    from rrnet_amd.datasets.synthetic import synth_batch
    from rrnet_amd.flat import FlatParams
    from rrnet_amd.models.rrnet import RRNet
    cfg = SimpleNamespace(num_classes=10, Model=SimpleNamespace(num_stacks=2, backbone="hourglass_tiny",
                          nms_type_for_stage1="nms", nms_per_class_for_stage1=True))
    batch = synth_batch(2, 512, 512, boxes_per_image=12)
    grads, counts = {}, {}
    saved = (ops._DGRAD_BNSUM, RF._G_INTO, ops.bn_bwd_reduce)
    try:
        for fused in (True, False):
            ops._DGRAD_BNSUM, RF._G_INTO = fused, fused
            calls = [0]

            def counting(*a, _f=saved[2], **kw):
                calls[0] += 1
                return _f(*a, **kw)
            ops.bn_bwd_reduce = counting
            torch.manual_seed(219)
            model = RRNet(cfg).cuda().to(memory_format=torch.channels_last).train()
            flat = FlatParams(model)
            imgs, annos, hms, whs, inds, offs, masks, _ = batch
            outs = model(imgs, k=100)
            loss = sum(RF.focal_loss_hm_from_logits(outs[0][i], hms) + RF.reg_l1_loss(outs[1][i], masks, inds, whs)
                       + RF.reg_l1_loss(outs[2][i], masks, inds, offs) for i in range(2))
            loss.backward()
            torch.cuda.synchronize()
            grads[fused], counts[fused] = flat.grad.clone(), calls[0]
    finally:
        ops._DGRAD_BNSUM, RF._G_INTO, ops.bn_bwd_reduce = saved
    assert counts[True] < counts[False], counts          # some reduce passes disappeared
    scale = grads[False].abs().max().item()
    assert (grads[True] - grads[False]).abs().max().item() <= 1e-4 * scale, ((grads[True] - grads[False]).abs().max().item(), scale)


@pytest.mark.parametrize("cfg", [(2, 3, 30, 34, 128, 7, 2, 3), (1, 3, 64, 64, 128, 7, 2, 3), (2, 3, 17, 19, 64, 5, 1, 2),
                                 (1, 6, 21, 20, 96, 3, 2, 1), (2, 3, 128, 128, 128, 7, 2, 3)],
                         ids=lambda c: "n%dc%dh%dw%dk%dr%ds%dp%d" % c)
def test_packed_small_channel_conv_vs_torch(cfg):
    """The few-channel path (7x7 stride-2 stem, backbones/hourglass.py:143): rr_conv_pack_taps + 1x1 GEMMs on the vector
    kernels == conv2d / its weight gradient on the original image (fp64 reference)."""
    from rrnet_amd import ops
    n, c, h, w, k, r, stride, pad = cfg
    x = ops.to_nhwc(_mk((n, c, h, w), 1).cuda())
    wt = ops.to_nhwc((_mk((k, c, r, r), 2) / np.sqrt(c * r * r)).cuda())
    assert ops.conv_packable(x, wt, stride)
    y, slab, xp = ops.conv_fprop_packed(x, wt, stride, (pad, pad), want_stats=True)
    ref = F.conv2d(x.cpu().double(), wt.cpu().double(), None, stride, pad)
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    unf = F.unfold(x.cpu(), r, padding=pad, stride=stride)                       # [n, c*r*r, P*Q], rows ordered (c, r, s)
    kg = c * r * r
    exp = unf.view(n, c, r * r, -1).permute(0, 3, 2, 1).reshape(n, ref.shape[2], ref.shape[3], kg)   # -> (r, s, c) minor
    got = xp.permute(0, 2, 3, 1).cpu()
    assert torch.equal(got[..., :kg], exp) and float(got[..., kg:].abs().max() if got.shape[-1] > kg else 0.0) == 0.0
    sums = ops.bn_reduce_slab(slab, k).cpu()
    assert (sums[:k] - ref.sum((0, 2, 3))).abs().max().item() <= 1e-4 * ref.abs().sum((0, 2, 3)).max().item()
    dy = ops.to_nhwc(_mk(tuple(ref.shape), 3).cuda())
    base = ops.to_nhwc(_mk((k, c, r, r), 4).cuda())
    dw = base.clone()
    ops.conv_wgrad_packed(xp, dy, dw)
    dref = torch.nn.grad.conv2d_weight(x.cpu().double(), (k, c, r, r), dy.cpu().double(), stride, pad)
    assert ((dw - base).cpu().double() - dref).abs().max().item() <= 2e-5 * dref.abs().max().item()


@pytest.mark.parametrize("cfg", [(2, 256, 96, 96, 10, 1, 0), (2, 256, 96, 96, 36, 1, 0), (1, 256, 128, 160, 2, 1, 0),
                                 (2, 64, 128, 128, 12, 3, 1)], ids=lambda c: "n%dc%dh%dw%dk%dr%dp%d" % c)
def test_dgrad_with_producer_relu_mask_and_bias_sums(cfg):
    """rr_conv_dgrad_s1_relubias (producer = conv + bias + ReLU, the heads' 3x3 layers): the data gradient stores
    dx * (z > 0) and returns its column sums == rr_conv_dgrad followed by rr_bias_relu_bwd; K = 10 / 2 go through the
    zero-padding to a multiple of 4."""
    from rrnet_amd import ops
    n, c, h, w, k, r, pad = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    dy = ops.to_nhwc(torch.randn(n, k, h, w, generator=g).cuda())
    wt = ops.to_nhwc((torch.randn(k, c, r, r, generator=g) / np.sqrt(k * r * r)).cuda())
    z = ops.to_nhwc(torch.relu(torch.randn(n, c, h, w, generator=g)).cuda())
    link = ops.BnLink()
    link.relu_bias = link.use_z = True
    dx = ops.conv_dgrad(dy, wt, (n, c, h, w), 1, (pad, pad), bnsum=link, bnsum_z=z)
    assert link.sums is not None and link.dz is dx
    ref = ops.conv_dgrad(dy, wt, (n, c, h, w), 1, (pad, pad))
    db = torch.zeros(c, device="cuda")
    masked = ops.bias_relu_bwd(ref, z, db)
    scale = ref.abs().max().item()
    assert (dx - masked).abs().max().item() <= 2e-6 * scale                 # K padded: another K-step split, same values
    assert float((dx * (z <= 0)).abs().max()) == 0.0
    exp = masked.double().sum((0, 2, 3))
    mag = masked.double().abs().sum((0, 2, 3)).max().item()
    assert (link.sums[:c] - exp).abs().max().item() <= 2e-6 * mag
    assert (db.double() - exp).abs().max().item() <= 1e-4 * mag


def test_dgrad_relu_mask_on_the_sum_of_a_fan_in():
    """accumulate form of rr_conv_dgrad_s1_relubias: the last contributor adds its gradient to the others' and stores the
    SUM masked with the producer's output (relu(feature) fanned out to the three heads' 3x3 layers)."""
    from rrnet_amd import ops
    n, c, h, w, k = 2, 256, 96, 96, 256
    g = torch.Generator().manual_seed(5)
    dy = ops.to_nhwc(torch.randn(n, k, h, w, generator=g).cuda())
    wt = ops.to_nhwc((torch.randn(k, c, 3, 3, generator=g) / 48.0).cuda())
    z = ops.to_nhwc(torch.relu(torch.randn(n, c, h, w, generator=g)).cuda())
    others = ops.to_nhwc(torch.randn(n, c, h, w, generator=g).cuda())
    link = ops.BnLink()
    link.relu_bias = link.use_z = True
    buf = others.clone()
    ops.conv_dgrad(dy, wt, (n, c, h, w), 1, (1, 1), out=buf, accumulate=True, bnsum=link, bnsum_z=z)
    assert link.sums is not None and link.dz is buf
    ref = (ops.conv_dgrad(dy, wt, (n, c, h, w), 1, (1, 1)) + others) * (z > 0)
    assert (buf - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("k", [10, 2, 34])
def test_head_1x1_dgrad_kernel_accumulate_and_legacy_path(k):
    """rr_head_dgrad_relubias (round 5: the heads' narrow 1x1 data gradients as ONE element-wise pass instead of a 12 / 4 / 36-deep
    implicit GEMM): the accumulate form stores (others + dy w) masked, and the result equals the round-3 path
    (rr_conv_dgrad_s1_relubias on zero-padded channels, ops._HEAD_DGRAD = False) — values, column sums, exact zeros under the mask."""
    from rrnet_amd import ops
    n, c, h, w = 2, 256, 96, 128
    g = torch.Generator().manual_seed(40 + k)
    dy = ops.to_nhwc(torch.randn(n, k, h, w, generator=g).cuda())
    wt = ops.to_nhwc((torch.randn(k, c, 1, 1, generator=g) / np.sqrt(k)).cuda())
    z = ops.to_nhwc(torch.relu(torch.randn(n, c, h, w, generator=g)).cuda())
    others = ops.to_nhwc(torch.randn(n, c, h, w, generator=g).cuda())
    outs = {}
    saved = (ops._HEAD_DGRAD, ops._HEAD_DGRAD_MAX_K)
    ops._HEAD_DGRAD_MAX_K = 36            # (the product keeps K = 34 on the implicit-GEMM kernel: slower there; checked here all the same)
    try:
        for mode in (True, False):
            ops._HEAD_DGRAD = mode
            link = ops.BnLink()
            link.relu_bias = link.use_z = True
            buf = others.clone()
            ops.conv_dgrad(dy, wt, (n, c, h, w), 1, (0, 0), out=buf, accumulate=True, bnsum=link, bnsum_z=z)
            assert link.sums is not None and link.dz is buf
            outs[mode] = (buf, link.sums[:c].clone())
    finally:
        ops._HEAD_DGRAD, ops._HEAD_DGRAD_MAX_K = saved
    ref = (torch.einsum("nkhw,kc->nchw", dy.double(), wt.double()[:, :, 0, 0]) + others.double()) * (z > 0)
    scale = float(ref.abs().max())
    for mode, (buf, sums) in outs.items():
        assert float((buf.double() - ref).abs().max()) <= 2e-6 * scale, mode
        assert float((buf * (z <= 0)).abs().max()) == 0.0
        exp = ref.sum((0, 2, 3))
        assert float((sums - exp).abs().max()) <= 2e-6 * float(ref.abs().sum((0, 2, 3)).max()), mode
    assert float((outs[True][0] - outs[False][0]).abs().max()) <= 2e-6 * scale
