"""GPU parity of the split-operand convolutions ("f16x3": csrc/conv_bf16.hip, rr_conv_*_f16x3; cfg.Model.conv_math).

The reference computes its convolutions in fp32 (backbones/hourglass.py:12-61 -> nn.Conv2d).  gfx950's fp32 matrix
instruction runs at 1/16 of the 16-bit rate; the split kernels write every fp32 operand as hi + lo, two fp16 values (22
significant bits) after a power-of-two scaling that puts the tensor's largest magnitude into [2^14, 2^15), and
accumulate hi*hi + hi*lo + lo*hi in fp32.  Contract, checked here:

  * against the fp32-MFMA kernels of csrc/conv.hip (pinned to torch / the reference goldens in tests/test_conv_gpu.py):
    max |diff| <= 4e-6 of the output scale — both sides carry fp32 summation noise of this size; a dropped cross product
    (hi*lo) shows up at 5e-4, a wrong scale at O(1);
  * against an fp64 convolution: the split kernel's error is not above 1.25x the fp32-MFMA kernel's own + 1e-7;
  * the result does not depend on the operands' magnitude (1e-6 ... 1e4: the scale is taken from the data), an all-zero
    operand gives exact zeros, a single huge outlier does not cost the small values more than 2^-22 of the OUTLIER's
    products (fp16's exponent range below the scaled maximum)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_conv_bf16_gpu import SHAPES, _mk  # noqa: E402  (the same layer shapes as the bf16 kernels' test)

BOUND = 4e-6


def _close(a, b, what, bound=BOUND):
    a, b = a.double(), b.double()
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= bound * max(scale, 1e-30), "%s: max |diff| %.3e vs scale %.3e (%.2e)" % (what, err, scale, err / max(scale, 1e-30))


@pytest.fixture()
def math_switch():
    from rrnet_amd import ops
    saved = ops.BF16

    def set_(v):
        ops.BF16 = v
    yield set_
    ops.BF16 = saved


@pytest.mark.parametrize("cfg", SHAPES, ids=lambda c: "n%dc%dh%dw%dk%dr%ds%d_s%d" % c[:8])
def test_split_kernels_equal_fp32_kernels(cfg, math_switch):
    from rrnet_amd import ops
    n, c, h, w, k, r, s, stride, ph, pw, use_bias, relu = cfg
    x = ops.to_nhwc(_mk((n, c, h, w), 1).cuda())
    wt = ops.to_nhwc((_mk((k, c, r, s), 2) * (1.0 / np.sqrt(c * r * s))).cuda())
    b = _mk((k,), 3).cuda() if use_bias else None
    math_switch(ops.MATH_F16X3)
    y, slab = ops.conv_fprop(x, wt, b, stride, (ph, pw), relu, want_stats=True)
    math_switch(ops.MATH_F32)
    y_ref, slab_ref = ops.conv_fprop(x, wt, b, stride, (ph, pw), relu, want_stats=True)
    _close(y, y_ref, "fprop")
    _close(slab.view(-1, 2, k).sum(0), slab_ref.view(-1, 2, k).sum(0), "fprop statistics", 2e-5)
    p, q = y.shape[2], y.shape[3]
    gy = ops.to_nhwc((_mk((n, k, p, q), 4) * 1e-4).cuda())          # gradient-sized values: the scale does the work
    saved = ops._DGRAD_VIA_FPROP_MIN_PIXELS
    ops._DGRAD_VIA_FPROP_MIN_PIXELS = 0
    try:
        if k % 4 == 0:
            math_switch(ops.MATH_F16X3)
            dx = ops.conv_dgrad(gy, wt, (n, c, h, w), stride, (ph, pw))
            base = ops.to_nhwc((_mk((n, c, h, w), 5) * 1e-4).cuda())
            acc = base.clone(memory_format=torch.channels_last)
            ops.conv_dgrad(gy, wt, (n, c, h, w), stride, (ph, pw), out=acc, accumulate=True)
            math_switch(ops.MATH_F32)
            dx_ref = ops.conv_dgrad(gy, wt, (n, c, h, w), stride, (ph, pw))
            _close(dx, dx_ref, "dgrad")
            _close(acc, dx_ref + base, "dgrad (accumulate)")
    finally:
        ops._DGRAD_VIA_FPROP_MIN_PIXELS = saved
    if c > 32 and k > 32 and k % 4 == 0:
        dw = torch.zeros((k, r, s, c), device="cuda").permute(0, 3, 1, 2)
        dw_ref = torch.zeros((k, r, s, c), device="cuda").permute(0, 3, 1, 2)
        math_switch(ops.MATH_F16X3)
        ops.conv_wgrad(x, gy, dw, stride, (ph, pw))
        math_switch(ops.MATH_F32)
        ops.conv_wgrad(x, gy, dw_ref, stride, (ph, pw))
        _close(dw, dw_ref, "wgrad", 1e-5)          # (pixel-split partial sums meet in float atomics on both sides)


def _fp64_conv(x, w, pad):
    cols = torch.nn.functional.unfold(x.double(), w.shape[2], padding=pad)
    return (w.double().reshape(w.shape[0], -1) @ cols).reshape(x.shape[0], w.shape[0], x.shape[2], x.shape[3])


@pytest.mark.parametrize("gain_x,gain_w", [(1.0, 1.0), (1e-6, 1.0), (1e4, 1e-3), (3e-5, 40.0)])
def test_split_error_against_fp64_is_not_above_the_fp32_kernels(gain_x, gain_w, math_switch):
    """n2 c256 128x128 k256 3x3: 512 tiles of 128 x 128 (the wave-specialised kernel), reduction over 2304 products."""
    from rrnet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(2, 256, 128, 128, device="cuda", generator=g) * torch.rand(2, 256, 1, 1, device="cuda", generator=g) * 3 * gain_x
    w = torch.randn(256, 256, 3, 3, device="cuda", generator=g) * 0.05 * gain_w
    ref = _fp64_conv(x, w, 1)
    rms = float(ref.pow(2).mean().sqrt())
    xc, wc = ops.to_nhwc(x), ops.to_nhwc(w)
    math_switch(ops.MATH_F16X3)
    e_split = float((ops.conv_fprop(xc, wc, None, 1, (1, 1), False).double() - ref).pow(2).mean().sqrt()) / rms
    math_switch(ops.MATH_F32)
    e_f32 = float((ops.conv_fprop(xc, wc, None, 1, (1, 1), False).double() - ref).pow(2).mean().sqrt()) / rms
    print("rms error / rms y: f16x3 %.3e   fp32 MFMA %.3e" % (e_split, e_f32))
    assert e_split <= 1.25 * e_f32 + 1e-7, (e_split, e_f32)
    assert e_split < 1e-6


def test_split_zero_operands_and_outliers(math_switch):
    from rrnet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(1, 128, 32, 32, device="cuda", generator=g)
    w = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.05
    math_switch(ops.MATH_F16X3)
    y0 = ops.conv_fprop(ops.to_nhwc(torch.zeros_like(x)), ops.to_nhwc(w), None, 1, (1, 1), False)
    assert float(y0.abs().max()) == 0.0
    y1 = ops.conv_fprop(ops.to_nhwc(x), ops.to_nhwc(torch.zeros_like(w)), None, 1, (1, 1), False)
    assert float(y1.abs().max()) == 0.0
    # one activation 1e6 times the rest: the scale follows it; the other products keep an ABSOLUTE error of 2^-22 of the
    # outlier's products at most, and far from the outlier (its 3x3 neighbourhood aside) the small values are still resolved
    xo = x.clone()
    xo[0, 5, 16, 16] = 1e6
    yo = ops.conv_fprop(ops.to_nhwc(xo), ops.to_nhwc(w), None, 1, (1, 1), False)
    ref = _fp64_conv(xo, w, 1)
    assert torch.isfinite(yo).all()
    err = (yo.double() - ref).abs()
    assert float(err.max()) <= 2.0 ** -21 * 1e6 * float(w.abs().max()), float(err.max())
    far = torch.ones_like(err, dtype=torch.bool)
    far[:, :, 14:19, 14:19] = False
    # fp16 at a scaled maximum of 2^14..2^15: values 1e-6 of it sit at ~2^-5, hi keeps 11 bits, lo is subnormal: ~2^-19 each
    assert float(err[far].max()) <= 2e-5 * float(ref[far].abs().max()), float(err[far].max() / ref[far].abs().max())
    # NaN / Inf propagate as in fp32
    xn = x.clone()
    xn[0, 0, 0, 0] = float("inf")
    yn = ops.conv_fprop(ops.to_nhwc(xn), ops.to_nhwc(w), None, 1, (1, 1), False)
    assert not torch.isfinite(yn[0, :, 0, 0]).any()


@pytest.mark.parametrize("relu,residual", [(True, False), (True, True), (False, False)])
def test_split_dgrad_carries_the_bn_backward_sums(relu, residual, math_switch):
    from rrnet_amd import ops
    n, c, h, w, k = 2, 256, 64, 64, 256
    y = ops.to_nhwc(_mk((n, c, h, w), 11).cuda())
    mean, invstd = y.mean((0, 2, 3)).contiguous(), (1.0 / (y.var((0, 2, 3), unbiased=False) + 1e-5).sqrt()).contiguous()
    scale, shift = invstd.clone(), (-mean * invstd).contiguous()
    res = ops.to_nhwc(_mk((n, c, h, w), 12).cuda()) if residual else None
    z = ops.bn_apply(y, scale, shift, res, relu)
    link = ops.BnLink()
    link.y, link.mean, link.invstd = y, mean, invstd
    link.use_z = bool(relu and residual)
    link.msc, link.msh = (scale, shift) if (relu and not residual) else (None, None)
    gy = ops.to_nhwc((_mk((n, k, h, w), 13) * 1e-3).cuda())
    wt = ops.to_nhwc((_mk((k, c, 3, 3), 14) * 0.02).cuda())
    math_switch(ops.MATH_F16X3)
    dx = ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (1, 1), bnsum=link, bnsum_z=z)
    math_switch(ops.MATH_F32)
    assert link.sums is not None and link.dz is dx
    ref = ops.bn_bwd_reduce(dx, z if link.use_z else None, y, mean, invstd, mask_scale=link.msc, mask_shift=link.msh)
    _close(link.sums[:2 * c], ref[:2 * c], "bn-backward sums", 1e-5)
    _close(dx, ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (1, 1)), "dgrad with sums")


def test_split_dgrad_relu_bias_epilogue(math_switch):
    from rrnet_amd import ops
    n, c, h, w, k = 2, 256, 64, 64, 10
    z = ops.to_nhwc(torch.relu(_mk((n, c, h, w), 21)).cuda())
    gy = ops.to_nhwc((_mk((n, k, h, w), 22) * 1e-3).cuda())
    wt = ops.to_nhwc((_mk((k, c, 1, 1), 23) * 0.05).cuda())
    link = ops.BnLink()
    link.relu_bias = link.use_z = True
    math_switch(ops.MATH_F16X3)
    dx = ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (0, 0), bnsum=link, bnsum_z=z)
    math_switch(ops.MATH_F32)
    ref = ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (0, 0)) * (z > 0)
    _close(dx, ref, "masked dgrad")
    _close(link.sums[:c], ref.double().sum((0, 2, 3)), "bias gradient", 1e-5)


def test_absmax_bits_is_the_maximum_magnitude():
    from rrnet_amd import _C
    g = torch.Generator(device="cuda").manual_seed(3)
    for n, gain in ((4, 1.0), (1 << 20, 1e-7), (12345 * 4, 1e9)):
        x = torch.randn(n, device="cuda", generator=g) * gain
        word = torch.zeros(2, dtype=torch.int32, device="cuda")
        _C.check(_C.fn("rr_absmax_bits")(_C.ptr(x), n, _C.ptr(word), _C.stream()), "rr_absmax_bits")
        assert word[:1].view(torch.float32).item() == x.abs().max().item()
        assert int(word[1]) == 0


def test_ready_made_filter_split_gives_the_same_numbers(math_switch):
    """rr_weight_split_f16 + the B16 instantiation (the filter split once per optimizer step instead of tile by tile inside the
    convolution): forward, data gradient with BN sums, on the flipped copy — identical products, so the results agree with
    the in-kernel split to fp32 summation noise, and with the fp32 kernels as everywhere else."""
    from rrnet_amd import _C, ops
    n, c, h, w, k = 2, 256, 96, 96, 256
    x = ops.to_nhwc(_mk((n, c, h, w), 31).cuda())
    wt = ops.to_nhwc((_mk((k, c, 3, 3), 32) * 0.02).cuda())
    gy = ops.to_nhwc((_mk((n, k, h, w), 33) * 1e-4).cuda())
    math_switch(ops.MATH_F16X3)
    saved = ops._SPLIT_PRESPLIT_PIXELS
    ops._SPLIT_PRESPLIT_PIXELS = 0
    try:
        word = ops.amax_of(wt)
        ws = ops.split_filter(wt, word, n * h * w)
        assert ws is not None and ws.numel() == 2 * wt.numel()
        # the images: hi + lo reproduce the scaled filter to 2^-22
        scale = 2.0 ** (14 - int(np.floor(np.log2(float(wt.abs().max())))))
        flat = wt.permute(0, 2, 3, 1).reshape(-1)
        rec = (ws[:flat.numel()].double() + ws[flat.numel():].double()) / scale
        assert float((rec - flat.double()).abs().max()) <= 2.0 ** -21 * float(flat.abs().max())
        y1 = ops.conv_fprop(x, wt, None, 1, (1, 1), False, w_split=ws)
        y0 = ops.conv_fprop(x, wt, None, 1, (1, 1), False)
        flip = torch.empty(wt.numel(), dtype=torch.float32, device="cuda")
        _C.check(_C.fn("rr_weight_flip_transpose")(_C.ptr(wt), _C.ptr(flip), k, c, 3, 3, _C.stream()), "flip")
        wts = ops.split_filter(wt, word, n * h * w, flat=flip)
        d1 = ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (1, 1), wt=flip, wt_split=wts)
        d0 = ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (1, 1), wt=flip)
    finally:
        ops._SPLIT_PRESPLIT_PIXELS = saved
    math_switch(ops.MATH_F32)
    _close(y1, y0, "fprop, ready-made split vs in-kernel split", 1e-6)
    _close(d1, d0, "dgrad, ready-made split vs in-kernel split", 1e-6)
    _close(y1, ops.conv_fprop(x, wt, None, 1, (1, 1), False), "fprop vs fp32 kernel")
    _close(d1, ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (1, 1)), "dgrad vs fp32 kernel")


def test_kernels_follow_the_oracle_of_the_split_arithmetic(math_switch):
    """oracle/split.py models the contract step by step (scale from the maximum's exponent, round-to-nearest fp16 hi / lo,
    three exact products, exact accumulation): rr_absmax_bits and rr_weight_split_f16 reproduce it BIT FOR BIT, and the
    convolution differs from it only by its fp32 accumulation (<= 4e-7 of the output scale at a reduction length of 576;
    against the fp32-MFMA kernels the bound is 4e-6)."""
    from oracle import split as osp
    from rrnet_amd import _C, ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(2, 64, 24, 24, generator=g) * torch.rand(2, 64, 1, 1, generator=g) * 3 * 1e-3)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    xd, wd = ops.to_nhwc(x.cuda()), ops.to_nhwc(w.cuda())
    # the maximum, as a bit pattern
    word = ops.amax_of(wd)
    assert int(word.view(torch.int32)[0]) == osp.absmax_bits(w.numpy())
    # the filter's two images
    saved = (ops._SPLIT_PRESPLIT_PIXELS, ops._SPLIT_MIN_PIXELS, ops._SPLIT_MIN_K, ops._SPLIT_MIN_CH)
    ops._SPLIT_PRESPLIT_PIXELS = ops._SPLIT_MIN_PIXELS = ops._SPLIT_MIN_K = ops._SPLIT_MIN_CH = 0
    try:
        img = torch.empty(2 * wd.numel(), dtype=torch.float16, device="cuda")
        _C.check(_C.fn("rr_weight_split_f16")(_C.ptr(wd), wd.numel(), _C.ptr(word), _C.ptr(img), _C.stream()), "split")
        hi, lo = osp.split(w.permute(0, 2, 3, 1).reshape(-1).numpy(), osp.scale_of(osp.absmax_bits(w.numpy())))
        got = img.cpu().numpy()
        assert np.array_equal(got[:hi.size].view(np.uint16), hi.view(np.uint16))
        assert np.array_equal(got[hi.size:].view(np.uint16), lo.view(np.uint16))
        math_switch(ops.MATH_F16X3)
        for stride in (1, 2):
            y = ops.conv_fprop(xd, wd, None, stride, (1, 1), False).double().cpu().numpy()
            ref = osp.conv2d(x.numpy(), w.numpy(), stride, (1, 1))
            err = np.abs(y - ref).max() / np.abs(ref).max()
            print("stride %d: kernel vs oracle of the split arithmetic %.2e" % (stride, err))
            assert err <= 4e-7, (stride, err)
    finally:
        ops._SPLIT_PRESPLIT_PIXELS, ops._SPLIT_MIN_PIXELS, ops._SPLIT_MIN_K, ops._SPLIT_MIN_CH = saved
