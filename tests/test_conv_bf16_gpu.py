"""GPU parity of the bf16-operand convolutions (csrc/conv_bf16.hip; BASELINE config 4 "bf16").

The reference is fp32-only (backbones/hourglass.py:12-61 -> nn.Conv2d), so the precision is builder-defined and its
contract is stated against our own fp32 kernels: a bf16 entry point rounds both operands of every product to bf16
(round-to-nearest-even) and accumulates in fp32, hence

    rr_conv_*_bf16(x, w)  ==  rr_conv_*(bf16(x), bf16(w))      up to the summation order

(a product of two bf16 values is exact in fp32).  Bound: 2e-5 of the output scale (max |ref|) — fp32 re-association over
<= 4608 terms (fprop / dgrad) or <= 16 k pixels per split (wgrad); it is NOT a bf16-sized bound: an operand that was
left unrounded, or rounded twice, or a dropped K-step shows up at >= 1e-3.  The fp32 kernels themselves are pinned
against torch / the reference goldens in tests/test_conv_gpu.py and tests/test_model_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (N, C, H, W, K, R, S, stride, pad_h, pad_w, bias, relu)
SHAPES = [
    (2, 256, 16, 16, 256, 3, 3, 1, 1, 1, False, False),
    (1, 64, 9, 13, 128, 3, 3, 1, 1, 1, False, False),       # ragged M / N tiles
    (2, 128, 16, 16, 256, 3, 3, 2, 1, 1, False, False),     # stride-2 3x3 (dgrad: one launch per output parity class)
    (2, 64, 15, 17, 128, 3, 3, 2, 1, 1, False, False),      # ... odd H and W: the four classes have different sizes
    (1, 64, 12, 12, 64, 3, 3, 2, 0, 0, False, False),       # ... no padding
    (2, 128, 15, 17, 256, 1, 1, 2, 0, 0, False, False),     # stride-2 1x1 skip, odd size
    (2, 256, 8, 8, 384, 1, 1, 1, 0, 0, False, False),
    (1, 256, 12, 12, 10, 1, 1, 1, 0, 0, True, False),       # hm head 1x1 (BN = 32 tile, N masked)
    (1, 256, 10, 10, 256, 3, 3, 1, 1, 1, True, True),       # head 3x3 + bias + ReLU
    (7, 256, 3, 3, 64, 1, 1, 1, 0, 0, False, False),        # stage-2 bottleneck on RoIs (BN = 64 tile)
    (2, 48, 8, 8, 24, 3, 3, 1, 1, 1, False, False),         # channels not a multiple of 32 (partial K-step)
    (1, 512, 4, 4, 512, 3, 3, 1, 1, 1, False, False),       # split-K
    (2, 160, 32, 32, 128, 1, 1, 1, 0, 0, False, False),     # the packed stem (147 -> 160 taps) as a 1x1
    (2, 256, 64, 64, 256, 3, 3, 1, 1, 1, False, False),     # dominant layer type
    (1, 256, 96, 96, 256, 3, 3, 1, 1, 1, True, True),
    (2, 128, 128, 128, 256, 3, 3, 2, 1, 1, False, False),
    (1, 384, 32, 32, 384, 3, 3, 1, 1, 1, False, False),     # 128x64 tiles, split-K
    (2, 256, 64, 64, 36, 1, 1, 1, 0, 0, False, False),      # WH head's fused 1x1
    (1, 256, 66, 70, 256, 3, 3, 1, 1, 1, False, False),     # Q % 32 != 0
]


def _mk(shape, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _close(a, b, what, bound=2e-5):
    a, b = a.float(), b.float()
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= bound * max(scale, 1e-30), "%s: max |diff| %.3e vs scale %.3e (%.2e)" % (what, err, scale, err / max(scale, 1e-30))


@pytest.fixture()
def bf16_switch():
    from rrnet_amd import ops
    saved = ops.BF16

    def set_(v):
        ops.BF16 = v
    yield set_
    ops.BF16 = saved


@pytest.mark.parametrize("cfg", SHAPES, ids=lambda c: "n%dc%dh%dw%dk%dr%ds%d_s%d" % c[:8])
def test_bf16_kernels_equal_fp32_kernels_on_rounded_operands(cfg, bf16_switch):
    from rrnet_amd import ops
    n, c, h, w, k, r, s, stride, ph, pw, use_bias, relu = cfg
    x = ops.to_nhwc(_mk((n, c, h, w), 1).cuda())
    wt = ops.to_nhwc((_mk((k, c, r, s), 2) * (1.0 / np.sqrt(c * r * s))).cuda())
    b = _mk((k,), 3).cuda() if use_bias else None
    xr, wr = ops.to_nhwc(_r(x)), ops.to_nhwc(_r(wt))
    # ---- forward (+ the BatchNorm partial sums of its epilogue)
    bf16_switch(True)
    y, slab = ops.conv_fprop(x, wt, b, stride, (ph, pw), relu, want_stats=True)
    bf16_switch(False)
    y_ref, slab_ref = ops.conv_fprop(xr, wr, b, stride, (ph, pw), relu, want_stats=True)
    _close(y, y_ref, "fprop")
    if c % 8 == 0:      # the filter handed over already rounded to bf16 (FlatParams' per-step copies): same numbers
        bf16_switch(True)
        w16 = wt.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)          # [k][r][s][c] memory order
        y16 = ops.conv_fprop(x, wt, b, stride, (ph, pw), relu, w16=w16)
        bf16_switch(False)
        _close(y16, y_ref, "fprop (bf16 filter source)")
    st, st_ref = slab.view(-1, 2, k).sum(0), slab_ref.view(-1, 2, k).sum(0)
    _close(st, st_ref, "fprop statistics", 1e-4)
    # the rounding is really there: against the UNROUNDED fp32 kernel the difference is bf16-sized
    y32 = ops.conv_fprop(x, wt, b, stride, (ph, pw), relu)
    d = float((y - y32).abs().max() / y32.abs().max())
    assert 1e-4 < d < 3e-2, d
    # ---- data gradient (stride 1: the forward kernel on flipped weights; stride 2 stays on the fp32 dgrad kernel)
    p, q = y.shape[2], y.shape[3]
    gy = ops.to_nhwc(_mk((n, k, p, q), 4).cuda())
    gyr = ops.to_nhwc(_r(gy))
    saved = ops._DGRAD_VIA_FPROP_MIN_PIXELS
    ops._DGRAD_VIA_FPROP_MIN_PIXELS = 0
    try:
        if stride in (1, 2) and k % 4 == 0:
            bf16_switch(True)
            dx = ops.conv_dgrad(gy, wt, (n, c, h, w), stride, (ph, pw))
            base = ops.to_nhwc(_mk((n, c, h, w), 5).cuda())
            acc = base.clone(memory_format=torch.channels_last)
            ops.conv_dgrad(gy, wt, (n, c, h, w), stride, (ph, pw), out=acc, accumulate=True)
            bf16_switch(False)
            dx_ref = ops.conv_dgrad(gyr, wr, (n, c, h, w), stride, (ph, pw))
            _close(dx, dx_ref, "dgrad")
            _close(acc, dx_ref + base, "dgrad (accumulate)")
    finally:
        ops._DGRAD_VIA_FPROP_MIN_PIXELS = saved
    # ---- weight gradient
    if c > 32 and k > 32 and k % 4 == 0:
        dw = torch.zeros((k, r, s, c), device="cuda").permute(0, 3, 1, 2)
        dw_ref = torch.zeros((k, r, s, c), device="cuda").permute(0, 3, 1, 2)
        bf16_switch(True)
        ops.conv_wgrad(x, gy, dw, stride, (ph, pw))
        bf16_switch(False)
        ops.conv_wgrad(xr, gyr, dw_ref, stride, (ph, pw))
        _close(dw, dw_ref, "wgrad", 5e-5)
        dw32 = torch.zeros((k, r, s, c), device="cuda").permute(0, 3, 1, 2)
        ops.conv_wgrad(x, gy, dw32, stride, (ph, pw))
        d = float((dw - dw32).abs().max() / dw32.abs().max())
        assert 1e-4 < d < 3e-2, d


@pytest.mark.parametrize("relu,residual", [(True, False), (True, True), (False, False)])
def test_bf16_dgrad_carries_the_bn_backward_sums(relu, residual, bf16_switch):
    """rr_conv_dgrad_s1_bnsum_bf16: the producer's BatchNorm-backward sums out of the bf16 data gradient's epilogue equal
    the separate reduce pass over the gradient it stored (same check as the fp32 variant's, tests/test_model_gpu.py)."""
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
    gy = ops.to_nhwc(_mk((n, k, h, w), 13).cuda())
    wt = ops.to_nhwc((_mk((k, c, 3, 3), 14) * 0.02).cuda())
    saved = ops._CONV16
    try:
        ops._CONV16 = False                 # the round-4 kernel: its epilogue carries the sums
        bf16_switch(True)
        dx = ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (1, 1), bnsum=link, bnsum_z=z)
        bf16_switch(False)
        assert link.sums is not None and link.dz is dx
        ref = ops.bn_bwd_reduce(dx, z if link.use_z else None, y, mean, invstd, mask_scale=link.msc, mask_shift=link.msh)
        _close(link.sums[:2 * c], ref[:2 * c], "bn-backward sums", 1e-5)
        dx_ref = ops.conv_dgrad(ops.to_nhwc(_r(gy)), ops.to_nhwc(_r(wt)), (n, c, h, w), 1, (1, 1))
        _close(dx, dx_ref, "dgrad with sums")
        # csrc/conv16.hip takes the same launch by default and does NOT carry the sums: the link stays empty, so the producer
        # runs its own reduce pass (functional._ConvBnAct._backward) — same gradient
        ops._CONV16 = True
        link.sums = link.dz = None
        bf16_switch(True)
        dx16 = ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (1, 1), bnsum=link, bnsum_z=z)
        bf16_switch(False)
        assert link.sums is None and link.dz is None
        _close(dx16, dx_ref, "conv16 dgrad")
    finally:
        ops._CONV16 = saved


def test_bf16_dgrad_relu_bias_epilogue(bf16_switch):
    """Heads (conv + bias + ReLU producer): masked store + bias column sums under cfg.Model.bf16.  The narrow 1x1 layers
    (K = 10 / 2 / 34) run rr_head_dgrad_relubias in every arithmetic — an element-wise fp32 pass, operands NOT rounded (closer to
    the reference than the bf16 contract asks); the implicit-GEMM form rr_conv_dgrad_s1_relubias_bf16 (ops._HEAD_DGRAD = False,
    and every wider layer) multiplies bf16-rounded operands."""
    from rrnet_amd import ops
    n, c, h, w, k = 2, 256, 64, 64, 10
    z = ops.to_nhwc(torch.relu(_mk((n, c, h, w), 21)).cuda())
    gy = ops.to_nhwc(_mk((n, k, h, w), 22).cuda())
    wt = ops.to_nhwc((_mk((k, c, 1, 1), 23) * 0.05).cuda())
    saved = ops._HEAD_DGRAD
    try:
        for head_kernel in (True, False):
            ops._HEAD_DGRAD = head_kernel
            link = ops.BnLink()
            link.relu_bias = link.use_z = True
            bf16_switch(True)
            dx = ops.conv_dgrad(gy, wt, (n, c, h, w), 1, (0, 0), bnsum=link, bnsum_z=z)
            bf16_switch(False)
            q = (lambda t: t) if head_kernel else _r
            ref = ops.conv_dgrad(ops.to_nhwc(q(gy)), ops.to_nhwc(q(wt)), (n, c, h, w), 1, (0, 0)) * (z > 0)
            _close(dx, ref, "masked dgrad (head kernel %s)" % head_kernel)
            _close(link.sums[:c], ref.double().sum((0, 2, 3)), "bias gradient", 1e-5)
    finally:
        ops._HEAD_DGRAD = saved
