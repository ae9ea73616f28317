"""Test helper: audits every kernel call of a run against a host (torch fp64) recomputation ON THE SAME INPUTS.

`with audit() as rec:` patches the conv / BatchNorm / elementwise wrappers of rrnet_amd.ops; each DISTINCT call
signature (kind, shapes, stride, padding, flags) is recomputed on the host the first time it is seen and the
relative error (max |diff| / max |ref|) is recorded in rec.seen; rec.bad lists the calls over tolerance.  Because
every call is checked on the inputs the previous kernels actually produced, the check is as tight as a unit test
(1e-6 level) even inside a 100-layer train step whose end-to-end gradients are chaotic.

`audit(sample=True)` is the mode for the BASELINE launch configuration itself (B=8, 1024x1024: 8x256x256x256 maps,
537 MB per tensor), where a full host recomputation of every call would take hours.  Calls whose host reference costs
more than SAMPLE_FLOPS are checked on samples that keep the property under test:
  * fprop / dgrad are local: three 16-row bands (top border, an interior band that straddles tile boundaries, bottom
    border) of three different images (first, middle, last) are recomputed in fp64 from the sliced input rows;
  * wgrad reduces over ALL pixels (where the round-2 row-walk defect lived): 16 filters (the first 4, 8 around the
    128-row tile boundary, the last 4) are recomputed in fp64 over the full pixel range, all channels and taps; for the
    largest layers the FULL dw is compared with torch-CPU's fp32 conv2d_weight as well;
  * element-wise BatchNorm / ReLU / fan-in kernels are compared on the first and last image; the reductions
    (BatchNorm statistics, backward sums, bias gradients) always run over the whole tensor in fp64."""
import contextlib

import torch
import torch.nn.functional as F

# Where the fp64 reference arithmetic runs.  "cpu": the host (torch CPU kernels) — the default, used at the small sizes.
# "cuda": torch's own fp64 kernels on the device (ATen element-wise / reduction kernels; double-precision conv2d / its
# gradients through ATen's slow_conv2d = unfold + rocBLAS dgemm — MIOpen has no fp64): an implementation that shares
# nothing with csrc/, ~50x faster than 8 host cores and without the 0.5-1 GB device->host copies.  The three full-size
# audits (B=8, 1024x1024) run with it: on the host they took 150 s EACH of the driver's 1200 s GPU-test limit.
REF = {"dev": "cpu"}


def _to_ref(t):
    return t.detach().to(REF["dev"])


def _zeros64(n):
    return torch.zeros(n, dtype=torch.float64, device=REF["dev"])


SAMPLE_FLOPS = 2.0e10          # host-reference cost above which a call is sampled (sample=True only)
FULL_FP32_WGRAD_FLOPS = 2.5e11   # sampled wgrad calls at least this large also get a full-dw fp32 host reference


class Record:
    def __init__(self):
        self.seen = {}
        self.bad = []

    def note(self, kind, sig, err, tol):
        key = (kind,) + tuple(sig)
        if key in self.seen and err <= self.seen[key]:
            return
        self.seen[key] = err
        if not err <= tol:
            self.bad.append((key, err))


def _rel(a, b):
    a = a.detach().to(REF["dev"]).double()
    b = b.detach().to(REF["dev"]).double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _V(t):
    return t.detach().to(REF["dev"]).double().view(1, -1, 1, 1)


def _c64(t):
    return t.detach().to(REF["dev"]).double()


def _q64(t, on):
    """fp64 copy of a convolution operand; with `on` rounded to bf16 first — the contract of the bf16-operand kernels
    (csrc/conv_bf16.hip: both operands of every product rounded to nearest even, fp32 accumulation)."""
    t = t.detach().to(REF["dev"])
    return (t.to(torch.bfloat16).to(torch.float32) if on else t).double()


def _bands(p):
    """Output-row bands of a sampled convolution check: (image selector in [0,1], p0, p1)."""
    if p <= 48:
        return [(0.0, 0, p)]
    mid = (p // 2 // 32) * 32 + 24           # crosses a 32-row boundary of the walk
    return [(0.0, 0, 16), (0.5, mid, min(mid + 16, p)), (1.0, p - 16, p)]


def _fprop_band(x, w64, b64, stride, pad, n0, p0, p1, quant=False):
    """fp64 conv2d of output rows [p0,p1) of image n0, from the input rows they reach (zero rows beyond the map)."""
    h = x.shape[2]
    r = w64.shape[2]
    h0, h1 = p0 * stride - pad[0], (p1 - 1) * stride - pad[0] + r
    a, b = max(h0, 0), min(h1, h)
    xs = F.pad(_q64(x[n0:n0 + 1, :, a:b, :], quant), (0, 0, a - h0, h1 - b))
    return F.conv2d(xs, w64, b64, stride, (0, pad[1]))


def _dgrad_band(dy, w64, x_shape, stride, pad, n0, h0, h1, quant=False):
    """fp64 data gradient rows [h0,h1) of image n0 from the dy rows that reach them."""
    _, c, h, wd = x_shape
    k, _, r, s = w64.shape
    p, q = dy.shape[2], dy.shape[3]
    pa = max(0, -((-(h0 + pad[0] - r + 1)) // stride))
    pb = min(p, (h1 - 1 + pad[0]) // stride + 1)
    ref = torch.zeros((1, c, h1 - h0, wd), dtype=torch.float64, device=REF["dev"])
    if pb <= pa:
        return ref
    opw = wd - ((q - 1) * stride - 2 * pad[1] + s)
    full = F.conv_transpose2d(_q64(dy[n0:n0 + 1, :, pa:pb, :], quant), w64, None, stride, (0, pad[1]), (0, opw))
    base = pa * stride - pad[0]                       # input row of full's row 0
    lo, hi = max(h0, base), min(h1, base + full.shape[2])
    if hi > lo:
        ref[:, :, lo - h0:hi - h0] = full[:, :, lo - base:hi - base]
    return ref


def _k_sample(k):
    if k <= 16:
        return list(range(k))
    mids = [m for m in (128, 256, 384) if m + 4 < k] or [k // 2]     # wgrad tiles are 128 filters tall
    pick = list(range(4)) + list(range(k - 4, k))
    for m in mids[:2]:
        pick += list(range(m - 4, m + 4))
    return sorted(set(pick))


def _img_sample(n):
    return sorted({0, n - 1})


def _colsum64(t):
    """Per-channel sum over (N,H,W) of a logical NCHW tensor, fp64 accumulation, without an fp64 copy of the tensor."""
    return t.detach().to(REF["dev"]).sum((0, 2, 3), dtype=torch.float64)


@contextlib.contextmanager
def audit(tol=2e-5, tol_wgrad=2e-4, sample=False, ref_device="cpu"):
    from rrnet_amd import ops
    REF["dev"] = ref_device
    tol_ = tol
    rec = Record()
    rec.sampled = set()
    names = ("conv_fprop", "conv_dgrad", "conv_wgrad", "stem_wgrad_s2d", "conv_fprop_packed", "conv_wgrad_packed", "bn_apply", "bn_bwd_reduce", "bn_bwd_apply",
             "sum_n", "upsample_add_fwd", "upsample_add_bwd", "bias_relu_bwd", "relu_fwd", "bn_finalize",
             "bn_stats_finalize", "dcn_fwd", "dcn_dgrad", "dcn_wgrad")
    orig = {n: getattr(ops, n) for n in names}

    def big(flops):
        return sample and flops > SAMPLE_FLOPS

    def _dat(t):
        """A bf16-only tensor (ops.phantom_f32: an fp32 handle without memory) -> its bf16 image, the data the kernel read / wrote."""
        return ops.image_of(t) if (t is not None and ops.is_phantom(t)) else t

    def big_elems(t):
        return sample and t.numel() > (1 << 24)

    packed_src = {}      # data_ptr of a packed tap image -> (x, stride, pad) it was made from

    def conv_fprop_packed(x, w, stride, pad, want_stats=False):
        """The few-channel stem through tap packing + 1x1 GEMM: checked END TO END against conv2d of the original image
        (the inner pack / GEMM launches are audited as ordinary calls on top of that)."""
        y, slab, xp = orig["conv_fprop_packed"](x, w, stride, pad, want_stats)
        packed_src[xp.data_ptr()] = (x, stride, tuple(pad))
        sig = (tuple(x.shape), tuple(w.shape), stride, tuple(pad))
        if ("fprop_packed",) + sig not in rec.seen:
            qq = ops._bf16_ok(xp.shape[1], 4, 1, 1, xp, y)        # the inner 1x1 GEMM rounds the packed taps = rounds x, w
            w64 = _q64(w, qq)
            flops = 2.0 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
            if big(flops):
                rec.sampled.add(("fprop_packed",) + sig)
                err = 0.0
                for sel, p0, p1 in _bands(y.shape[2]):
                    n0 = int(round(sel * (x.shape[0] - 1)))
                    err = max(err, _rel(y[n0:n0 + 1, :, p0:p1, :], _fprop_band(x, w64, None, stride, pad, n0, p0, p1, qq)))
            else:
                err = _rel(y, F.conv2d(_q64(x, qq), w64, None, stride, tuple(pad)))
            rec.note("fprop_packed", sig, err, tol)
        return y, slab, xp

    def conv_wgrad_packed(xp, dy, dw):
        x, stride, pad = packed_src.get(xp.data_ptr(), (None, None, None))
        sig = (tuple(xp.shape), tuple(dy.shape), tuple(dw.shape))
        check = x is not None and ("wgrad_packed",) + sig not in rec.seen
        base = dw.clone() if check else None
        res = orig["conv_wgrad_packed"](xp, dy, dw)
        if check:
            got = _c64(res) - _c64(base)
            flops = 2.0 * dy.numel() * dw.shape[1] * dw.shape[2] * dw.shape[3]
            ks = _k_sample(dw.shape[0]) if big(flops) else list(range(dw.shape[0]))
            if big(flops):
                rec.sampled.add(("wgrad_packed",) + sig)
            qq = ops._bf16_ok(xp.shape[1], dy.shape[1], 1, 1, xp, dy) and xp.shape[1] > 32 and dy.shape[1] > 32
            ref = torch.nn.grad.conv2d_weight(_q64(x, qq), (len(ks),) + tuple(dw.shape[1:]), _q64(dy[:, ks], qq), stride, pad)
            rec.note("wgrad_packed", sig, float((got[ks] - ref).abs().max() / max(float(ref.abs().max()), 1e-30)), tol_wgrad)
        return res

    def conv_fprop(x, w, bias=None, stride=1, pad=(0, 0), relu=False, want_stats=False, algo_kg=None, w16=None, w_split=None,
                   y_bf16_only=False):
        out = orig["conv_fprop"](x, w, bias, stride, pad, relu, want_stats, algo_kg, w16, w_split, y_bf16_only)   # w16 / w_split: cached copies of w
        x = _dat(x)
        y = out[0] if want_stats else out
        y_img = ops.is_phantom(y)        # the output exists only as its bf16 image: the fp64 value up to one bf16 rounding
        y = _dat(y)
        tol = 2.0 ** -8 if y_img else tol_
        sig = (tuple(x.shape), tuple(w.shape), stride, tuple(pad), bias is not None, relu, want_stats) + (("y-bf16-only",) if y_img else ())
        flops = 2.0 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        qq = ops._bf16_ok(w.shape[1], 4, w.shape[2], w.shape[3], x, w, y)
        if qq:
            sig = sig + ("bf16",)
        if ("fprop",) + sig not in rec.seen and big(flops):
            rec.sampled.add(("fprop",) + sig)
            w64, b64 = _q64(w, qq), None if bias is None else _c64(bias)
            err = 0.0
            for sel, p0, p1 in _bands(y.shape[2]):
                n0 = int(round(sel * (x.shape[0] - 1)))
                ref = _fprop_band(x, w64, b64, stride, pad, n0, p0, p1, qq)
                if relu:
                    ref = ref.relu()
                err = max(err, _rel(y[n0:n0 + 1, :, p0:p1, :], ref))
            rec.note("fprop", sig, err, tol)
            if want_stats:      # the statistics are a function of y (checked above): compare with fp64 sums of y itself
                k = w.shape[0]
                sums = ops.bn_reduce_slab(out[1], k).to(REF["dev"])
                yc = y.detach().to(REF["dev"])
                s1 = yc.sum((0, 2, 3), dtype=torch.float64)
                s2 = _zeros64(k)
                sa = _zeros64(k)
                for i in range(yc.shape[0]):
                    yi = yc[i].double()
                    s2 += (yi * yi).sum((1, 2))
                    sa += yi.abs().sum((1, 2))
                e1 = float((sums[:k] - s1).abs().max() / max(float(sa.max()), 1e-30))
                # (a bf16-only y: the kernel's sums come from the fp32 accumulators, these from the rounded image)
                rec.note("fprop_stats", sig, max(e1, float((sums[k:2 * k] - s2).abs().max() / float(s2.max()))), 1e-4 if y_img else tol)
        elif ("fprop",) + sig not in rec.seen:
            ref = F.conv2d(_q64(x, qq), _q64(w, qq), None if bias is None else bias.to(REF["dev"]).double(), stride, pad)
            if relu:
                ref = ref.relu()
            rec.note("fprop", sig, _rel(y, ref), tol)
            if want_stats:
                k = w.shape[0]
                sums = ops.bn_reduce_slab(out[1], k).to(REF["dev"])
                s1, s2 = ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))
                e1 = float((sums[:k] - s1).abs().max() / max(float(ref.abs().sum((0, 2, 3)).max()), 1e-30))
                rec.note("fprop_stats", sig, max(e1, _rel(sums[k:2 * k], s2)), tol)
        return out

    def _data(t):
        """A bf16-only tensor (ops.phantom_f32: an fp32 handle without memory) -> its bf16 image, the data the kernel read / wrote."""
        return ops.image_of(t) if ops.is_phantom(t) else t

    def conv_dgrad(dy, w, x_shape, stride=1, pad=(0, 0), out=None, accumulate=False, bnsum=None, bnsum_z=None, wt=None, wt16=None,
                   wt_split=None):
        base = out.clone() if (out is not None and accumulate) else None
        res = orig["conv_dgrad"](dy, w, x_shape, stride, pad, out, accumulate, bnsum, bnsum_z, wt, wt16, wt_split)   # the cached flipped filters
        dy, bnsum_z = _dat(dy), _dat(bnsum_z)
        relu_mask = None          # conv + bias + ReLU producer: this launch stored the masked gradient
        if bnsum is not None and bnsum.relu_bias and bnsum.sums is not None and bnsum.dz is res:
            relu_mask = bnsum_z
            bsig = (tuple(res.shape), tuple(dy.shape))
            # (conv16's ReLU-mask epilogue — a bare ReLU producer — leaves a two-word marker instead of column sums)
            if ("dgrad_relubias",) + bsig not in rec.seen and bnsum.sums.numel() >= 2 * res.shape[1]:
                c = res.shape[1]
                tot = _colsum64(res)
                mag = res.detach().to(REF["dev"]).abs().sum((0, 2, 3), dtype=torch.float64)
                o = bnsum.sums.to(REF["dev"])
                e = float((o[:c] - tot).abs().max() / max(float(mag.max()), 1e-30))
                # nothing may survive where the producer's output is zero
                e = max(e, float((res * (relu_mask <= 0)).abs().max()))
                rec.note("dgrad_relubias", bsig, e, tol)
        elif bnsum is not None and bnsum.sums is not None and bnsum.dz is res:
            # the producer's BatchNorm-backward sums from the epilogue: compare with fp64 sums over the whole of the
            # gradient this launch left in memory (the gradient itself is checked below)
            zt = bnsum_z if bnsum.use_z else None
            yb = _dat(bnsum.y)            # (conv16: the producer's pre-BN output may exist only as its bf16 image)
            bsig = (tuple(res.shape), zt is not None, bnsum.msc is not None, bool(accumulate)) + \
                   (("y-image",) if ops.is_phantom(bnsum.y) else ())
            if ("dgrad_bnsum",) + bsig not in rec.seen:
                c = res.shape[1]
                s1, s2, a1, a2 = (_zeros64(c) for _ in range(4))
                for i in range(res.shape[0]):
                    d = _masked(res[i:i + 1], None if zt is None else zt[i:i + 1], yb[i:i + 1], bnsum.msc, bnsum.msh)
                    xh = (_c64(yb[i:i + 1]) - _V(bnsum.mean)) * _V(bnsum.invstd)
                    s1 += d.sum((0, 2, 3)); a1 += d.abs().sum((0, 2, 3))
                    d = d * xh
                    s2 += d.sum((0, 2, 3)); a2 += d.abs().sum((0, 2, 3))
                o = bnsum.sums.to(REF["dev"])
                e1 = float((o[:c] - s1).abs().max() / max(float(a1.max()), 1e-30))
                e2 = float((o[c:2 * c] - s2).abs().max() / max(float(a2.max()), 1e-30))
                rec.note("dgrad_bnsum", bsig, max(e1, e2), tol)
        sig = (tuple(dy.shape), tuple(w.shape), tuple(x_shape), stride, tuple(pad), bool(accumulate)) + \
              (("relu-masked",) if relu_mask is not None else ())
        flops = 2.0 * dy.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        # bf16 operands: stride-1 layers (the forward kernel on the flipped filter); a 10- / 2-channel dy was zero-padded to 12 / 4
        qq = (stride == 1 or (stride == 2 and ops._BF16_S2_DGRAD)) and pad[0] < w.shape[2] and pad[1] < w.shape[3] and \
            ops._bf16_ok(w.shape[0] if relu_mask is None else (w.shape[0] + 3) // 4 * 4, w.shape[1], w.shape[2], w.shape[3], dy, res)
        if relu_mask is not None and w.shape[2] == 1 and w.shape[3] == 1 and w.shape[0] <= ops._HEAD_DGRAD_MAX_K and ops._HEAD_DGRAD:
            qq = 0            # rr_head_dgrad_relubias: an fp32 element-wise pass in every arithmetic, operands not rounded
        if qq:
            sig = sig + ("bf16",)
        if ("dgrad",) + sig not in rec.seen and big(flops):
            rec.sampled.add(("dgrad",) + sig)
            w64 = _q64(w, qq)
            err = 0.0
            for sel, h0, h1 in _bands(x_shape[2]):
                n0 = int(round(sel * (x_shape[0] - 1)))
                ref = _dgrad_band(dy, w64, x_shape, stride, pad, n0, h0, h1, qq)
                if base is not None:
                    ref = ref + _c64(base[n0:n0 + 1, :, h0:h1, :])
                if relu_mask is not None:
                    ref = ref * (relu_mask[n0:n0 + 1, :, h0:h1, :].detach().to(REF["dev"]) > 0)
                err = max(err, _rel(res[n0:n0 + 1, :, h0:h1, :], ref))
            rec.note("dgrad", sig, err, tol)
        elif ("dgrad",) + sig not in rec.seen:
            ref = torch.nn.grad.conv2d_input(tuple(x_shape), _q64(w, qq), _q64(dy, qq), stride, pad)
            if base is not None:
                ref = ref + base.to(REF["dev"]).double()
            if relu_mask is not None:
                ref = ref * (relu_mask.detach().to(REF["dev"]) > 0)
            rec.note("dgrad", sig, _rel(res, ref), tol)
        return res

    def conv_wgrad(x, dy, dw, stride=1, pad=(0, 0), explicit_out=False, algo_c=None):
        sig = (tuple(x.shape), tuple(dy.shape), tuple(dw.shape), stride, tuple(pad), explicit_out)
        qq = ops._bf16_ok(x.shape[1], dy.shape[1], dw.shape[2], dw.shape[3], x, dy) and \
            ops.wgrad_16bit_shape(x.shape[1], dy.shape[1], dw.shape[2], dw.shape[3])
        if qq:
            sig = sig + ("bf16",)
        check = ("wgrad",) + sig not in rec.seen and not explicit_out
        base = dw.clone() if check else None
        res = orig["conv_wgrad"](x, dy, dw, stride, pad, explicit_out, algo_c)
        dy, x = _dat(dy), _dat(x)
        flops = 2.0 * dy.numel() * dw.shape[1] * dw.shape[2] * dw.shape[3]
        if check and big(flops):
            rec.sampled.add(("wgrad",) + sig)
            got = _c64(res) - _c64(base)
            ks = _k_sample(dw.shape[0])
            x64 = _q64(x, qq)
            ref = torch.nn.grad.conv2d_weight(x64, (len(ks),) + tuple(dw.shape[1:]), _q64(dy[:, ks], qq), stride, pad)
            del x64
            scale = max(float(ref.abs().max()), 1e-30)
            err = float((got[ks] - ref).abs().max() / scale)
            rec.note("wgrad", sig, err, tol_wgrad)
            if flops >= FULL_FP32_WGRAD_FLOPS:
                # always on the HOST: an fp32 convolution on the device would go through MIOpen, whose first use of a shape
                # compiles kernels for ~100 s on a fresh box (and is not the independent implementation wanted here)
                full = torch.nn.grad.conv2d_weight(_q64(x, qq).float().cpu().contiguous(), tuple(dw.shape),
                                                   _q64(dy, qq).float().cpu().contiguous(), stride, pad).double()
                rec.note("wgrad_full_fp32", sig, float((got.cpu() - full).abs().max() / max(float(full.abs().max()), 1e-30)),
                         tol_wgrad)
        elif check:
            ref = torch.nn.grad.conv2d_weight(_q64(x, qq), tuple(dw.shape), _q64(dy, qq), stride, pad)
            rec.note("wgrad", sig, _rel(res.to(REF["dev"]).double() - base.to(REF["dev"]).double(), ref), tol_wgrad)
        return res

    def stem_wgrad_s2d(x, dy, dw):
        """The 7x7 stride-2 stem's weight gradient, computed by the product on the space-to-depth image."""
        sig = (tuple(x.shape), tuple(dy.shape), tuple(dw.shape))
        check = ("stem_wgrad",) + sig not in rec.seen
        base = dw.clone() if check else None
        saved = ops.conv_wgrad
        ops.conv_wgrad = orig["conv_wgrad"]            # the inner explicit-output call is part of this op
        try:
            res = orig["stem_wgrad_s2d"](x, dy, dw)
        finally:
            ops.conv_wgrad = saved
        if check:
            got = _c64(res) - _c64(base)
            flops = 2.0 * dy.numel() * 3 * 49
            ks = _k_sample(dw.shape[0]) if big(flops) else list(range(dw.shape[0]))
            if big(flops):
                rec.sampled.add(("stem_wgrad",) + sig)
            ref = torch.nn.grad.conv2d_weight(_c64(x), (len(ks), 3, 7, 7), _c64(dy[:, ks]), 2, (3, 3))
            rec.note("stem_wgrad", sig, float((got[ks] - ref).abs().max() / max(float(ref.abs().max()), 1e-30)), tol_wgrad)
        return res

    def _masked(dz, z, y, mask_scale, mask_shift):
        d = dz.detach().to(REF["dev"]).double()
        if z is not None:
            d = d * (z.detach().to(REF["dev"]) > 0)
        elif mask_scale is not None:
            d = d * ((y.detach().to(REF["dev"]).double() * _V(mask_scale) + _V(mask_shift)) > 0)
        return d

    def bn_apply(y, scale, shift, residual=None, relu=False, res_scale=None, res_shift=None, bf16_only=False):
        out = orig["bn_apply"](y, scale, shift, residual, relu, res_scale, res_shift, bf16_only)
        sig = (tuple(y.shape), residual is not None, relu, ops.is_phantom(residual), ops.is_phantom(out), ops.is_phantom(y))
        if ("bn_apply",) + sig not in rec.seen:
            ns = _img_sample(y.shape[0]) if big_elems(y) else list(range(y.shape[0]))
            if big_elems(y):
                rec.sampled.add(("bn_apply",) + sig)
            ref = _c64(_dat(y)[ns]) * _V(scale) + _V(shift)    # (a bf16-only y / residual: the image is the data)
            if residual is not None:
                ref = ref + _c64(_dat(residual)[ns])
            if relu:
                ref = ref.relu()
            scale_ = max(float(ref.abs().max()), 1e-30)
            if ops.is_phantom(out):       # the output exists only as its bf16 image: the fp64 value up to one bf16 rounding
                rec.note("bn_apply_bf16_only", sig, float((_c64(_dat(out)[ns]) - ref).abs().max() / scale_), 2.0 ** -8)
                rec.note("bn_apply", sig, 0.0, tol)
            else:
                rec.note("bn_apply", sig, _rel(out[ns], ref), tol)
                img = ops.b16_carry(out)
                if img is not None:           # the bf16 image written next to the fp32 output (conv16 reads it)
                    rec.note("bn_apply_image", sig, float((_c64(img[ns]) - ref).abs().max() / scale_), 2.0 ** -8)
        return out

    def bn_bwd_reduce(dz, z, y, mean, invstd, extra=0, mask_scale=None, mask_shift=None):
        out = orig["bn_bwd_reduce"](dz, z, y, mean, invstd, extra, mask_scale, mask_shift)
        sig = (tuple(y.shape), z is not None, mask_scale is not None, ops.is_phantom(z), ops.is_phantom(y))
        z, y = _dat(z), _dat(y)
        if ("bn_bwd_reduce",) + sig not in rec.seen:
            c = y.shape[1]
            s1, s2, a1, a2 = (_zeros64(c) for _ in range(4))
            for i in range(y.shape[0]):                   # whole tensor, one image at a time (memory)
                d = _masked(dz[i:i + 1], None if z is None else z[i:i + 1], y[i:i + 1], mask_scale, mask_shift)
                xh = (_c64(y[i:i + 1]) - _V(mean)) * _V(invstd)
                s1 += d.sum((0, 2, 3)); a1 += d.abs().sum((0, 2, 3))
                d = d * xh
                s2 += d.sum((0, 2, 3)); a2 += d.abs().sum((0, 2, 3))
            o = out.to(REF["dev"])
            # column sums: error relative to the sum of magnitudes (the quantity the rounding scales with)
            e1 = float((o[:c] - s1).abs().max() / max(float(a1.max()), 1e-30))
            e2 = float((o[c:2 * c] - s2).abs().max() / max(float(a2.max()), 1e-30))
            rec.note("bn_bwd_reduce", sig, max(e1, e2), tol)
        return out

    def bn_bwd_apply(dz, z, y, mean, invstd, gamma, sums, count, want_g=False, dgamma=None, dbeta=None, count_dev=None,
                     mask_scale=None, mask_shift=None, g_into=None, bf16_only=False):
        sig = (tuple(y.shape), z is not None, mask_scale is not None, want_g, g_into is not None, bool(bf16_only), ops.is_phantom(z),
               ops.is_phantom(y))
        todo = ("bn_bwd_apply",) + sig not in rec.seen and count_dev is None
        gbase = g_into.clone() if (g_into is not None and todo) else None
        out = orig["bn_bwd_apply"](dz, z, y, mean, invstd, gamma, sums, count, want_g, dgamma, dbeta, count_dev,
                                   mask_scale, mask_shift, g_into, bf16_only)
        z, y = _dat(z), _dat(y)
        if todo:
            ns = _img_sample(y.shape[0]) if big_elems(y) else list(range(y.shape[0]))
            if big_elems(y):
                rec.sampled.add(("bn_bwd_apply",) + sig)
            d = _masked(dz[ns], None if z is None else z[ns], y[ns], mask_scale, mask_shift)
            c = y.shape[1]
            xh = (_c64(y[ns]) - _V(mean)) * _V(invstd)
            ref = _V(gamma) * _V(invstd) * (d - _V(sums[:c]) / count - xh * _V(sums[c:2 * c]) / count)
            scale = max(float((_V(gamma) * _V(invstd) * d).abs().max()), 1e-30)
            phantom = ops.is_phantom(out[0])
            if phantom:       # dx exists only as its bf16 image (conv16): equal to the fp64 value up to one bf16 rounding
                eb = float((_c64(_data(out[0])[ns]) - ref).abs().max() / scale)
                rec.note("bn_bwd_apply_bf16_only", sig, eb, 2.0 ** -8)
                e = 0.0
            else:
                e = float((_c64(out[0][ns]) - ref).abs().max() / scale)
                img = ops.b16_carry(out[0])
                if img is not None:       # the bf16 image written next to dx
                    rec.note("bn_bwd_apply_image", sig, float((_c64(img[ns]) - ref).abs().max() / scale), 2.0 ** -8)
            if want_g:
                e = max(e, _rel(out[1][ns], d if gbase is None else d + _c64(gbase[ns])))
            rec.note("bn_bwd_apply", sig, e, tol)
        return out

    def sum_n(grads, z=None):
        out = orig["sum_n"](grads, z)
        sig = (tuple(grads[0].shape), len(grads), z is not None)
        if ("sum_n",) + sig not in rec.seen:
            g0 = grads[0]
            sub = big_elems(g0) and g0.dim() == 4
            ns = _img_sample(g0.shape[0]) if sub else slice(None)
            if sub:
                rec.sampled.add(("sum_n",) + sig)
            ref = sum(_c64(g[ns]) for g in grads)
            if z is not None:
                ref = ref * (z[ns].detach().to(REF["dev"]) > 0)
            rec.note("sum_n", sig, _rel(out[ns], ref), tol)
        return out

    def upsample_add_fwd(up1, low, bf16_only=False):
        out = orig["upsample_add_fwd"](up1, low, bf16_only)
        sig = (tuple(up1.shape), tuple(low.shape), ops.is_phantom(up1), ops.is_phantom(low), ops.is_phantom(out))
        if ("upsample_add_fwd",) + sig not in rec.seen:
            ns = _img_sample(up1.shape[0]) if big_elems(up1) else list(range(up1.shape[0]))
            u = F.interpolate(_c64(_dat(low)[ns]), scale_factor=2)
            u = F.interpolate(u, size=tuple(up1.shape[2:]), mode="bilinear", align_corners=True)
            ref = _c64(_dat(up1)[ns]) + u
            if ops.is_phantom(out):
                rec.note("upsample_add_bf16_only", sig, float((_c64(_dat(out)[ns]) - ref).abs().max() / max(float(ref.abs().max()), 1e-30)), 2.0 ** -8)
                rec.note("upsample_add_fwd", sig, 0.0, tol)
            else:
                rec.note("upsample_add_fwd", sig, _rel(out[ns], ref), tol)
        return out

    def upsample_add_bwd(dout, low_shape):
        out = orig["upsample_add_bwd"](dout, low_shape)
        sig = (tuple(dout.shape), tuple(low_shape))
        if ("upsample_add_bwd",) + sig not in rec.seen:
            ns = _img_sample(dout.shape[0]) if big_elems(dout) else list(range(dout.shape[0]))
            with torch.enable_grad():                 # we are inside autograd's backward: grad mode is off here
                low = torch.zeros((len(ns),) + tuple(low_shape[1:]), dtype=torch.float64, device=REF["dev"], requires_grad=True)
                u = F.interpolate(F.interpolate(low, scale_factor=2), size=tuple(dout.shape[2:]), mode="bilinear",
                                  align_corners=True)
                u.backward(_c64(dout[ns]))
            rec.note("upsample_add_bwd", sig, _rel(out[ns], low.grad), tol)
        return out

    def bias_relu_bwd(dy, z, dbias):
        base = dbias.clone()
        out = orig["bias_relu_bwd"](dy, z, dbias)
        c = dbias.numel()
        sig = (tuple(dy.shape), z is not None, c)
        if ("bias_relu_bwd",) + sig not in rec.seen:
            tot, mag, e = _zeros64(c), _zeros64(c), 0.0
            chunks = range(dy.shape[0]) if (dy.dim() == 4 and big_elems(dy)) else [slice(None)]
            for i in chunks:                              # whole tensor, one image at a time when it is large
                sl = slice(i, i + 1) if isinstance(i, int) else i
                d = _c64(dy[sl])
                if z is not None:
                    d = d * (z[sl].detach().to(REF["dev"]) > 0)
                    e = max(e, float((_c64(out[sl]) - d).abs().max()))
                flat = d.permute(0, 2, 3, 1).reshape(-1, c) if d.dim() == 4 else d.reshape(-1, c)
                tot += flat.sum(0); mag += flat.abs().sum(0)
            e = e / max(float(dy.abs().max()), 1e-30)
            e = max(e, float(((dbias - base).to(REF["dev"]).double() - tot).abs().max() / max(float(mag.max()), 1e-30)))
            rec.note("bias_relu_bwd", sig, e, tol)
        return out

    def relu_fwd(x):
        out = orig["relu_fwd"](x)
        sig = (tuple(x.shape),)
        if ("relu_fwd",) + sig not in rec.seen:
            ns = _img_sample(x.shape[0]) if (big_elems(x) and x.dim() == 4) else slice(None)
            rec.note("relu_fwd", sig, _rel(out[ns], _c64(x[ns]).relu()), tol)
        return out

    def _finalize_ref(sums64, count, gamma, beta, rm0, rv0, momentum, eps):
        c = gamma.numel()
        m = sums64[:c] / count
        var = (sums64[c:2 * c] / count - m * m).clamp(min=0)
        istd = 1.0 / torch.sqrt(var + eps)
        sc = _c64(gamma) * istd
        unb = var * count / (count - 1.0) if count > 1 else var
        return m, istd, sc, _c64(beta) - m * sc, (1 - momentum) * rm0 + momentum * m, (1 - momentum) * rv0 + momentum * unb

    def _finalize_err(got, ref, rm, rv):
        e = 0.0
        for g, r in zip(tuple(got) + (rm, rv), ref):
            e = max(e, float((_c64(g) - r).abs().max() / max(float(r.abs().max()), 1e-30)))
        return e

    def bn_finalize(sums, count, gamma, beta, running_mean, running_var, momentum, eps, count_dev=None,
                    num_batches_tracked=None):
        sig = (gamma.numel(), float(count), count_dev is not None)
        check = ("bn_finalize",) + sig not in rec.seen and count_dev is None
        if check:
            rm0, rv0, nb0 = _c64(running_mean), _c64(running_var), None if num_batches_tracked is None else int(num_batches_tracked)
        got = orig["bn_finalize"](sums, count, gamma, beta, running_mean, running_var, momentum, eps, count_dev,
                                  num_batches_tracked)
        if check:
            ref = _finalize_ref(sums.detach().to(REF["dev"]).double(), count, gamma, beta, rm0, rv0, momentum, eps)
            e = _finalize_err(got, ref, running_mean, running_var)
            if nb0 is not None and int(num_batches_tracked) != nb0 + 1:
                e = float("inf")
            rec.note("bn_finalize", sig, e, 1e-5)        # fp32 outputs of fp64 arithmetic: a few ulp
        return got

    def bn_stats_finalize(slab, count, gamma, beta, running_mean, running_var, momentum, eps, num_batches_tracked=None):
        c = gamma.numel()
        sig = (c, float(count), slab.numel() // (2 * c))
        check = ("bn_stats_finalize",) + sig not in rec.seen
        if check:
            rm0, rv0 = _c64(running_mean), _c64(running_var)
        got = orig["bn_stats_finalize"](slab, count, gamma, beta, running_mean, running_var, momentum, eps,
                                        num_batches_tracked)
        if check:
            sums = slab.detach().to(REF["dev"]).view(-1, 2 * c).sum(0)
            ref = _finalize_ref(sums, count, gamma, beta, rm0, rv0, momentum, eps)
            rec.note("bn_stats_finalize", sig, _finalize_err(got, ref, running_mean, running_var), 1e-5)
        return got

    # ---- DCNv2 (config 4's heads; ext/dcn/dcn_v2.py:16-52, dcn_v2_im2col_cuda.cu:125-327) ------------------------------------
    # Reference = oracle/dcn.py in fp64 on the reference device.  bf16 calls: the kernels' contract (both operands of the
    # three matrix products rounded to bf16 — oracle/dcn.py:_ContractBf16 — everything around them in fp32).  The op is
    # local for bounded offsets: forward / data gradient are recomputed on windows (top-left corner, an interior window
    # straddling the kernels' 8 x 16 pixel blocks, bottom-right corner; three different images) cropped with a margin of
    # pad + ceil(max |offset|) + 2 pixels, which reproduces out, d input, d offset and d mask of the window's core exactly;
    # the weight gradient reduces over ALL pixels and is recomputed over the full range for a sample of filters.
    # bf16 calls, forward and weight gradient: the kernel blends a sample in fp32 and THEN rounds it to bf16; the reference blends in
    # fp64.  A sample within fp32 evaluation error of a bf16 rounding boundary may legitimately round the other way (about one in
    # 2e4; measured without this: 3e-4 of the output scale at the worst of 8 M outputs).  The reference therefore carries, per
    # output element, the exact slack those samples allow — sum over the near-boundary samples of one bf16 ulp of the sample times
    # |weight| (|dY| for the weight gradient) — and the call must be within tol * scale + slack: flip-free elements (the rest) are
    # held to the same 2e-5 as every convolution.
    def _flip_slack(cols, mag):
        """cols: fp64 samples; mag >= |cols|: the magnitude of the blend's terms (its fp32 evaluation error is <= 4e-7 * mag).
        -> one bf16 ulp where the fp32-evaluated sample may round to the other neighbour, 0 elsewhere."""
        a = cols.abs()
        ulp = torch.exp2(torch.floor(torch.log2(a.clamp_min(1e-300))) - 7.0)
        frac = torch.remainder(a / ulp, 1.0)
        near = (frac - 0.5).abs() * ulp <= 4e-7 * mag + 1e-300
        return torch.where(near & (a > 0), ulp, torch.zeros_like(ulp))

    def _dcn_windows(n, h, wd):
        if h <= 48 and wd <= 64:
            return [(i, 0, h, 0, wd) for i in range(n)]
        r_mid, c_mid = (h // 2 // 8) * 8 - 3, (wd // 2 // 16) * 16 - 8
        return [(0, 0, min(20, h), 0, min(28, wd)), (n // 2, r_mid, min(r_mid + 24, h), c_mid, min(c_mid + 36, wd)),
                (n - 1, max(h - 20, 0), h, max(wd - 28, 0), wd)]

    def _dcn_window_ref(x, off, mask, w, bias, dy, win, cfg, quant, local=True):
        """fp64 reference on one window -> (out, dx, doffset, dmask) of the window's core (the gradients None without dy).
        local=False (out and in sizes differ / strided): the window is a whole image, nothing is cropped."""
        from oracle import dcn as odcn
        stride, pad, dil, dg = cfg
        n0, r0, r1, c0, c1 = win
        h, wd = x.shape[2], x.shape[3]
        if local:
            m = pad[0] + int(float(off.detach().abs().max()) + 0.999) + 2
            ra, rb, ca, cb = max(0, r0 - m), min(h, r1 + m), max(0, c0 - m), min(wd, c1 + m)
            cut = lambda t: t[n0:n0 + 1, :, ra:rb, ca:cb]
            core = (slice(None), slice(None), slice(r0 - ra, r1 - ra), slice(c0 - ca, c1 - ca))
        else:
            cut = lambda t: t[n0:n0 + 1]
            core = (slice(None),) * 4
            ra = ca = 0
        with torch.enable_grad():
            crop = [_c64(cut(t)).contiguous().requires_grad_(dy is not None) for t in (x, off, mask)]
            # (pos_fp32: sample positions in one float32 addition, as the reference's kernel and ours evaluate them)
            ref = odcn.dcn_v2_conv(crop[0], crop[1], crop[2], _c64(w).contiguous(), None if bias is None else _c64(bias),
                                   stride, pad, dil, dg, bf16=bool(quant), pos_fp32=True, origin=(ra, ca))
            if dy is not None:
                ref.backward(_c64(cut(dy)).contiguous())
        return (ref.detach()[core],) + tuple(None if dy is None else t.grad[core] for t in crop)

    def _dcn_fwd_window(x, off, mask, w, bias, win, cfg, quant, local=True):
        """fp64 forward on one window -> (out, slack) of the window's core; slack: see _flip_slack (zeros for fp32 calls)."""
        from oracle import dcn as odcn
        stride, pad, dil, dg = cfg
        n0, r0, r1, c0, c1 = win
        h, wd = x.shape[2], x.shape[3]
        k, c, kh, kw = w.shape
        if local:
            m = pad[0] + int(float(off.detach().abs().max()) + 0.999) + 2
            ra, rb, ca, cb = max(0, r0 - m), min(h, r1 + m), max(0, c0 - m), min(wd, c1 + m)
            cut = lambda t: t[n0:n0 + 1, :, ra:rb, ca:cb]
            core = (slice(None), slice(None), slice(r0 - ra, r1 - ra), slice(c0 - ca, c1 - ca))
        else:
            cut = lambda t: t[n0:n0 + 1]
            core = (slice(None),) * 4
            ra = ca = 0
        xs, os_, ms = (_c64(cut(t)).contiguous() for t in (x, off, mask))
        cols = odcn.dcn_columns(xs, os_, ms, kh, kw, stride, pad, dil, dg, pos_fp32=True, origin=(ra, ca))
        w3 = _c64(w).reshape(k, c, kh * kw)
        slack = None
        if quant:
            sl = _flip_slack(cols, odcn.dcn_columns(xs.abs(), os_, ms.abs(), kh, kw, stride, pad, dil, dg, pos_fp32=True, origin=(ra, ca)))
            cols, w3 = cols.to(torch.bfloat16).double(), w3.to(torch.bfloat16).double()
            slack = torch.einsum('nctpq,kct->nkpq', sl, w3.abs())[core]
        ref = torch.einsum('nctpq,kct->nkpq', cols, w3)
        if bias is not None:
            ref = ref + _c64(bias).view(1, -1, 1, 1)
        ref = ref[core]
        return ref, (slack if slack is not None else torch.zeros_like(ref))

    def _dcn_local(x, w, off, stride, pad, dil):
        """The window decomposition holds for same-size stride-1 layers (the window kernels' domain; the heads' 3x3)."""
        return stride == 1 and off.shape[2] == x.shape[2] and off.shape[3] == x.shape[3]

    def dcn_fwd(x, offset, mask, w, bias, stride, pad, dilation, dg, bf16=False):
        y = orig["dcn_fwd"](x, offset, mask, w, bias, stride, pad, dilation, dg, bf16)
        sig = (tuple(x.shape), tuple(w.shape), stride, tuple(pad), dilation, dg, bool(bf16))
        if ("dcn_fwd",) + sig not in rec.seen:
            n, _, h, wd = x.shape
            cfg = (stride, tuple(pad), dilation, dg)
            local = _dcn_local(x, w, offset, stride, pad, dilation)
            wins = _dcn_windows(n, h, wd) if local else [(i, 0, y.shape[2], 0, y.shape[3]) for i in range(n)]
            if len(wins) < n:
                rec.sampled.add(("dcn_fwd",) + sig)
            scale = max(float(y.abs().max()), 1e-30)
            err = 0.0
            for win in wins:
                n0, r0, r1, c0, c1 = win
                ref, slack = _dcn_fwd_window(x, offset, mask, w, bias, win, cfg, bf16, local)
                d = (_c64(y[n0:n0 + 1, :, r0:r1, c0:c1]) - ref).abs() - slack
                err = max(err, float(d.max()) / scale)
            rec.note("dcn_fwd", sig, max(err, 0.0), tol_)
        return y

    def dcn_dgrad(x, offset, mask, w, dy, stride, pad, dilation, dg, bf16=False, out=None):
        img = ops.b16_carry(dy) if bf16 else None
        sig = (tuple(x.shape), tuple(w.shape), stride, tuple(pad), dilation, dg, bool(bf16), out is not None, img is not None)
        todo = ("dcn_dgrad",) + sig not in rec.seen
        base = out.clone() if (out is not None and todo) else None
        dx, doff, dmask = orig["dcn_dgrad"](x, offset, mask, w, dy, stride, pad, dilation, dg, bf16, out)
        if todo:
            n, _, h, wd = x.shape
            cfg = (stride, tuple(pad), dilation, dg)
            if img is not None:
                # image-fed dY: the kernel read the producer's bf16 image instead of rounding dy itself — it must BE round(dy)
                ns = _img_sample(n)
                same = torch.equal(img[ns].float(), dy[ns].to(torch.bfloat16).float())
                rec.note("dcn_dy_image", sig, 0.0 if same else 1.0, 0.5)
            local = _dcn_local(x, w, offset, stride, pad, dilation)
            wins = _dcn_windows(n, h, wd) if local else [(i, 0, h, 0, wd) for i in range(n)]
            if len(wins) < n:
                rec.sampled.add(("dcn_dgrad",) + sig)
            got = dx if base is None else (dx.double() - base.double())
            scales = [max(float(t.abs().max()), 1e-30) for t in (got, doff, dmask)]
            errs = [0.0, 0.0, 0.0]
            for win in wins:
                n0, r0, r1, c0, c1 = win
                ref = _dcn_window_ref(x, offset, mask, w, None, dy, win, cfg, bf16, local)
                o = (slice(n0, n0 + 1), slice(None), slice(r0, r1), slice(c0, c1)) if local else (slice(n0, n0 + 1),)
                errs[0] = max(errs[0], float((_c64(got[o]) - ref[1]).abs().max()) / scales[0])
                errs[1] = max(errs[1], float((_c64(doff[o]) - ref[2]).abs().max()) / scales[1])
                errs[2] = max(errs[2], float((_c64(dmask[o]) - ref[3]).abs().max()) / scales[2])
            t = tol_
            rec.note("dcn_dgrad", sig, errs[0], t)
            rec.note("dcn_dgrad_doffset", sig, errs[1], t)
            rec.note("dcn_dgrad_dmask", sig, errs[2], t)
        return dx, doff, dmask

    def dcn_wgrad(x, offset, mask, dy, dw, stride, pad, dilation, dg, bf16=False, dy_img=None):
        sig = (tuple(x.shape), tuple(dw.shape), stride, tuple(pad), dilation, dg, bool(bf16), dy_img is not None)
        todo = ("dcn_wgrad",) + sig not in rec.seen
        base = dw.clone() if todo else None
        res = orig["dcn_wgrad"](x, offset, mask, dy, dw, stride, pad, dilation, dg, bf16, dy_img)
        if todo:
            from oracle import dcn as odcn
            n = x.shape[0]
            k, c, r, s_ = dw.shape
            if dy_img is not None:
                ns = _img_sample(n)
                same = torch.equal(dy_img[ns].float(), dy[ns].to(torch.bfloat16).float())
                rec.note("dcn_dy_image", sig, 0.0 if same else 1.0, 0.5)
            flops = 2.0 * dy.numel() * c * r * s_
            ks = _k_sample(k) if big(flops) else list(range(k))
            if big(flops):
                rec.sampled.add(("dcn_wgrad",) + sig)
            ref = torch.zeros((len(ks), c, r * s_), dtype=torch.float64, device=REF["dev"])
            slack = torch.zeros_like(ref)
            for i in range(n):                                    # whole pixel range, one image at a time (memory)
                xi, oi, mi = _c64(x[i:i + 1]), _c64(offset[i:i + 1]), _c64(mask[i:i + 1])
                cols = odcn.dcn_columns(xi, oi, mi, r, s_, stride, tuple(pad), dilation, dg, pos_fp32=True)
                dyi = _c64(dy[i:i + 1, ks])
                if bf16:
                    sl = _flip_slack(cols, odcn.dcn_columns(xi.abs(), oi, mi.abs(), r, s_, stride, tuple(pad), dilation, dg, pos_fp32=True))
                    cols, dyi = cols.to(torch.bfloat16).double(), dyi.to(torch.bfloat16).double()
                    slack += torch.einsum('nkpq,nctpq->kct', dyi.abs(), sl)
                    del sl
                ref += torch.einsum('nkpq,nctpq->kct', dyi, cols)
                del cols
            got = (_c64(res) - _c64(base))[ks].reshape(len(ks), c, r * s_)
            d = (got - ref).abs() - slack
            rec.note("dcn_wgrad", sig, max(float(d.max()), 0.0) / max(float(ref.abs().max()), 1e-30), tol_wgrad)
        return res

    patched = dict(conv_fprop=conv_fprop, conv_dgrad=conv_dgrad, conv_wgrad=conv_wgrad, stem_wgrad_s2d=stem_wgrad_s2d,
                   conv_fprop_packed=conv_fprop_packed, conv_wgrad_packed=conv_wgrad_packed,
                   bn_finalize=bn_finalize, bn_stats_finalize=bn_stats_finalize, bn_apply=bn_apply,
                   bn_bwd_reduce=bn_bwd_reduce, bn_bwd_apply=bn_bwd_apply, sum_n=sum_n, upsample_add_fwd=upsample_add_fwd,
                   upsample_add_bwd=upsample_add_bwd, bias_relu_bwd=bias_relu_bwd, relu_fwd=relu_fwd,
                   dcn_fwd=dcn_fwd, dcn_dgrad=dcn_dgrad, dcn_wgrad=dcn_wgrad)
    for n, f in patched.items():
        setattr(ops, n, f)
    try:
        yield rec
    finally:
        REF["dev"] = "cpu"
        for n, f in orig.items():
            setattr(ops, n, f)
