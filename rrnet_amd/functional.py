"""Autograd plumbing over the HIP kernels (rrnet_amd/ops.py -> include/rrnet_hip.h).

PyTorch is used for the graph bookkeeping only: every node below runs hand-written gfx950
kernels forward and backward.  Activations are NHWC in memory (logical NCHW tensors with
channels_last strides); convolution weights OHWI.

Parameter gradients: when a parameter carries a pre-allocated gradient view (`_rr_grad`, set by
rrnet_amd.flat.FlatParams — a slice of the one flat fp32 gradient buffer that the RCCL
all-reduce and the fused Adam kernel operate on) the backward kernels accumulate straight into
it (wgrad's float atomics, the BN apply kernel's dgamma/dbeta) and autograd sees `None`;
otherwise a fresh gradient tensor is returned the usual way.
"""
import os

import torch
import torch.distributed as dist

from rrnet_amd import dptrace, ops


def _grad_target(p):
    return getattr(p, "_rr_grad", None)


def _mark(*params):
    """Report finished parameter gradients to the flat buffer's data-parallel bucketing (rrnet_amd.flat)."""
    for p in params:
        flat = getattr(p, "_rr_flat", None)
        if flat is not None and p is not None:
            flat.mark_ready(p)


# Weight gradients on a second HIP stream.  A layer's wgrad is off the critical path of backward (nothing waits for it
# before the optimizer step / the gradient exchange), so it is enqueued on a side stream behind the kernels that produce
# its operands and runs concurrently with whatever the main stream does next.  Both streams are mostly MFMA-bound, so
# this is not "more FLOPs at once"; what it buys is filled tails (a 256x256 layer is 10.67 rounds of resident workgroups:
# the last third of a round idles 3 % of the launch) and HBM-bound stretches (BatchNorm backward, fan-in sums) that no
# longer leave the matrix cores idle.  Measured at B=8, 1024x1024 on one box, back to back: 455.4 ms per step against 459.6 ms
# (step_mfma_frac 0.784 / 0.777); the dominant kernel's own launches lengthen by 1.7 % (0.868 -> 0.853 of the MFMA peak)
# because some of them now share the chip with a wgrad.  RR_WGRAD_STREAM: 2 (default) free-running side stream, joined
# at the end of backward, before a gradient bucket's all-reduce and before the optimizer step; 1: additionally joined
# before every data gradient, so that wgrad only ever overlaps the HBM-bound stretch (measured SLOWER than no side
# stream at all: 470.0 vs 463.0 ms — two stream hand-overs per layer); 0: everything on one stream.
_WGRAD_STREAM = os.environ.get("RR_WGRAD_STREAM", "2") in ("1", "2")
_WGRAD_FREE = os.environ.get("RR_WGRAD_STREAM", "2") == "2"
_WG_STATE = {"pending": False, "task": None}      # task: the autograd graph task whose end-of-backward join is queued
# RR_WGRAD_STRESS=<cycles> (test switch, tests/test_streams_gpu.py): a spin kernel of that many clock cycles is put in front
# of every side-stream launch, so the side stream falls further and further behind the main one; an operand that is not
# kept alive (record_stream) or not ordered (event) for the side stream then shows up as a wrong gradient instead of
# passing because the two streams happened to run in step.  "1" = 2 M cycles (~1 ms) per launch.
_STRESS_CYCLES = int(os.environ.get("RR_WGRAD_STRESS", "0") or 0)
if _STRESS_CYCLES == 1:
    _STRESS_CYCLES = 2_000_000


def _stress_delay():
    if _STRESS_CYCLES > 0:
        torch.cuda._sleep(_STRESS_CYCLES)


def _wgrad_join(device):
    """Main stream waits for the weight gradients in flight on the side stream (before the next MFMA-bound kernel)."""
    if _WG_STATE["pending"] and not _WGRAD_FREE:
        torch.cuda.current_stream(device).wait_stream(_side_stream(device, "wgrad"))
        _WG_STATE["pending"] = False


def _wgrad_async(fn, device, *tensors):
    """Run fn() (a weight-gradient launch + its mark_ready reports) on the side stream, ordered behind everything the
    current stream holds so far; `tensors` are kept from being recycled by the allocator until the side stream passes."""
    if not (_WGRAD_STREAM and device.type == "cuda"):
        return fn()
    cur = torch.cuda.current_stream(device)
    side = _side_stream(device, "wgrad")
    side.wait_stream(cur)
    for t in tensors:                     # maxima reduced / bf16 images written on the current stream so far are complete for the side stream too
        ops.amax_publish(t)
        ops.b16_publish(t)
    with torch.cuda.stream(side):
        _stress_delay()
        fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
            img = ops.b16_carry(t)
            if img is not None:
                img.record_stream(side)
    _WG_STATE["pending"] = True
    task = torch._C._current_graph_task_id()       # one join per backward pass (a pass that raised leaves no stale flag)
    if task < 0:                                   # not inside a backward pass (a backward called by hand): join right away
        torch.cuda.current_stream(device).wait_stream(side)
        _WG_STATE["pending"] = False
    elif _WG_STATE["task"] != task:
        _WG_STATE["task"] = task

        def _end_of_backward():
            torch.cuda.current_stream(device).wait_stream(side)
            _WG_STATE["pending"] = False
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)


_SYNC_COALESCE = True   # joint SyncBN exchange of layers that share their input
_G_INTO = True     # residual gradient added into the fan-in buffer by bn_bwd_apply itself
BN_FUSED_STATS = True    # single process: slab -> statistics -> coefficients in one launch


def _is_sync(bn):
    return isinstance(bn, torch.nn.SyncBatchNorm) and dptrace.dp_active()


class _ConvBnAct(torch.autograd.Function):
    """z = relu?( BN(conv(x, w)) [+ residual] ), training statistics fused into the conv epilogue.
    Reference: conv->bn->relu->(+skip) sequences of backbones/hourglass.py:31-40,56-61,
    backbones/resnet.py:33-53."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, residual, bn, stride, pad, relu, x_acc=None, res_acc=None, in_link=None,
                out_link=None):
        ctx.accs = (x_acc, res_acc)
        ctx.links = (in_link, out_link)
        ctx.bf16 = ops.BF16
        x = ops.to_nhwc(x, keep_phantom=True)          # (a bf16-only input: the conv16 kernels read its image)
        wc = ops.to_nhwc(w)
        _wamax_attach(w, wc)
        n, _, h, wd = x.shape
        k = w.shape[0]
        sync = _is_sync(bn)
        ctx.packed = False
        if bn.training:
            # (packability is decided from the node's own needs_input_grad: inside Function.forward grad mode is off and a
            # layout-converted copy of an input that wants a gradient would look like a constant)
            if not ctx.needs_input_grad[0] and ops.conv_packable(x, wc, stride):
                # very few input channels (the 7x7 stride-2 stem): taps packed per output pixel, 1x1 GEMM on the vector
                # kernels; the packed image replaces x as the tensor saved for the weight gradient
                y, slab, x = ops.conv_fprop_packed(x, wc, stride, pad, want_stats=True)
                ctx.packed = True
            else:
                # (inside the backbone, where the shape qualifies, the pre-BN output too is written as a bf16 image only: the
                # statistics come out of the fp32 accumulators, BatchNorm apply / backward read the image)
                y, slab = ops.conv_fprop(x, wc, None, stride, pad, False, want_stats=True, w16=_w16_of(w)[0],
                                         y_bf16_only=ops.phantom_y_ok(k, x, wc, stride, pad))
            count = float(y.numel() // k)
            cnt_dev = None
            mom = bn.momentum if bn.momentum is not None else 0.1
            # (layers of more than 512 pixel tiles keep the two-kernel form: the fused kernel runs C/32 workgroups only)
            if sync or not BN_FUSED_STATS or slab.numel() > 512 * 2 * k:   # SyncBN exchange: one small all-reduce of [sum, sumsq, count] (C5 in SURVEY §2.2);
                if sync:
                    # SyncBN exchange: one small all-reduce of [sum, sumsq, count] (C5 in SURVEY §2.2).  The local count is stored into
                    # the buffer by the slab reduction itself and read back on the device by the finalize kernel, which also leaves it
                    # in storage of its own for the backward: two launches around the collective (round 5: reduce, fill_, clone,
                    # finalize — and a `sums[2 * k] = count` item assignment before that, a pageable H2D copy that blocked the host
                    # until the stream had drained: 430 ms of host time per 487 ms step)
                    sums = ops.bn_reduce_slab(slab, k, extra=1, count=count)
                    dptrace.all_reduce(sums, None, "default", "syncbn_fwd")
                    mean, invstd, scale, shift, cnt_dev = ops.bn_finalize_sync(sums, sums[2 * k:], gamma, beta, bn.running_mean,
                                                                               bn.running_var, mom, bn.eps, bn.num_batches_tracked)
                else:
                    sums = ops.bn_reduce_slab(slab, k, extra=1)
                    mean, invstd, scale, shift = ops.bn_finalize(sums, count, gamma, beta, bn.running_mean, bn.running_var,
                                                                 mom, bn.eps, None, bn.num_batches_tracked)
            else:      # nothing to exchange: slab -> statistics -> coefficients in one launch
                mean, invstd, scale, shift = ops.bn_stats_finalize(slab, count, gamma, beta, bn.running_mean,
                                                                   bn.running_var, mom, bn.eps, bn.num_batches_tracked)
        else:
            scale, shift = ops.bn_eval_coeffs(gamma, beta, bn.running_mean, bn.running_var, bn.eps)
            mean = invstd = cnt_dev = None
            count = 0.0
            if residual is None:
                # inference: BN is a per-channel affine map -> folded into the weights and the conv epilogue's
                # bias (+ReLU); no separate pass over the output
                ctx.cfg = (stride, pad, relu, count, sync, False)
                return ops.conv_fprop(x, ops.to_nhwc(wc * scale.view(-1, 1, 1, 1)), shift, stride, pad, relu)
            y = ops.conv_fprop(x, wc, None, stride, pad, False, w16=_w16_of(w)[0])
        res = ops.to_nhwc(residual, keep_phantom=True) if residual is not None else None
        # inside the backbone (ops.phantom_scope) the activation is written as a bf16 image only where the shape qualifies: every
        # consumer there reads the image (conv16 operands, the next block's residual, the ReLU mask of this layer's backward)
        z = ops.bn_apply(y, scale, shift, res, relu,
                         bf16_only=bn.training and ops.phantom_out_ok(k, y.shape[0] * y.shape[2] * y.shape[3], y.device))
        if bn.training:
            # ReLU mask: layers with a residual need their output z; the others recompute it from y (one tensor
            # read less in each of the two backward passes)
            remask = relu and residual is None
            ctx.save_for_backward(x, wc, y, z if (relu and not remask) else None, mean, invstd, gamma, cnt_dev,
                                  scale if remask else None, shift if remask else None)
            ctx.x_amax = ops.amax_carry(x)      # (split-operand kernels: the weight gradient reuses the forward's reduction)
            ctx.x_b16 = ops.b16_carry(x)        # (conv16 kernels: the weight gradient reads the forward's bf16 image of x)
            ctx.z_b16 = ops.b16_carry(z) if (relu and not remask) else None      # (a bf16-only z: its image is the mask's source)
            ctx.y_b16 = ops.b16_carry(y) if ops.is_phantom(y) else None
            if out_link is not None:
                # what a consumer's data gradient needs to produce this layer's BatchNorm-backward sums in its epilogue
                out_link.y, out_link.mean, out_link.invstd = y, mean, invstd
                out_link.use_z = bool(relu and not remask)      # mask from this layer's output = the consumer's input
                out_link.msc, out_link.msh = (scale, shift) if remask else (None, None)
        ctx.cfg = (stride, pad, relu, count, sync, residual is not None)
        ctx.params = (w, gamma, beta)
        ctx.xshape = None if ctx.packed else tuple(x.shape)
        return z

    @staticmethod
    def backward(ctx, dz):
        with ops.bf16_scope(ctx.bf16):      # the precision this node's forward ran at
            return _ConvBnAct._backward(ctx, dz)

    @staticmethod
    def _backward(ctx, dz):
        x, wc, y, z, mean, invstd, gamma, cnt_dev, msc, msh = ctx.saved_tensors
        ops.amax_restore(x, getattr(ctx, "x_amax", None))
        ops.b16_restore(x, getattr(ctx, "x_b16", None))
        ops.b16_restore(z, getattr(ctx, "z_b16", None))
        ops.b16_restore(y, getattr(ctx, "y_b16", None))
        stride, pad, relu, count, sync, has_res = ctx.cfg
        w, gamma_p, beta_p = ctx.params
        dz = ops.to_nhwc(dz)
        k = y.shape[1]
        in_link, out_link = ctx.links
        sums = None
        if out_link is not None and out_link.sums is not None:
            # the data gradient that produced dz already reduced it (rr_conv_dgrad_s1_bnsum) — valid only if what
            # arrives here is that very tensor (no other consumer's gradient was added on the way)
            if out_link.dz is not None and out_link.dz.data_ptr() == dz.data_ptr() and out_link.dz.shape == dz.shape:
                sums = out_link.sums
            out_link.sums = out_link.dz = None
        if sums is None:
            sums = ops.bn_bwd_reduce(dz, z, y, mean, invstd, mask_scale=msc, mask_shift=msh)
        dg_t, db_t = _grad_target(gamma_p), _grad_target(beta_p)
        ret_dg = ret_db = None
        fused_affine = dg_t is not None and db_t is not None and not sync
        if not fused_affine:
            # gradients of gamma/beta come from the LOCAL sums (SyncBN averages them later like any grad)
            if dg_t is not None and db_t is not None and dg_t.is_cuda:
                ops.bn_affine_grad(sums, dg_t, db_t)            # one launch, before the sums are exchanged
            else:
                db = sums[:k].float()
                dg = sums[k:2 * k].float()
                if dg_t is not None:
                    dg_t.add_(dg)
                    db_t.add_(db)
                else:
                    ret_dg, ret_db = dg, db
        if sync:
            dptrace.all_reduce(sums, None, "default", "syncbn_bwd")
        want_g = has_res and relu
        x_acc, res_acc = ctx.accs
        # the residual's fan-in buffer already holds another consumer's gradient: add the masked gradient into it
        # inside this kernel (no separate g tensor, no add pass)
        g_into = res_acc.buf if (want_g and res_acc is not None and res_acc.buf is not None and ctx.needs_input_grad[4]
                                 and _G_INTO) else None
        if g_into is not None:
            res_acc.begin(dz.device)
        # conv16: when the data gradient (if one is wanted) and the weight gradient of this convolution both read dy's bf16
        # image, the fp32 dy is never written
        w_t0 = _grad_target(w)
        rb_link = (in_link is not None and in_link.relu_bias) or (ctx.accs[0] is not None and ctx.accs[0].link is not None
                                                                    and ctx.accs[0].link.relu_bias)
        only16 = (not ctx.packed and ctx.xshape is not None and w_t0 is not None and not rb_link
                  and ops.wgrad16_takes(ctx.xshape, tuple(y.shape), tuple(w.shape), stride)
                  and (not ctx.needs_input_grad[0] or ops.dgrad16_takes(tuple(y.shape), tuple(w.shape), ctx.xshape, stride, pad)))
        dy, g = ops.bn_bwd_apply(dz, z, y, mean, invstd, gamma, sums, count, want_g,
                                 dg_t if fused_affine else None, db_t if fused_affine else None, cnt_dev, msc, msh,
                                 g_into=g_into, bf16_only=only16)
        if g_into is not None:
            res_acc.end(dz.device)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dy, wc, ctx.xshape, stride, pad, x_acc, in_link, x, w)
        w_t = _grad_target(w)
        ret_dw = None
        stem = (not ctx.packed and tuple(w.shape[1:]) == (3, 7, 7) and stride == 2 and tuple(pad) == (3, 3)
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)
        if ctx.packed:                 # x is the packed tap image of the forward
            wg = lambda tgt: ops.conv_wgrad_packed(x, dy, tgt)
        elif stem:
            wg = lambda tgt: ops.stem_wgrad_s2d(x, dy, tgt)
        else:
            wg = lambda tgt: ops.conv_wgrad(x, dy, tgt, stride, pad)
        if w_t is not None:
            _wgrad_async(lambda: (wg(w_t), _mark(w)), x.device, x, dy)
        else:
            dw = ops.zeros_nhwc(*w.shape, device=x.device)
            wg(dw)
            ret_dw = dw
        if dg_t is not None and db_t is not None:
            _mark(gamma_p, beta_p)
        dres = None
        if has_res and ctx.needs_input_grad[4]:
            dres = g if relu else dz
            if res_acc is not None and relu:           # g is a fresh tensor of ours: it may serve as the fan-in target
                res_acc.pending -= 1
                if res_acc.buf is None:
                    res_acc.buf = g
                    res_acc.end(dz.device)
                else:
                    if g_into is None:
                        res_acc.begin(dz.device)
                        res_acc.buf.add_(g)
                        res_acc.end(dz.device)
                    dres = None
        return dx, ret_dw, ret_dg, ret_db, dres, None, None, None, None, None, None, None, None


def _wt_of(w_param):
    """The cached flipped / transposed copy of a filter parameter that lives in a FlatParams buffer (None otherwise)."""
    flat = getattr(w_param, "_rr_flat", None) if w_param is not None else None
    return flat.wt_view(w_param) if flat is not None else None


def _w16_of(w_param):
    """bf16 mode: (bf16 copy, flipped bf16 copy) of a filter parameter of a FlatParams buffer, else (None, None)."""
    flat = getattr(w_param, "_rr_flat", None) if (w_param is not None and ops.BF16 == ops.MATH_BF16) else None
    return flat.w16_views(w_param) if flat is not None else (None, None)


def _wamax_attach(w_param, wc):
    """Split-operand kernels: hand the filter's maximum (one small reduction per parameter and optimizer step, remembered on
    the parameter) to the NHWC view `wc` the launch receives — forward and data gradient see different view objects."""
    if ops.BF16 != ops.MATH_F16X3 or w_param is None or not wc.is_cuda:
        return
    flat = getattr(w_param, "_rr_flat", None)
    # (flat.flat._version: FlatParams.broadcast / any in-place write of the flat buffer; epoch: optimizer steps and
    # FlatParams.invalidate_wt(), the documented call after a write through `p.data`)
    key = (flat.epoch if flat is not None else -1, flat.flat._version if flat is not None else -1, w_param._version,
           w_param.data_ptr())
    hit = getattr(w_param, "_rr_wamax", None)
    if hit is None or hit[0] != key:
        hit = (key, ops.amax_of(wc))
        w_param._rr_wamax = hit
    ops.amax_restore(wc, hit[1])


def _input_grad(dy, wc, xshape, stride, pad, x_acc, in_link, x, w_param=None):
    """Data gradient of a convolution node -> the tensor to hand to autograd (None when it was added into the fan-in
    buffer of x's fan-out).  Where the launch can carry them it also produces the BatchNorm-backward sums of the layer
    that produced x (ops.BnLink): when x has this node as its only consumer, or when this node is the LAST registered
    contributor to the fan-in buffer of x's fan-out (the epilogue then holds the complete gradient)."""
    _wgrad_join(dy.device)        # the previous layer's weight gradient has had the HBM-bound stretch to itself
    _wamax_attach(w_param, wc)
    wt = _wt_of(w_param) if stride == 1 else None
    wt16 = _w16_of(w_param)[1] if (stride == 1 and wt is not None) else None
    if x_acc is None:
        link = in_link if (in_link is not None and in_link.consumers == 1) else None
        return ops.conv_dgrad(dy, wc, xshape, stride, pad, bnsum=link, bnsum_z=x, wt=wt, wt16=wt16)
    link = x_acc.link if x_acc.pending == 1 else None
    x_acc.pending -= 1
    if x_acc.buf is not None:
        # another consumer of x already produced its gradient: add into it inside the dgrad epilogue
        x_acc.begin(dy.device)
        ops.conv_dgrad(dy, wc, xshape, stride, pad, out=x_acc.buf, accumulate=True, bnsum=link, bnsum_z=x, wt=wt, wt16=wt16)
        x_acc.end(dy.device)
        return None
    x_acc.buf = ops.conv_dgrad(dy, wc, xshape, stride, pad, bnsum=link, bnsum_z=x, wt=wt, wt16=wt16)
    x_acc.end(dy.device)
    return x_acc.buf


def conv_bn_act(x, conv, bn, relu=True, residual=None, x_acc=None, res_acc=None):
    """conv: nn.Conv2d (bias-free), bn: nn.BatchNorm2d / nn.SyncBatchNorm used as parameter holders.
    x_acc / res_acc: the GradAcc of the fan-out `x` / `residual` came from (see fanout_shared)."""
    if x_acc is None:
        x_acc = getattr(x, "_rr_acc", None)
    if res_acc is None and residual is not None:
        res_acc = getattr(residual, "_rr_acc", None)     # set only on raw fan-out views (identity skip)
    grad_on = torch.is_grad_enabled()
    in_link = getattr(x, "_rr_bnlink", None) if x_acc is None else None
    if grad_on and x.requires_grad:
        # registered contributors of a fan-in buffer: the last one to run its backward may reduce the complete gradient
        if x_acc is not None:
            x_acc.pending += 1
        elif in_link is not None:
            in_link.consumers += 1
    if grad_on and res_acc is not None and relu and residual.requires_grad:
        res_acc.pending += 1
    out_link = ops.BnLink() if (grad_on and bn.training) else None
    out = _ConvBnAct.apply(x, conv.weight, bn.weight, bn.bias, residual, bn, conv.stride[0], tuple(conv.padding), relu,
                           x_acc, res_acc, in_link, out_link)
    if out_link is not None and out_link.y is not None:
        out._rr_bnlink = out_link
    return out


class _ConvBnSyncMulti(torch.autograd.Function):
    """SyncBN, several conv -> bn [-> relu] layers on the SAME input (a projection block's skip and conv1,
    backbones/hourglass.py:31-40; the first layers of an hourglass module's up1 and low1 branches, :96-101): one autograd
    node so that their statistic exchanges share ONE all-reduce forward ([sum, sumsq] of every layer + one sample count
    per layer: the strides may differ) and ONE backward ([sum dz, sum dz*xhat] of every layer) instead of one each — the
    layers are independent given x, so nothing is re-ordered.  Used only when world_size > 1; a single process keeps the
    per-layer nodes (and their fused single-launch statistics)."""

    @staticmethod
    def forward(ctx, x, x_acc, specs, *params):
        # specs: [(bn module, stride, pad, relu, BnLink)], params: w0, gamma0, beta0, w1, gamma1, beta1, ...
        ctx.bf16 = ops.BF16
        x = ops.to_nhwc(x)
        L = len(specs)
        ws = [ops.to_nhwc(params[3 * i]) for i in range(L)]
        ys, ks = [], []
        for i, (bn, stride, pad, relu, _link) in enumerate(specs):
            y, slab = ops.conv_fprop(x, ws[i], None, stride, pad, False, want_stats=True)
            ys.append((y, slab))
            ks.append(ws[i].shape[0])
        tot = 2 * sum(ks) + L
        packed = ops._ZEROS.take(tot, x.device)                      # [sums_0 | sums_1 | .. | count_0 | count_1 | ..]
        # one sample count per layer (layers of different stride see different numbers of pixels), stored into the packed buffer by
        # the layer's slab reduction (no fill launch); after the exchange the finalize kernel leaves it in storage of its own
        counts = [float(y.numel() // k) for (y, _), k in zip(ys, ks)]
        off = 0
        for i, ((y, slab), k) in enumerate(zip(ys, ks)):
            mt = slab.numel() // (2 * k)
            ops._C.check(ops._C.fn("rr_bn_reduce_slab_count")(ops._C.ptr(slab), mt, k, ops._C.ptr(packed[off:off + 2 * k]), counts[i],
                                                              ops._C.ptr(packed[tot - L + i:]), ops._C.stream()), "rr_bn_reduce_slab_count")
            off += 2 * k
        dptrace.all_reduce(packed, None, "default", "syncbn_fwd x%d" % L)
        cnt_devs = []
        outs, saved, off = [], [], 0
        for i, ((bn, stride, pad, relu, link), (y, _), k) in enumerate(zip(specs, ys, ks)):
            gamma, beta = params[3 * i + 1], params[3 * i + 2]
            mom = bn.momentum if bn.momentum is not None else 0.1
            mean, invstd, scale, shift, cnt = ops.bn_finalize_sync(packed[off:off + 2 * k], packed[tot - L + i:], gamma, beta,
                                                                   bn.running_mean, bn.running_var, mom, bn.eps, bn.num_batches_tracked)
            cnt_devs.append(cnt)
            off += 2 * k
            outs.append(ops.bn_apply(y, scale, shift, None, relu))
            if link is not None:
                # what a consumer's data gradient needs to produce this layer's BatchNorm-backward sums in its epilogue (as
                # _ConvBnAct does: no residual here, so the ReLU mask is recomputed from y)
                link.y, link.mean, link.invstd, link.use_z = y, mean, invstd, False
                link.msc, link.msh = (scale, shift) if relu else (None, None)
            saved += [y, mean, invstd, gamma, scale if relu else None, shift if relu else None]
        ctx.save_for_backward(x, *cnt_devs, *ws, *saved)
        ctx.meta = (L, [(st, tuple(pd), rl) for (_, st, pd, rl, _l) in specs], ks, counts, x_acc, tuple(x.shape))
        ctx.links = [lk for (_, _, _, _, lk) in specs]
        ctx.params = params
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dzs):
        with ops.bf16_scope(ctx.bf16):
            return _ConvBnSyncMulti._backward(ctx, *dzs)

    @staticmethod
    def _backward(ctx, *dzs):
        L, cfgs, ks, counts, x_acc, xshape = ctx.meta
        t = ctx.saved_tensors
        x, cnt_devs, ws, rest = t[0], t[1:1 + L], t[1 + L:1 + 2 * L], t[1 + 2 * L:]
        params = ctx.params
        packed = ops._ZEROS.take(2 * sum(ks), x.device)
        dzl, off = [], 0
        for i in range(L):
            y, mean, invstd, gamma, msc, msh = rest[6 * i:6 * i + 6]
            dz = ops.to_nhwc(dzs[i]) if dzs[i] is not None else torch.zeros_like(y)
            dzl.append(dz)
            k = ks[i]
            n, c, h, w = y.shape
            link = ctx.links[i]
            fused = None
            if link is not None and link.sums is not None:
                # the data gradient that produced dz already reduced it (rr_conv_dgrad_s1_bnsum) — valid only if what arrives
                # here is that very tensor
                if link.dz is not None and link.dz.data_ptr() == dz.data_ptr() and link.dz.shape == dz.shape:
                    fused = link.sums
                link.sums = link.dz = None
            if fused is not None:
                packed[off:off + 2 * k].copy_(fused[:2 * k])
            else:
                ops._C.check(ops._C.fn("rr_bn_bwd_reduce")(ops._C.ptr(dz), ops._C.ptr(None), ops._C.ptr(y), ops._C.ptr(mean),
                                                           ops._C.ptr(invstd), ops._C.ptr(msc), ops._C.ptr(msh),
                                                           ops._C.ptr(packed[off:off + 2 * k]), n * h * w, c, 1, ops._C.stream()),
                             "rr_bn_bwd_reduce")
            # gradients of gamma / beta come from the LOCAL sums (averaged later like any gradient)
            gp, bp = params[3 * i + 1], params[3 * i + 2]
            dg_t, db_t = _grad_target(gp), _grad_target(bp)
            if dg_t is not None:
                if dg_t.is_cuda:
                    ops.bn_affine_grad(packed[off:off + 2 * k], dg_t, db_t)
                else:
                    db_t.add_(packed[off:off + k].float())
                    dg_t.add_(packed[off + k:off + 2 * k].float())
                _mark(gp, bp)
            off += 2 * k
        local = None
        if any(_grad_target(params[3 * i + 1]) is None for i in range(L)):
            local = packed.clone()
        dptrace.all_reduce(packed, None, "default", "syncbn_bwd x%d" % L)
        if x_acc is not None:
            x_acc.pending += L - 1                 # this node contributes L data gradients to the fan-in buffer
        grads, dx, off = [], None, 0
        for i in range(L):
            y, mean, invstd, gamma, msc, msh = rest[6 * i:6 * i + 6]
            stride, pad, relu = cfgs[i]
            k = ks[i]
            dy, _ = ops.bn_bwd_apply(dzl[i], None, y, mean, invstd, gamma, packed[off:off + 2 * k], counts[i], False, None, None,
                                     cnt_devs[i], msc, msh)
            if ctx.needs_input_grad[0]:
                if x_acc is not None:
                    r = _input_grad(dy, ws[i], xshape, stride, pad, x_acc, None, x, params[3 * i])
                    dx = r if r is not None else dx
                elif dx is None:
                    dx = ops.conv_dgrad(dy, ws[i], xshape, stride, pad)
                else:
                    ops.conv_dgrad(dy, ws[i], xshape, stride, pad, out=dx, accumulate=True)
            w = params[3 * i]
            w_t = _grad_target(w)
            if w_t is not None:
                ops.conv_wgrad(x, dy, w_t, stride, pad)
                _mark(w)
                grads += [None]
            else:
                dw = ops.zeros_nhwc(*w.shape, device=x.device)
                ops.conv_wgrad(x, dy, dw, stride, pad)
                grads += [dw]
            if _grad_target(params[3 * i + 1]) is None:
                grads += [local[off + k:off + 2 * k].float(), local[off:off + k].float()]
            else:
                grads += [None, None]
            off += 2 * k
        return (dx, None, None) + tuple(grads)


def sync_coalescing(bn):
    """True when the joint SyncBN node applies: statistics exchanged across ranks, training, gradients on."""
    return _SYNC_COALESCE and _is_sync(bn) and bn.training and torch.is_grad_enabled()


def conv_bn_act_multi(x, layers):
    """layers: [(conv, bn, relu), ...] applied to the same x -> list of outputs.  With SyncBN across ranks the layers'
    statistic exchanges are coalesced (one all-reduce per direction for the whole group); otherwise the ordinary
    per-layer nodes run, in the given order."""
    if all(sync_coalescing(b) for _, b, _ in layers):
        x_acc = getattr(x, "_rr_acc", None)
        if x.requires_grad and x_acc is not None:
            x_acc.pending += 1
        links = [ops.BnLink() for _ in layers]
        specs = [(bn, conv.stride[0], tuple(conv.padding), relu, lk) for (conv, bn, relu), lk in zip(layers, links)]
        params = []
        for conv, bn, _ in layers:
            params += [conv.weight, bn.weight, bn.bias]
        outs = list(_ConvBnSyncMulti.apply(x, x_acc, specs, *params))
        for o, lk in zip(outs, links):
            if lk.y is not None:
                o._rr_bnlink = lk          # a single convolution consumer's data gradient may carry this layer's backward sums
        return outs
    return [conv_bn_act(x, conv, bn, relu=relu) for conv, bn, relu in layers]


class _ConvBias(torch.autograd.Function):
    """y = relu?(conv(x, w) + b) — the bias / ReLU live in the conv epilogue
    (detectors/centernet_detector.py:62,73,85-93; fasterrcnn_detector.py:17)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu, x_acc=None, in_link=None, out_link=None):
        ctx.x_acc = x_acc
        ctx.in_link = in_link
        ctx.out_link = out_link
        ctx.bf16 = ops.BF16
        x = ops.to_nhwc(x)
        wc = ops.to_nhwc(w)
        _wamax_attach(w, wc)
        y = ops.conv_fprop(x, wc, b, stride, pad, relu, w16=_w16_of(w)[0])
        ctx.save_for_backward(x, wc, y if relu else None)
        ctx.x_amax = ops.amax_carry(x)
        ctx.x_b16 = ops.b16_carry(x)
        ctx.cfg = (stride, pad, relu)
        ctx.params = (w, b)
        ctx.xshape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        with ops.bf16_scope(ctx.bf16):
            return _ConvBias._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, wc, y = ctx.saved_tensors
        ops.amax_restore(x, getattr(ctx, "x_amax", None))
        ops.b16_restore(x, getattr(ctx, "x_b16", None))
        stride, pad, relu = ctx.cfg
        w, b = ctx.params
        dy = ops.to_nhwc(dy)
        ret_db = None
        ol = ctx.out_link
        fused = None
        if ol is not None and ol.sums is not None:
            # the consumer's data gradient already masked dy with this layer's ReLU and summed its columns
            if ol.dz is not None and ol.dz.data_ptr() == dy.data_ptr() and ol.dz.shape == dy.shape:
                fused = ol.sums
            ol.sums = ol.dz = None
        if b is not None:
            b_t = _grad_target(b)
            tgt = b_t if b_t is not None else torch.zeros_like(b)
            if fused is not None:
                tgt.add_(fused[:b.numel()].float())
            else:
                dy = ops.bias_relu_bwd(dy, y if relu else None, tgt)
            ret_db = None if b_t is not None else tgt
            if b_t is not None:
                _mark(b)
        elif relu:
            dy = ops.sum_n([dy], y)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dy, wc, ctx.xshape, stride, pad, ctx.x_acc, ctx.in_link, x, w)
        w_t = _grad_target(w)
        ret_dw = None
        if w_t is not None:
            _wgrad_async(lambda: (ops.conv_wgrad(x, dy, w_t, stride, pad), _mark(w)), x.device, x, dy)
        else:
            dw = ops.zeros_nhwc(*w.shape, device=x.device)
            ops.conv_wgrad(x, dy, dw, stride, pad)
            ret_dw = dw
        return dx, ret_dw, ret_db, None, None, None, None, None, None


def _conv_bias_apply(x, weight, bias, stride, pad, relu):
    x_acc = getattr(x, "_rr_acc", None)
    in_link = getattr(x, "_rr_bnlink", None) if x_acc is None else None
    grad_on = torch.is_grad_enabled()
    if grad_on and x.requires_grad:
        if x_acc is not None:
            x_acc.pending += 1
        elif in_link is not None:
            in_link.consumers += 1
    out_link = None
    if grad_on and relu and bias is not None and weight.shape[0] % 4 == 0:
        # conv + bias + ReLU: a single consumer's data gradient can apply this layer's ReLU mask and sum its bias gradient
        out_link = ops.BnLink()
        out_link.relu_bias = out_link.use_z = True
    out = _ConvBias.apply(x, weight, bias, stride, pad, relu, x_acc, in_link, out_link)
    if out_link is not None:
        out._rr_bnlink = out_link
    return out


def conv_bias(x, conv, relu=False):
    return _conv_bias_apply(x, conv.weight, conv.bias, conv.stride[0], tuple(conv.padding), relu)


def conv_weight(x, weight, bias=None, stride=1, pad=(0, 0), relu=False):
    """_ConvBias on a bare weight tensor (e.g. one assembled from several parameters)."""
    return _conv_bias_apply(x, weight, bias, stride, tuple(pad), relu)


class _WHShiftSum(torch.autograd.Function):
    """Second half of the fused WH head: per-tap partial products -> the two 17-tap sums + biases."""

    @staticmethod
    def forward(ctx, t, bias_w, bias_h, k):
        t = ops.to_nhwc(t)
        ctx.k = k
        ctx.ct = t.shape[1]
        ctx.params = (bias_w, bias_h)
        return ops.wh_shift_sum_fwd(t, bias_w, bias_h, k)

    @staticmethod
    def backward(ctx, dout):
        dout = ops.to_nhwc(dout)
        bias_w, bias_h = ctx.params
        db = torch.zeros(2, dtype=torch.float32, device=dout.device)
        ops.bias_relu_bwd(dout, None, db)                     # column sums of [pixels, 2]
        dbw, dbh = db[0:1].view_as(bias_w), db[1:2].view_as(bias_h)
        tw, th = _grad_target(bias_w), _grad_target(bias_h)
        if tw is not None and th is not None:
            tw.add_(dbw)
            th.add_(dbh)
            dbw = dbh = None
            _mark(bias_w, bias_h)
        return ops.wh_shift_sum_bwd(dout, ctx.k, ctx.ct), dbw, dbh, None


def wh_shift_sum(t, bias_w, bias_h, k):
    return _WHShiftSum.apply(t, bias_w, bias_h, k)


class _FanOut(torch.autograd.Function):
    """Identity with n consumers; backward sums the n incoming gradients in ONE pass
    (what autograd would do with n-1 separate add kernels)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        ctx.acc = None                     # the shared GradAcc of the views (fanout_shared), for its stream events
        ctx.set_materialize_grads(False)   # a consumer that accumulated into a shared GradAcc hands back None
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        if ctx.acc is not None:
            dev = next((g.device for g in grads if g is not None), None)
            if dev is not None:
                ctx.acc.begin(dev)         # contributors that added into the buffer from other streams
        gs = [ops.to_nhwc(g) if g.dim() == 4 else g.contiguous() for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        return ops.sum_n(gs), None


def _share_b16(x, views):
    """conv16 kernels: the fan-out views ARE x — its bf16 image (written by x's producer, or converted once here so that the
    n consumers do not convert n times) travels with them."""
    if ops.is_phantom(x):                  # a bf16-only tensor: its views must keep the image, whatever the mode
        img = ops.image_of(x)
        for o in views:
            o._rr_b16 = (o._version, x._rr_b16[1], img)
        return
    if ops.BF16 != ops.MATH_BF16 or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32:
        return
    hit = getattr(x, "_rr_b16", None)
    if hit is None or hit[0] != x._version:
        if not (ops._CONV16 and x.shape[1] % 128 == 0
                and x.shape[0] * x.shape[2] * x.shape[3] >= ops._CONV16_MIN_PIXELS and ops.is_nhwc(x)):
            return
        ops.bf16_of(x)
        hit = x._rr_b16
    for o in views:
        o._rr_b16 = (o._version, hit[1], hit[2])


def fanout(x, n):
    if n == 1 or not x.requires_grad:
        return (x,) * n
    outs = _FanOut.apply(x, n)
    _share_b16(x, outs)
    return outs


_SHARED_ACC = True

class GradAcc:
    """Fan-in target shared by the consumers of one fan-out: the first consumer to run its backward publishes its
    input gradient here, the others add into it (inside their dgrad epilogue) and hand autograd `None`, so the
    fan-out's backward finds one complete gradient and launches no sum kernel.  The views a fan-out returns carry
    the accumulator as `_rr_acc`; convolution nodes pick it up from their input, and a nested fan-out of such a view
    (a residual block at the head of an hourglass branch) joins the same accumulator."""
    __slots__ = ("buf", "pending", "link")

    def __init__(self):
        self.buf = None
        self.pending = 0        # registered contributors that have not run their backward yet
        self.link = None        # ops.BnLink of the fanned-out tensor when a conv -> bn layer produced it

    # The buffer is written through raw pointers, which autograd's own stream hand-over never sees: every contributor
    # brackets its write with begin / end.  All contributors run on the caller's stream today, so both are no-ops — the
    # hourglass branch streams that needed events here were measured (0 to -12 %) and removed in round 6; the
    # bracket stays as the one place where an ordering would have to go.
    def begin(self, device):
        pass

    def end(self, device):
        pass


def fanout_shared(x, n):
    """fanout whose views carry a shared GradAcc -> (view_1, .., view_n, acc); acc is None when x carries no
    gradient."""
    if n == 1 or not x.requires_grad:
        return (x,) * n + (None,)
    outs = _FanOut.apply(x, n)
    if not _SHARED_ACC:
        return outs + (None,)
    acc = getattr(x, "_rr_acc", None)
    if acc is None:
        acc = GradAcc()
        acc.link = getattr(x, "_rr_bnlink", None)
    if outs[0].grad_fn is not None:         # (None under torch.no_grad(): nothing will run a backward)
        outs[0].grad_fn.acc = acc           # (the node object is the ctx its backward receives)
    amax = getattr(x, "_rr_amax", None)     # (split-operand kernels: the views ARE x — its remembered maximum travels with them)
    for o in outs:
        o._rr_acc = acc
        if amax is not None and amax[0] == x._version:
            o._rr_amax = (o._version, amax[1], amax[2])
    _share_b16(x, outs)
    return outs + (acc,)


class _ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_link=None):
        x = ops.to_nhwc(x) if x.dim() == 4 else x.contiguous()
        z = ops.relu_fwd(x)
        ctx.save_for_backward(z)
        ctx.out_link = out_link
        return z

    @staticmethod
    def backward(ctx, dz):
        (z,) = ctx.saved_tensors
        dz = ops.to_nhwc(dz) if dz.dim() == 4 else dz.contiguous()
        ol = ctx.out_link
        if ol is not None and ol.sums is not None:
            # the data gradient that completed dz (the one consumer, or the last contributor of the fan-in) already
            # stored it masked with z > 0: the backward of the ReLU is the identity on that very tensor
            same = ol.dz is not None and ol.dz.data_ptr() == dz.data_ptr() and ol.dz.shape == dz.shape
            ol.sums = ol.dz = None
            if same:
                return dz, None
        return ops.sum_n([dz], z), None


def relu(x):
    link = None
    if torch.is_grad_enabled() and x.requires_grad and x.dim() == 4 and x.shape[1] % 4 == 0:
        link = ops.BnLink()
        link.relu_bias = link.use_z = link.mask_only = True
    out = _ReLU.apply(x, link)
    if link is not None:
        out._rr_bnlink = link
    return out


class _UpsampleAdd(torch.autograd.Function):
    """up1 + bilinear_align_corners(nearest2x(low), size(up1))  (backbones/hourglass.py:121-124)."""

    @staticmethod
    def forward(ctx, up1, low):
        up1, low = ops.to_nhwc(up1, keep_phantom=True), ops.to_nhwc(low, keep_phantom=True)
        ctx.low_shape = tuple(low.shape)
        return ops.upsample_add_fwd(up1, low, bf16_only=ops.phantom_out_ok(up1.shape[1], up1.shape[0] * up1.shape[2] * up1.shape[3],
                                                                           up1.device))

    @staticmethod
    def backward(ctx, dout):
        dout = ops.to_nhwc(dout)
        dlow = ops.upsample_add_bwd(dout, ctx.low_shape) if ctx.needs_input_grad[1] else None
        return dout, dlow


def upsample_add(up1, low):
    return _UpsampleAdd.apply(up1, low)


class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = ops.to_nhwc(x)
        ctx.shape = tuple(x.shape)
        return ops.avgpool_fwd(x)

    @staticmethod
    def backward(ctx, dout):
        return ops.avgpool_bwd(ops.to_nhwc(dout), ctx.shape)


def global_avg_pool(x):
    return _AvgPool.apply(x)


class _RoIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, out_size, spatial_scale, sampling_ratio):
        feat = ops.to_nhwc(feat)
        rois = rois.contiguous()
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(feat.shape), out_size, spatial_scale, sampling_ratio)
        return ops.roi_align_fwd(feat, rois, out_size, spatial_scale, sampling_ratio)

    @staticmethod
    def backward(ctx, dout):
        (rois,) = ctx.saved_tensors
        shape, out_size, scale, sr = ctx.cfg
        return ops.roi_align_bwd(ops.to_nhwc(dout), rois, shape, out_size, scale, sr), None, None, None, None


def roi_align(feat, rois, out_size, spatial_scale=1.0, sampling_ratio=-1):
    """torchvision.ops.roi_align signature (models/rrnet.py:51)."""
    if isinstance(out_size, int):
        out_size = (out_size, out_size)
    return _RoIAlign.apply(feat, rois.detach(), tuple(out_size), float(spatial_scale), int(sampling_ratio))


# ---------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------
class _FocalHM(torch.autograd.Function):
    """clamp(sigmoid(x),1e-4,1-1e-4) + focal_loss_for_hm in one fused pass
    (operators/rrnet_operator.py:55-57, modules/loss/functional.py:25-51)."""

    @staticmethod
    def forward(ctx, logits, gt):
        logits = ops.to_nhwc(logits)
        gt = ops.to_nhwc(gt)
        sums = ops.focal_fwd(logits, gt)
        ctx.save_for_backward(logits, gt, sums)
        npos = sums[2]
        tot = sums[0] + sums[1]
        loss = torch.where(npos > 0, -tot / torch.clamp(npos, min=1.0), -sums[1])
        return loss.float()

    @staticmethod
    def backward(ctx, gout):
        logits, gt, sums = ctx.saved_tensors
        return ops.focal_bwd(logits, gt, sums, gout.contiguous().float()), None


def focal_loss_hm_from_logits(logits, gt):
    return _FocalHM.apply(logits, gt)


class _RegL1(torch.autograd.Function):
    """modules/loss/regl1loss.py:9-17 without the NHWC permute copy of the whole map."""

    @staticmethod
    def forward(ctx, pred, mask, ind, target):
        pred = ops.to_nhwc(pred)
        mask = mask.contiguous().float()
        ind = ind.contiguous().float()
        target = target.contiguous().float()
        sums = ops.regl1_fwd(pred, mask, ind, target)
        ctx.save_for_backward(pred, mask, ind, target, sums)
        return (sums[0] / (sums[1] + 1e-4)).float()

    @staticmethod
    def backward(ctx, gout):
        pred, mask, ind, target, sums = ctx.saved_tensors
        return ops.regl1_bwd(pred, mask, ind, target, sums, gout.contiguous().float()), None, None, None


def reg_l1_loss(pred, mask, ind, target):
    return _RegL1.apply(pred, mask, ind, target)


class _Stage2Loss(torch.autograd.Function):
    """box_iou + positive matching + generate_bbox_target + smooth-L1 of
    operators/rrnet_operator.py:63-102 as small kernels (no per-image host loop, no sync).  Like the
    reference's F.smooth_l1_loss it differentiates its targets too: when `rois` carries a graph (hard-NMS
    proposals, models/rrnet.py:70) dloss/drois is returned."""

    @staticmethod
    def forward(ctx, reg, rois, gt_xyxy, scale):
        reg = reg.contiguous()
        want = rois.requires_grad
        loss, dreg, _tgt, _pos, _npos, droi = ops.stage2_loss(rois.detach().contiguous(), reg, gt_xyxy.contiguous(),
                                                             scale, want_droi=want)
        ctx.save_for_backward(dreg, droi)
        return loss[0].float()

    @staticmethod
    def backward(ctx, gout):
        dreg, droi = ctx.saved_tensors
        d_rois = None
        if droi is not None:
            d_rois = torch.zeros((droi.shape[0], 5), dtype=droi.dtype, device=droi.device)
            d_rois[:, 1:] = droi * gout
        return dreg * gout, d_rois, None, None


def stage2_reg_loss(reg, rois, gt_xyxy, scale):
    return _Stage2Loss.apply(reg, rois, gt_xyxy, float(scale))


class _Proposals(torch.autograd.Function):
    """Makes the packed stage-1 boxes a differentiable function of the last stack's wh / offset maps, as they
    are in the reference's hard-NMS path (transform_bbox's arithmetic + index selects).  Forward = identity
    on the precomputed RoIs; backward = scatter of d roi through the box assembly."""

    @staticmethod
    def forward(ctx, wh, offset, rois, roi_pix):
        ctx.save_for_backward(ops.to_nhwc(wh), rois, roi_pix)
        return rois.clone()

    @staticmethod
    def backward(ctx, d_rois):
        wh, rois, roi_pix = ctx.saved_tensors
        droi = d_rois[:, 1:].contiguous()
        dwh, doff = ops.proposal_bwd(droi, rois, roi_pix, wh)
        return dwh, doff, None, None


def differentiable_proposals(wh, offset, rois, roi_pix):
    return _Proposals.apply(wh, offset, rois, roi_pix)


# ---------------------------------------------------------------------------------------------
# DCNv2 (ext/dcn/dcn_v2.py:16-52)
# ---------------------------------------------------------------------------------------------
def _acc_publish(acc, g):
    """A non-convolution contributor's gradient g of a fanned-out tensor -> what to hand autograd: g itself when it is the
    first (it becomes the shared buffer), None after adding it to the buffer otherwise.  (The caller has taken itself off
    acc.pending.)"""
    if acc.buf is None:
        acc.buf = g
        acc.end(g.device)
        return g
    acc.begin(g.device)
    ops.amax_drop(acc.buf)
    acc.buf.add_(g)
    acc.end(g.device)
    return None


class _DCNv2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, bf16=False, x_acc=None):
        x, off, m, w = ops.to_nhwc(input), ops.to_nhwc(offset), ops.to_nhwc(mask), ops.to_nhwc(weight)
        ctx.cfg = (int(stride), tuple(padding), int(dilation), int(deformable_groups))
        ctx.save_for_backward(x, off, m, w)
        ctx.params = (weight, bias)
        ctx.bf16 = bool(bf16)
        ctx.x_acc = x_acc                  # shared fan-in buffer of input's fan-out (dcn_v2_conv registered this node as a contributor)
        return ops.dcn_fwd(x, off, m, w, bias, *ctx.cfg, bf16=bf16)

    @staticmethod
    def backward(ctx, dy):
        x, off, m, w = ctx.saved_tensors
        stride, pad, dil, dg = ctx.cfg
        weight, bias = ctx.params
        dy = ops.to_nhwc(dy)
        k, c, r, s = w.shape
        w_t = _grad_target(weight)
        dw = w_t if w_t is not None else ops.zeros_nhwc(*weight.shape, device=x.device)
        if DCN_FUSED_BWD and ops.dcn_fused_bwd_supported(c, k, r, s, stride, dg, dil):
            # fused: the columns live in registers / LDS inside the two GEMM kernels (csrc/dcn.hip).  The two kernels
            # are independent and bound by different things (wgrad: corner gather + MFMA; dgrad: the float atomics of
            # d input at the memory side), so they run concurrently on two HIP streams.
            cur = torch.cuda.current_stream()
            side = _side_stream(x.device) if DCN_BWD_STREAMS else None
            # a bf16 forward takes the bf16-operand gradients (input window in LDS, d input pre-summed on chip)
            bf = bool(ctx.bf16 and DCN_BF16_BWD)
            # both kernels take dY's bf16 image by LDS-DMA: its producer's, or ONE conversion pass made here, before the fork
            # (behind the weight-gradient kernel, which holds every CU, the pass took 2.4 ms instead of 0.2)
            img = ops.bf16_of(dy) if (bf and ops._DCN_DYB and not ops.is_phantom(dy)) else None
            if side is not None:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    _stress_delay()
                    ops.dcn_wgrad(x, off, m, dy, dw, stride, pad, dil, dg, bf16=bf, dy_img=img)
            else:
                ops.dcn_wgrad(x, off, m, dy, dw, stride, pad, dil, dg, bf16=bf, dy_img=img)
            x_acc = ctx.x_acc
            if x_acc is None:
                dx, doff, dmask = ops.dcn_dgrad(x, off, m, w, dy, stride, pad, dil, dg, bf16=bf)
            else:
                # input's fan-out shares one gradient buffer among its consumers (GradAcc): the first to run publishes its
                # gradient there, the others add into it — here inside the kernel, whose scatter is float atomics anyway
                x_acc.pending -= 1
                if x_acc.buf is not None and ops.dcn_dgrad_accumulates(bf):
                    x_acc.begin(dy.device)
                    _, doff, dmask = ops.dcn_dgrad(x, off, m, w, dy, stride, pad, dil, dg, bf16=bf, out=x_acc.buf)
                    x_acc.end(dy.device)
                    dx = None
                else:
                    dx, doff, dmask = ops.dcn_dgrad(x, off, m, w, dy, stride, pad, dil, dg, bf16=bf)
                    dx = _acc_publish(x_acc, dx)
            if side is not None:
                cur.wait_stream(side)
        else:
            # column path (the reference's structure): columns and their gradient materialised, GEMMs on the conv kernels
            mtot = dy.shape[0] * dy.shape[2] * dy.shape[3]
            dy1 = dy.permute(0, 2, 3, 1).reshape(1, mtot, 1, k).permute(0, 3, 1, 2)            # [1,k,M,1] view
            w1 = w.permute(0, 2, 3, 1).reshape(k, 1, 1, r * s * c).permute(0, 3, 1, 2)          # [k, r*s*c, 1, 1] view
            col = ops.dcn_im2col(x, off, m, r, s, stride, pad, dil, dg)
            dw1 = dw.permute(0, 2, 3, 1).reshape(k, 1, 1, r * s * c).permute(0, 3, 1, 2)
            ops.conv_wgrad(col, dy1, dw1, 1, (0, 0))
            del col
            dcol = ops.conv_dgrad(dy1, w1, (1, r * s * c, mtot, 1), 1, (0, 0))
            dx, doff, dmask = ops.dcn_col2im(x, off, m, dcol, r, s, stride, pad, dil, dg)
            if ctx.x_acc is not None:
                ctx.x_acc.pending -= 1
                dx = _acc_publish(ctx.x_acc, dx)
        if w_t is not None:
            _mark(weight)
        db = None
        if bias is not None:
            b_t = _grad_target(bias)
            tgt = b_t if b_t is not None else torch.zeros_like(bias)
            ops.bias_relu_bwd(dy, None, tgt)
            db = None if b_t is not None else tgt
            if b_t is not None:
                _mark(bias)
        return dx, doff, dmask, (None if w_t is not None else dw), db, None, None, None, None, None, None


class _DCNSplit(torch.autograd.Function):
    """ext/dcn/dcn_v2.py:117-121: chunk(3) -> offset = cat(o1, o2), mask = sigmoid(o3), as one pass each way."""

    @staticmethod
    def forward(ctx, om):
        offset, mask = ops.dcn_split_fwd(ops.to_nhwc(om))
        ctx.save_for_backward(mask)
        return offset, mask

    @staticmethod
    def backward(ctx, doffset, dmask):
        (mask,) = ctx.saved_tensors
        if doffset is None:
            doffset = torch.zeros((mask.shape[0], 2 * mask.shape[1]) + tuple(mask.shape[2:]), device=mask.device)
        if dmask is None:
            dmask = torch.zeros_like(mask)
        return ops.dcn_split_bwd(ops.to_nhwc(doffset), ops.to_nhwc(dmask), mask)


def dcn_offset_mask(om):
    return _DCNSplit.apply(om)


class _DCNv2Pooling(torch.autograd.Function):
    """ext/dcn/dcn_v2.py:130-182 (`_DCNv2Pooling`): deformable position-sensitive RoI pooling."""

    @staticmethod
    def forward(ctx, input, rois, offset, spatial_scale, pooled_size, output_dim, no_trans, group_size=1, part_size=None,
                sample_per_part=4, trans_std=.0):
        part_size = pooled_size if part_size is None else part_size
        x = ops.to_nhwc(input)
        rois = rois.contiguous().float()
        no_trans = int(no_trans)
        trans = None if no_trans else offset.contiguous().float()
        ctx.cfg = (no_trans, float(spatial_scale), int(output_dim), int(group_size), int(pooled_size), int(part_size),
                   int(sample_per_part), float(trans_std))
        out, count = ops.dcn_psroi_fwd(x, rois, trans, *ctx.cfg)
        ctx.save_for_backward(x, rois, trans, count)
        ctx.has_offset = not no_trans
        return out

    @staticmethod
    def backward(ctx, grad_output):
        x, rois, trans, count = ctx.saved_tensors
        dx, dtrans = ops.dcn_psroi_bwd(ops.to_nhwc(grad_output), x, rois, trans, count, *ctx.cfg)
        return (dx, None, dtrans if ctx.has_offset else None) + (None,) * 8


dcn_v2_pooling = _DCNv2Pooling.apply

_SIDE = {}


def _side_stream(device, tag="dcn"):
    """Auxiliary HIP streams (one per device and purpose) for work that may overlap the current stream."""
    key = (device.type, device.index, tag)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
        ops.AUX_STREAMS.setdefault((device.type, device.index), []).append(_SIDE[key])
    return _SIDE[key]


DCN_BF16_BWD = True         # bf16 forward -> bf16-operand wgrad / dgrad (0: fp32 kernels)
DCN_BWD_STREAMS = os.environ.get("RR_DCN_BWD_STREAMS", "1") != "0"   # wgrad and dgrad of the fused backward side by side
DCN_FUSED_BWD = True   # 0: the reference's column-buffer structure (A/B switch)
DCN_BF16 = os.environ.get("RR_DCN_BF16", "0") == "1"   # BASELINE config 4: bf16 matrix operands in the DCN forward


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, bf16=None):
    """Reference signature (ext/dcn/dcn_v2.py:52).  stride / dilation: int or equal pair; padding: int or pair.
    bf16 (extension): round the sampled columns and the weights to bf16 for the forward MFMA (fp32 accumulation,
    fp32 backward); default from RR_DCN_BF16."""
    st = stride if isinstance(stride, int) else stride[0]
    dl = dilation if isinstance(dilation, int) else dilation[0]
    pd = (padding, padding) if isinstance(padding, int) else tuple(padding)
    x_acc = getattr(input, "_rr_acc", None)      # input is a view of a shared fan-out (fanout_shared): contribute to its buffer
    if x_acc is not None:
        if torch.is_grad_enabled() and input.requires_grad:
            x_acc.pending += 1
        else:
            x_acc = None
    return _DCNv2.apply(input, offset, mask, weight, bias, st, pd, dl, deformable_groups,
                        DCN_BF16 if bf16 is None else bool(bf16), x_acc)
