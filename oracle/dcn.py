"""ORACLE (test infrastructure only) — torch-CPU restatement of the reference's modulated deformable
convolution (DCNv2).  Only tests/ import this file.

The reference op has NO CPU path (`ext/dcn/src/dcn_v2.h:38,72` raise "Not implemented on the CPU") and
cannot be built here (CUDA), so no golden vector can be generated from it: **parity is pinned by
properties, not by reference outputs** —
  * the reference's own analytic check `ext/dcn/test.py:32-67` (zero offsets + identity weight +
    mask = sigmoid(0) = 0.5  =>  2*out == in), reproduced in tests/test_oracle_dcn.py;
  * zero offsets + unit mask  =>  F.conv2d (any stride / padding / dilation);
  * integer offsets  =>  a shifted F.conv2d tap;  fractional offsets  =>  F.grid_sample bilinear values;
  * the gradcheck configuration of `ext/dcn/test.py:69-97` (N,C,H,W = 2,2,4,4, 3x3, eps 1e-3,
    atol 1e-4, rtol 1e-2) run on this restatement in float64.
Restated from:
  ext/dcn/src/cuda/dcn_v2_im2col_cuda.cu:25-54   dmcn_im2col_bilinear (corner-wise zero padding)
  ext/dcn/src/cuda/dcn_v2_im2col_cuda.cu:125-195 modulated_deformable_im2col_gpu_kernel
      (sample is zero unless -1 < h < H and -1 < w < W; offset channels are interleaved
       (dh, dw) pairs per tap inside each deformable group: 2*(i*kw+j), +1)
  ext/dcn/src/cuda/dcn_v2_cuda.cu:126-163        out = bias + W . columns
  ext/dcn/dcn_v2.py:92-122                       DCN: offsets / mask from a zero-initialised conv,
                                                 offset = cat(o1, o2), mask = sigmoid(.)
The backward is torch autograd of this forward, which equals the reference's hand-written
col2im / coord kernels (:197-327): d/dh of the bilinear weights with the same corner validity, zero
outside the (-1, H) x (-1, W) window.
"""
import torch
import torch.nn.functional as F


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _bilinear_zero(x, h, w):
    """x [N,C,H,W]; h, w [N,P,Q] sample coordinates -> [N,C,P,Q] with the reference's rules."""
    n, c, H, W = x.shape
    inside = (h > -1) & (w > -1) & (h < H) & (w < W)
    h0 = torch.floor(h)
    w0 = torch.floor(w)
    lh, lw = h - h0, w - w0
    hh, hw = 1 - lh, 1 - lw
    h0, w0 = h0.long(), w0.long()
    h1, w1 = h0 + 1, w0 + 1
    flat = x.reshape(n, c, H * W)

    def corner(hi, wi, ok):
        ok = ok & inside
        idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).reshape(n, 1, -1).expand(n, c, -1)
        v = flat.gather(2, idx).reshape(n, c, *h.shape[1:])
        return v * ok.unsqueeze(1).to(x.dtype)

    v1 = corner(h0, w0, (h0 >= 0) & (w0 >= 0))
    v2 = corner(h0, w1, (h0 >= 0) & (w1 <= W - 1))
    v3 = corner(h1, w0, (h1 <= H - 1) & (w0 >= 0))
    v4 = corner(h1, w1, (h1 <= H - 1) & (w1 <= W - 1))
    return ((hh * hw).unsqueeze(1) * v1 + (hh * lw).unsqueeze(1) * v2 +
            (lh * hw).unsqueeze(1) * v3 + (lh * lw).unsqueeze(1) * v4)


def _position(base, off, origin, pos_fp32):
    """Sample coordinate = integer grid position + learned offset.  pos_fp32: evaluated as the reference's kernel does — ONE
    float32 addition of the (exact) integer position and the offset (dcn_v2_im2col_cuda.cu:159-160, `const float h_im = h_in +
    i * dilation_h + offset_h`): at row 200 that rounds the position to 1.5e-5 of a pixel, which an fp64 restatement would
    not reproduce.  origin: the crop's first row / column in the full map, so that a cropped window rounds exactly like the
    full tensor.  The gradient passes straight through (the rounding is a constant shift)."""
    if not pos_fp32:
        return base + off
    rounded = ((base + origin).to(torch.float32) + off.detach().to(torch.float32)).to(off.dtype)
    return off + (rounded - origin - off.detach())


def dcn_columns(x, offset, mask, kh, kw, stride=1, padding=0, dilation=1, deformable_groups=1, pos_fp32=False, origin=(0, 0)):
    """The modulated deformable columns [N, C, kh*kw, P, Q] = mask * bilinear sample (dcn_v2_im2col_cuda.cu:125-195)."""
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    n, c, H, W = x.shape
    dg = deformable_groups
    cpg = c // dg
    P = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Q = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    base_h = (torch.arange(P, dtype=x.dtype, device=x.device) * sh - ph).view(1, P, 1)
    base_w = (torch.arange(Q, dtype=x.dtype, device=x.device) * sw - pw).view(1, 1, Q)
    cols = []
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg]
        taps = []
        for i in range(kh):
            for j in range(kw):
                t = i * kw + j
                oh = offset[:, g * 2 * kh * kw + 2 * t]
                ow = offset[:, g * 2 * kh * kw + 2 * t + 1]
                val = _bilinear_zero(xg, _position(base_h + i * dh, oh, origin[0], pos_fp32), _position(base_w + j * dw, ow, origin[1], pos_fp32))
                taps.append(val * mask[:, g * kh * kw + t].unsqueeze(1))
        cols.append(torch.stack(taps, dim=2))
    return cols[0] if dg == 1 else torch.cat(cols, dim=1)


def _bf16_round(t):
    return t.detach().to(torch.bfloat16).to(t.dtype)


class _ContractBf16(torch.autograd.Function):
    """out = cols . W with the builder-defined bf16 contract of BASELINE configs[3] (rrnet_amd/csrc/dcn.hip: rr_dcn_fwd_bf16,
    rr_dcn_dgrad_bf16, rr_dcn_wgrad_bf16) restated: BOTH operands of every matrix product rounded to bf16 (nearest even),
    products and sums in the tensors' own dtype — forward round(cols) . round(W); backward d cols = round(dY) . round(W),
    d W = round(dY)^T . round(cols).  Everything around the products (sampling, mask, bias, the column gradient's way
    back to input / offset / mask) stays in the tensors' dtype.  The reference is fp32-only (dcn_v2_cuda.cu:58)."""

    @staticmethod
    def forward(ctx, cols, w3):
        cq, wq = _bf16_round(cols), _bf16_round(w3)
        ctx.save_for_backward(cq, wq)
        return torch.einsum('nctpq,kct->nkpq', cq, wq)

    @staticmethod
    def backward(ctx, dy):
        cq, wq = ctx.saved_tensors
        dq = _bf16_round(dy)
        return torch.einsum('nkpq,kct->nctpq', dq, wq), torch.einsum('nkpq,nctpq->kct', dq, cq)


def dcn_v2_conv(x, offset, mask, weight, bias, stride=1, padding=0, dilation=1, deformable_groups=1, bf16=False, pos_fp32=False,
                origin=(0, 0)):
    """`dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups)` of
    ext/dcn/dcn_v2.py:16-52.  x [N,C,H,W]; offset [N, 2*dg*kh*kw, P, Q]; mask [N, dg*kh*kw, P, Q].
    bf16 (builder-defined, configs[3]): the matrix products on bf16-rounded operands (_ContractBf16)."""
    k, c, kh, kw = weight.shape
    cols = dcn_columns(x, offset, mask, kh, kw, stride, padding, dilation, deformable_groups, pos_fp32, origin)
    w3 = weight.reshape(k, c, kh * kw)
    out = _ContractBf16.apply(cols, w3) if bf16 else torch.einsum('nctpq,kct->nkpq', cols, w3)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def dcn_forward(x, weight, bias, om_weight, om_bias, stride=1, padding=1, dilation=1, deformable_groups=1, conv=None, bf16=False):
    """`DCN.forward` of ext/dcn/dcn_v2.py:114-128: offsets and mask come from conv_offset_mask.
    conv: the caller's convolution for the offset / mask layer (oracle/model.py's bf16-contract conv); bf16: see dcn_v2_conv."""
    if conv is None:
        out = F.conv2d(x, om_weight, om_bias, stride=stride, padding=padding)
    else:
        out = conv(x, om_weight, om_bias, stride, padding)
    o1, o2, m = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    return dcn_v2_conv(x, offset, torch.sigmoid(m), weight, bias, stride, padding, dilation, deformable_groups, bf16=bf16)
