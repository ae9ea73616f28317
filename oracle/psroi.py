"""ORACLE (test infrastructure only) — numpy restatement of the reference's deformable PS-RoI pooling.
Only tests/ import this file.

The reference op is CUDA-only (`ext/dcn/src/dcn_v2.h` raises on CPU tensors) and cannot be built here, and no test of
the reference holds outputs for it: **parity unpinned** — pinned by properties in tests/test_oracle_psroi.py (constant
map -> constant output, linear map -> mean of the sample positions, finite differences of the backward).
Restated line by line from /root/reference/ext/dcn/src/cuda/dcn_v2_psroi_pooling_cuda.cu:
  :30-52    bilinear_interp (floor / ceil corners)
  :59-153   DeformablePSROIPoolForwardKernel
  :155-290  DeformablePSROIPoolBackwardAccKernel (the trans gradient ignores the clamp of the sample position, as the
            reference does)
  :300-418  host: num_classes = trans channels / 2, channels_each_class = output_dim / num_classes, count output.
All arithmetic in float32 like the kernel (`T = float`)."""
import numpy as np

f32 = np.float32


def _bins(rois, trans, n, ctop, ph, pw, no_trans, scale, out_dim, gs, P, part, spp, trans_std):
    q = rois[n]
    b = int(q[0])
    sw = f32(f32(np.round(q[1])) * f32(scale) - f32(0.5))
    sh = f32(f32(np.round(q[2])) * f32(scale) - f32(0.5))
    ew = f32(f32(np.round(q[3]) + f32(1)) * f32(scale) - f32(0.5))
    eh = f32(f32(np.round(q[4]) + f32(1)) * f32(scale) - f32(0.5))
    roi_w = f32(max(f32(ew - sw), f32(0.1)))
    roi_h = f32(max(f32(eh - sh), f32(0.1)))
    bin_h, bin_w = f32(roi_h / f32(P)), f32(roi_w / f32(P))
    sub_h, sub_w = f32(bin_h / f32(spp)), f32(bin_w / f32(spp))
    part_h = int(np.floor(f32(ph) / f32(P) * f32(part)))
    part_w = int(np.floor(f32(pw) / f32(P) * f32(part)))
    num_classes = 1 if no_trans else trans.shape[1] // 2
    cpc = out_dim if no_trans else out_dim // num_classes
    cls = ctop // cpc
    tx = f32(0) if no_trans else f32(trans[n, cls * 2, part_h, part_w] * f32(trans_std))
    ty = f32(0) if no_trans else f32(trans[n, cls * 2 + 1, part_h, part_w] * f32(trans_std))
    wstart = f32(f32(f32(pw) * bin_w + sw) + f32(tx * roi_w))
    hstart = f32(f32(f32(ph) * bin_h + sh) + f32(ty * roi_h))
    gw = min(max(int(np.floor(f32(pw) * f32(gs) / f32(P))), 0), gs - 1)
    gh = min(max(int(np.floor(f32(ph) * f32(gs) / f32(P))), 0), gs - 1)
    c = (ctop * gs + gh) * gs + gw
    return b, c, cls, part_h, part_w, wstart, hstart, sub_w, sub_h, roi_w, roi_h


def _samples(wstart, hstart, sub_w, sub_h, spp, H, W):
    for ih in range(spp):
        for iw in range(spp):
            w = f32(wstart + f32(iw) * sub_w)
            h = f32(hstart + f32(ih) * sub_h)
            if w < -0.5 or w > W - 0.5 or h < -0.5 or h > H - 0.5:
                continue
            w = f32(min(max(w, f32(0)), f32(W - 1)))
            h = f32(min(max(h, f32(0)), f32(H - 1)))
            yield w, h


def psroi_forward(x, rois, trans, no_trans, scale, out_dim, gs, P, part, spp, trans_std):
    """x [B,C,H,W] float32, rois [N,5], trans [N,2*classes,part,part] -> out, count [N,out_dim,P,P]."""
    B, C, H, W = x.shape
    N = rois.shape[0]
    out = np.zeros((N, out_dim, P, P), f32)
    cnt = np.zeros((N, out_dim, P, P), f32)
    for n in range(N):
        for ctop in range(out_dim):
            for ph in range(P):
                for pw in range(P):
                    b, c, _, _, _, ws, hs, sw_, sh_, _, _ = _bins(rois, trans, n, ctop, ph, pw, no_trans, scale, out_dim,
                                                                 gs, P, part, spp, trans_std)
                    s, k = f32(0), 0
                    for w, h in _samples(ws, hs, sw_, sh_, spp, H, W):
                        x1, x2, y1, y2 = int(np.floor(w)), int(np.ceil(w)), int(np.floor(h)), int(np.ceil(h))
                        dx, dy = f32(w - x1), f32(h - y1)
                        v = f32((1 - dx) * (1 - dy) * x[b, c, y1, x1] + (1 - dx) * dy * x[b, c, y2, x1] +
                                dx * (1 - dy) * x[b, c, y1, x2] + dx * dy * x[b, c, y2, x2])
                        s = f32(s + v)
                        k += 1
                    out[n, ctop, ph, pw] = 0 if k == 0 else f32(s / f32(k))
                    cnt[n, ctop, ph, pw] = k
    return out, cnt


def psroi_backward(dout, x, rois, trans, cnt, no_trans, scale, out_dim, gs, P, part, spp, trans_std):
    B, C, H, W = x.shape
    N = rois.shape[0]
    dx_ = np.zeros_like(x, dtype=np.float64)
    dtrans = np.zeros(trans.shape if not no_trans else (0,), np.float64)
    for n in range(N):
        for ctop in range(out_dim):
            for ph in range(P):
                for pw in range(P):
                    if cnt[n, ctop, ph, pw] <= 0:
                        continue
                    b, c, cls, part_h, part_w, ws, hs, sw_, sh_, roi_w, roi_h = _bins(
                        rois, trans, n, ctop, ph, pw, no_trans, scale, out_dim, gs, P, part, spp, trans_std)
                    diff = f32(dout[n, ctop, ph, pw] / cnt[n, ctop, ph, pw])
                    for w, h in _samples(ws, hs, sw_, sh_, spp, H, W):
                        x0, x1, y0, y1 = int(np.floor(w)), int(np.ceil(w)), int(np.floor(h)), int(np.ceil(h))
                        ddx, ddy = f32(w - x0), f32(h - y0)
                        dx_[b, c, y0, x0] += f32((1 - ddx) * (1 - ddy) * diff)
                        dx_[b, c, y1, x0] += f32((1 - ddx) * ddy * diff)
                        dx_[b, c, y0, x1] += f32(ddx * (1 - ddy) * diff)
                        dx_[b, c, y1, x1] += f32(ddx * ddy * diff)
                        if no_trans:
                            continue
                        u00, u01, u10, u11 = x[b, c, y0, x0], x[b, c, y1, x0], x[b, c, y0, x1], x[b, c, y1, x1]
                        gx = f32(f32(u11 * ddy + u10 * (1 - ddy) - u01 * ddy - u00 * (1 - ddy)) * f32(trans_std) * diff * roi_w)
                        gy = f32(f32(u11 * ddx + u01 * (1 - ddx) - u10 * ddx - u00 * (1 - ddx)) * f32(trans_std) * diff * roi_h)
                        dtrans[n, cls * 2, part_h, part_w] += gx
                        dtrans[n, cls * 2 + 1, part_h, part_w] += gy
    return dx_.astype(f32), dtrans.astype(f32)
