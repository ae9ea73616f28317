// CenterNet training targets on the device for gfx950.
//
// Replaces datasets/transforms/functional.py:177-262 of the reference (gaussian_radius, gaussian2d,
// draw_umich_gaussian and the heat-map transform :230-262: a Python loop over the boxes of every image on the host) and the padding of
// datasets/drones_det.py:70-94 (collate_fn_ctnet) for a whole batch in one launch: one 64-lane workgroup per
// (image, box) computes the box's regression targets and splats its Gaussian into the class plane with an integer
// atomicMax on the float bit patterns (all values are >= 0, so the orders agree; max is order-independent, the
// result is deterministic).  Quirks kept: the CornerNet radius formula divides by 2 instead of 2a; `ind` uses the
// hard-coded image_width // 4 (functional.py:257); sigma = diameter / 6; values below eps * max are dropped.
// Built with -ffp-contract=off: every step rounds like the reference's float32 torch / numpy ops, so wh / offset /
// ind / reg_mask / radius are bit-identical and the heat-map differs only by expf's last bit.
// Boxes whose centre falls outside the map are given regression targets but not drawn (the reference's slice
// arithmetic wraps around for them).
// HBM-bound: algorithmic bytes = the zero-fill of hm (B*C*Hf*Wf*4) + the (2r+1)^2 window per box.
#include "common.h"
#include "rrnet_hip.h"

namespace {

__device__ __forceinline__ float radius_of(float height, float width)
{
    // min_overlap = 0.7 enters as the float32 images of the reference's python-float expressions
    const float ov = 0.7f;
    const float one_minus = (float)(1.0 - 0.7), one_plus = (float)(1.0 + 0.7);
    const float b1 = height + width;
    const float c1 = width * height * one_minus / one_plus;
    const float r1 = (b1 + sqrtf(b1 * b1 - 4.0f * c1)) / 2.0f;
    const float b2 = 2.0f * (height + width);
    const float c2 = one_minus * width * height;
    const float r2 = (b2 + sqrtf(b2 * b2 - 16.0f * c2)) / 2.0f;
    const float a3 = (float)(4 * 0.7);
    const float b3 = (float)(-2 * 0.7) * (height + width);
    const float c3 = (float)(0.7 - 1) * width * height;
    const float r3 = (b3 + sqrtf(b3 * b3 - 4.0f * a3 * c3)) / 2.0f;
    (void)ov;
    return fminf(fminf(r1, r2), r3);
}

__global__ __launch_bounds__(64) void ctnet_targets_kernel(const float *annos, const int *counts, int M, int astride,
                                                           int img_w, int Hf, int Wf, int C, float scale, float *hm,
                                                           float *wh, float *ind, float *offset, float *mask)
{
    const int k = blockIdx.x, b = blockIdx.y;
    const long row = (long)b * M + k;
    if (k >= counts[b]) {                     // collate padding rows
        if (threadIdx.x == 0) {
            wh[row * 2] = wh[row * 2 + 1] = 0.f;
            offset[row * 2] = offset[row * 2 + 1] = 0.f;
            ind[row] = 0.f;
            mask[row] = 0.f;
        }
        return;
    }
    const float *a = annos + row * astride;
    const float x1 = a[0] / scale, y1 = a[1] / scale;
    const float x2 = (a[2] + a[0]) / scale, y2 = (a[3] + a[1]) / scale;
    const float bh = y2 - y1, bw = x2 - x1;
    const float cx = (x1 + x2) / 2.0f, cy = (y1 + y2) / 2.0f;
    const float cxi = floorf(cx), cyi = floorf(cy);
    if (threadIdx.x == 0) {
        wh[row * 2] = bw;
        wh[row * 2 + 1] = bh;
        offset[row * 2] = cx - cxi;
        offset[row * 2 + 1] = cy - cyi;
        ind[row] = cyi * (float)(img_w / 4) + cxi;
        mask[row] = (bh > 0.f && bw > 0.f) ? 1.f : 0.f;
    }
    const int cls = (int)(a[5] - 1.0f);
    float r = floorf(radius_of(ceilf(bh), ceilf(bw)));
    r = r > 0.f ? r : 0.f;                    // clamp(min=0); NaN (negative discriminant) also lands on 0
    if (!(r == r)) r = 0.f;
    if (cls < 0 || cls >= C || cxi < 0.f || cyi < 0.f || cxi >= (float)Wf || cyi >= (float)Hf) return;
    const float diameter = 2.0f * r + 1.0f;
    const float sigma = diameter / 6.0f;
    const float denom = (2.0f * sigma) * sigma;
    const int ri = (int)r, x = (int)cxi, y = (int)cyi;
    const int left = x < ri ? x : ri, right = (Wf - x) < (ri + 1) ? (Wf - x) : (ri + 1);
    const int top = y < ri ? y : ri, bottom = (Hf - y) < (ri + 1) ? (Hf - y) : (ri + 1);
    const int ww = left + right, hh = top + bottom;
    const float eps = 1.1920928955078125e-07f;        // np.finfo(float32).eps * h.max(), h.max() == 1
    for (int t = threadIdx.x; t < ww * hh; t += 64) {
        const int i = t / ww, j = t - i * ww;
        const float yy = (float)(i - top), xx = (float)(j - left);
        float v = expf(-(xx * xx + yy * yy) / denom);
        if (v < eps) v = 0.f;
        int *p = reinterpret_cast<int *>(hm + (((long)b * Hf + (y - top + i)) * Wf + (x - left + j)) * C + cls);
        atomicMax(p, __float_as_int(v));
    }
}

}  // namespace

extern "C" int rr_ctnet_targets(const float *annos, const int *counts, int b, int m, int anno_stride, int img_h, int img_w,
                                int scale_factor, int num_classes, float *hm, float *wh, float *ind, float *offset,
                                float *reg_mask, hipStream_t stream)
{
    RR_CHECK_ARG(b > 0 && m >= 0 && anno_stride >= 6 && img_h > 0 && img_w > 0 && scale_factor > 0 && num_classes > 0,
                 "rr_ctnet_targets: bad dims");
    const int hf = img_h / scale_factor, wf = img_w / scale_factor;
    hipMemsetAsync(hm, 0, sizeof(float) * (size_t)b * hf * wf * num_classes, stream);
    if (m == 0) return RR_OK;
    hipLaunchKernelGGL(ctnet_targets_kernel, dim3(m, b), dim3(64), 0, stream, annos, counts, m, anno_stride, img_w, hf, wf,
                       num_classes, (float)scale_factor, hm, wh, ind, offset, reg_mask);
    RR_CHECK_LAUNCH("rr_ctnet_targets");
    return RR_OK;
}
