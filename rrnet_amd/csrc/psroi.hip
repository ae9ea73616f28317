// Deformable position-sensitive RoI pooling (DCNv2 pooling) for gfx950 — SURVEY §8 row f4.
//
// Replaces ext/dcn/src/cuda/dcn_v2_psroi_pooling_cuda.cu of the reference (DeformablePSROIPoolForwardKernel :59-153,
// DeformablePSROIPoolBackwardAccKernel :155-290), bound by ext/dcn/dcn_v2.py:130-300 (dcn_v2_pooling, DCNv2Pooling,
// DCNPooling).  No model of the reference calls it; built for API completeness of ext/dcn.
//   out[n, ctop, ph, pw] = mean over the sample_per_part^2 samples of bin (ph, pw) that fall inside the map of
//   bilinear(input[b, c], h, w),  c = (ctop*gs + gh)*gs + gw, the bin start shifted by trans[n, class, {x,y}, part] *
//   trans_std * roi size; RoI corners rounded to integers, (x2, y2) + 1, all scaled by spatial_scale, then - 0.5.
// NHWC: input [b,h,w,c], out / count [n, ph, pw, output_dim] (logical [n, output_dim, ph, pw]); trans NCHW
// [n, 2*num_classes, part, part] (tiny; the reference's layout).  One thread per (n, ph, pw, ctop): consecutive lanes
// are consecutive channels of the same bin, so with one class (the DCNPooling module) every bilinear corner is one
// coalesced row read.  Gather-bound; backward scatters with float atomics into zeroed gradients.
#include "common.h"
#include "rrnet_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct PsArgs {
    const float *x, *rois, *trans;
    int N, H, W, C, P, out_dim, gs, part, spp, no_trans, num_classes, ch_per_class;
    float scale, trans_std;
    int batch;     // images in x when known (backward): RoIs naming an image outside [0, batch) are skipped; 0 = only < 0 is
};

struct Bin {
    int b, c, cls, part_h, part_w;
    float wstart, hstart, sub_w, sub_h, roi_w, roi_h;
};

__device__ __forceinline__ Bin make_bin(const PsArgs &a, int n, int ctop, int ph, int pw)
{
    Bin r;
    const float *q = a.rois + (long)n * 5;
    r.b = (int)q[0];
    const float sw = roundf(q[1]) * a.scale - 0.5f, sh = roundf(q[2]) * a.scale - 0.5f;
    const float ew = (roundf(q[3]) + 1.f) * a.scale - 0.5f, eh = (roundf(q[4]) + 1.f) * a.scale - 0.5f;
    r.roi_w = fmaxf(ew - sw, 0.1f);
    r.roi_h = fmaxf(eh - sh, 0.1f);
    const float bin_h = r.roi_h / (float)a.P, bin_w = r.roi_w / (float)a.P;
    r.sub_h = bin_h / (float)a.spp;
    r.sub_w = bin_w / (float)a.spp;
    r.part_h = (int)floorf((float)ph / (float)a.P * (float)a.part);
    r.part_w = (int)floorf((float)pw / (float)a.P * (float)a.part);
    r.cls = ctop / a.ch_per_class;
    float tx = 0.f, ty = 0.f;
    if (!a.no_trans) {
        const long t0 = (((long)n * a.num_classes + r.cls) * 2) * a.part * a.part + (long)r.part_h * a.part + r.part_w;
        tx = a.trans[t0] * a.trans_std;
        ty = a.trans[t0 + (long)a.part * a.part] * a.trans_std;
    }
    r.wstart = (float)pw * bin_w + sw + tx * r.roi_w;
    r.hstart = (float)ph * bin_h + sh + ty * r.roi_h;
    int gw = (int)floorf((float)pw * (float)a.gs / (float)a.P), gh = (int)floorf((float)ph * (float)a.gs / (float)a.P);
    gw = min(max(gw, 0), a.gs - 1);
    gh = min(max(gh, 0), a.gs - 1);
    r.c = (ctop * a.gs + gh) * a.gs + gw;
    return r;
}

__global__ void psroi_fwd_kernel(const PsArgs a, float *out, float *count_out)
{
    const long total = (long)a.N * a.P * a.P * a.out_dim;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ctop = (int)(idx % a.out_dim);
        long r = idx / a.out_dim;
        const int pw = (int)(r % a.P); r /= a.P;
        const int ph = (int)(r % a.P);
        const int n = (int)(r / a.P);
        const Bin bn = make_bin(a, n, ctop, ph, pw);
        if (bn.b < 0 || (a.batch > 0 && bn.b >= a.batch)) {       // a RoI that names no image of the batch
            out[idx] = 0.f;
            count_out[idx] = 0.f;
            continue;
        }
        const float *img = a.x + (long)bn.b * a.H * a.W * a.C + bn.c;
        float sum = 0.f;
        int cnt = 0;
        for (int ih = 0; ih < a.spp; ++ih)
            for (int iw = 0; iw < a.spp; ++iw) {
                float w = bn.wstart + (float)iw * bn.sub_w, h = bn.hstart + (float)ih * bn.sub_h;
                if (w < -0.5f || w > (float)a.W - 0.5f || h < -0.5f || h > (float)a.H - 0.5f) continue;
                w = fminf(fmaxf(w, 0.f), (float)a.W - 1.f);
                h = fminf(fmaxf(h, 0.f), (float)a.H - 1.f);
                const int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
                const float dx = w - (float)x1, dy = h - (float)y1;
                const float v11 = img[((long)y1 * a.W + x1) * a.C], v12 = img[((long)y2 * a.W + x1) * a.C];
                const float v21 = img[((long)y1 * a.W + x2) * a.C], v22 = img[((long)y2 * a.W + x2) * a.C];
                sum += (1.f - dx) * (1.f - dy) * v11 + (1.f - dx) * dy * v12 + dx * (1.f - dy) * v21 + dx * dy * v22;
                ++cnt;
            }
        out[idx] = cnt == 0 ? 0.f : sum / (float)cnt;
        count_out[idx] = (float)cnt;
    }
}

__global__ void psroi_bwd_kernel(const PsArgs a, const float *dout, const float *count, float *dx_out, float *dtrans)
{
    const long total = (long)a.N * a.P * a.P * a.out_dim;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ctop = (int)(idx % a.out_dim);
        long r = idx / a.out_dim;
        const int pw = (int)(r % a.P); r /= a.P;
        const int ph = (int)(r % a.P);
        const int n = (int)(r / a.P);
        if (count[idx] <= 0.f) continue;
        const Bin bn = make_bin(a, n, ctop, ph, pw);
        if (bn.b < 0 || (a.batch > 0 && bn.b >= a.batch)) continue;
        const float diff = dout[idx] / count[idx];
        const long base = (long)bn.b * a.H * a.W * a.C + bn.c;
        float gx = 0.f, gy = 0.f;
        for (int ih = 0; ih < a.spp; ++ih)
            for (int iw = 0; iw < a.spp; ++iw) {
                float w = bn.wstart + (float)iw * bn.sub_w, h = bn.hstart + (float)ih * bn.sub_h;
                if (w < -0.5f || w > (float)a.W - 0.5f || h < -0.5f || h > (float)a.H - 0.5f) continue;
                w = fminf(fmaxf(w, 0.f), (float)a.W - 1.f);
                h = fminf(fmaxf(h, 0.f), (float)a.H - 1.f);
                const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
                const float dxx = w - (float)x0, dyy = h - (float)y0;
                const long o00 = base + ((long)y0 * a.W + x0) * a.C, o01 = base + ((long)y1 * a.W + x0) * a.C;
                const long o10 = base + ((long)y0 * a.W + x1) * a.C, o11 = base + ((long)y1 * a.W + x1) * a.C;
                unsafeAtomicAdd(dx_out + o00, (1.f - dxx) * (1.f - dyy) * diff);
                unsafeAtomicAdd(dx_out + o01, (1.f - dxx) * dyy * diff);
                unsafeAtomicAdd(dx_out + o10, dxx * (1.f - dyy) * diff);
                unsafeAtomicAdd(dx_out + o11, dxx * dyy * diff);
                if (a.no_trans) continue;
                const float u00 = a.x[o00], u01 = a.x[o01], u10 = a.x[o10], u11 = a.x[o11];
                gx += (u11 * dyy + u10 * (1.f - dyy) - u01 * dyy - u00 * (1.f - dyy)) * a.trans_std * diff * bn.roi_w;
                gy += (u11 * dxx + u01 * (1.f - dxx) - u10 * dxx - u00 * (1.f - dxx)) * a.trans_std * diff * bn.roi_h;
            }
        if (!a.no_trans) {
            const long t0 = (((long)n * a.num_classes + bn.cls) * 2) * a.part * a.part + (long)bn.part_h * a.part + bn.part_w;
            unsafeAtomicAdd(dtrans + t0, gx);
            unsafeAtomicAdd(dtrans + t0 + (long)a.part * a.part, gy);
        }
    }
}

int fill(PsArgs &a, const float *x, const float *rois, const float *trans, int n, int h, int w, int c, int no_trans,
         float scale, int out_dim, int gs, int pooled, int part, int spp, float trans_std, int trans_channels)
{
    RR_CHECK_ARG(n >= 0 && h > 0 && w > 0 && c > 0 && out_dim > 0 && gs > 0 && pooled > 0 && part > 0 && spp > 0,
                 "rr_dcn_psroi: bad dims");
    RR_CHECK_ARG(c == out_dim * gs * gs, "rr_dcn_psroi: input channels (%d) must equal output_dim * group_size^2 (%d)", c,
                 out_dim * gs * gs);
    RR_CHECK_ARG(no_trans || (trans && trans_channels >= 2 && trans_channels % 2 == 0 && out_dim % (trans_channels / 2) == 0),
                 "rr_dcn_psroi: trans must have 2*num_classes channels with num_classes dividing output_dim");
    a.x = x; a.rois = rois; a.trans = trans;
    a.N = n; a.H = h; a.W = w; a.C = c; a.P = pooled; a.out_dim = out_dim; a.gs = gs; a.part = part; a.spp = spp;
    a.no_trans = no_trans; a.scale = scale; a.trans_std = trans_std;
    a.num_classes = no_trans ? 1 : trans_channels / 2;
    a.ch_per_class = no_trans ? out_dim : out_dim / a.num_classes;
    return RR_OK;
}

}  // namespace

extern "C" int rr_dcn_psroi_fwd(const float *x, const float *rois, const float *trans, int n, int h, int w, int c,
                                int no_trans, float spatial_scale, int output_dim, int group_size, int pooled_size,
                                int part_size, int sample_per_part, float trans_std, int trans_channels, float *out,
                                float *count, hipStream_t stream)
{
    PsArgs a{};
    const int rc = fill(a, x, rois, trans, n, h, w, c, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
                        sample_per_part, trans_std, trans_channels);
    if (rc != RR_OK) return rc;
    const long total = (long)n * pooled_size * pooled_size * output_dim;
    if (total == 0) return RR_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(psroi_fwd_kernel, dim3((int)blocks), dim3(256), 0, stream, a, out, count);
    RR_CHECK_LAUNCH("rr_dcn_psroi_fwd");
    return RR_OK;
}

extern "C" int rr_dcn_psroi_bwd(const float *dout, const float *x, const float *rois, const float *trans,
                                const float *count, int n, int b, int h, int w, int c, int no_trans, float spatial_scale,
                                int output_dim, int group_size, int pooled_size, int part_size, int sample_per_part,
                                float trans_std, int trans_channels, float *dx, float *dtrans, hipStream_t stream)
{
    PsArgs a{};
    const int rc = fill(a, x, rois, trans, n, h, w, c, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
                        sample_per_part, trans_std, trans_channels);
    if (rc != RR_OK) return rc;
    RR_CHECK_ARG(b > 0, "rr_dcn_psroi_bwd: bad batch");
    RR_CHECK_ARG(no_trans || dtrans != nullptr, "rr_dcn_psroi_bwd: dtrans is required unless no_trans is set");
    a.batch = b;
    if (hipMemsetAsync(dx, 0, sizeof(float) * (size_t)b * h * w * c, stream) != hipSuccess) {
        rr_set_error("rr_dcn_psroi_bwd: clearing dx failed");
        return RR_ERR_LAUNCH;
    }
    if (!no_trans &&
        hipMemsetAsync(dtrans, 0, sizeof(float) * (size_t)n * trans_channels * part_size * part_size, stream) != hipSuccess) {
        rr_set_error("rr_dcn_psroi_bwd: clearing dtrans failed");
        return RR_ERR_LAUNCH;
    }
    const long total = (long)n * pooled_size * pooled_size * output_dim;
    if (total == 0) return RR_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(psroi_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, a, dout, count, dx, dtrans);
    RR_CHECK_LAUNCH("rr_dcn_psroi_bwd");
    return RR_OK;
}
