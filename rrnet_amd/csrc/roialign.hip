// RoIAlign on NHWC feature maps for gfx950.
//
// Replaces torchvision.ops.roi_align(relu(feat), rois, (3,3)) at models/rrnet.py:51 (torchvision
// is not vendored by the reference: parity unpinned; this follows the torchvision-0.3 definition:
// spatial_scale given, sampling_ratio -1 => ceil(roi_size/bins) samples per bin, RoI width/height
// clamped to >= 1, legacy (aligned=False) coordinates, samples outside [-1, size] contribute 0).
// One workgroup per RoI, one thread per channel: every bilinear tap is a coalesced C*4-byte row.
// Gather/HBM-bound: algorithmic bytes <= R*ph*pw*samples*4 taps*C*4 read, R*ph*pw*C*4 written;
// backward scatters the same taps with float atomics into a zeroed gradient map.
#include "common.h"
#include "rrnet_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct Tap { long o00, o01, o10, o11; float w00, w01, w10, w11; bool ok; };

__device__ __forceinline__ Tap make_tap(float y, float x, int H, int W, long C)
{
    Tap t;
    t.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (!t.ok) return t;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
    t.w00 = hy * hx; t.w01 = hy * lx; t.w10 = ly * hx; t.w11 = ly * lx;
    t.o00 = ((long)yl * W + xl) * C; t.o01 = ((long)yl * W + xh) * C;
    t.o10 = ((long)yh * W + xl) * C; t.o11 = ((long)yh * W + xh) * C;
    return t;
}

template <bool BWD>
__global__ __launch_bounds__(256) void roi_align_kernel(const float *feat, float *dfeat, const float *rois, float *out,
                                                        const float *dout, int H, int W, int C, int PH, int PW,
                                                        float scale, int sampling)
{
    const int r = blockIdx.x;
    const float *q = rois + (long)r * 5;
    const int b = (int)q[0];
    const float x1 = q[1] * scale, y1 = q[2] * scale, x2 = q[3] * scale, y2 = q[4] * scale;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bh = rh / (float)PH, bw = rw / (float)PW;
    const int gh = sampling > 0 ? sampling : (int)ceilf(rh / (float)PH);
    const int gw = sampling > 0 ? sampling : (int)ceilf(rw / (float)PW);
    const float count = (float)(gh * gw);
    const long img = (long)b * H * W * C;
    for (int bin = 0; bin < PH * PW; ++bin) {
        const int ph = bin / PW, pw = bin % PW;
        for (int c = threadIdx.x; c < C; c += 256) {
            float acc = 0.f;
            const float go = BWD ? dout[((long)r * PH * PW + bin) * C + c] / count : 0.f;
            for (int iy = 0; iy < gh; ++iy) {
                const float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                for (int ix = 0; ix < gw; ++ix) {
                    const float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                    const Tap t = make_tap(y, x, H, W, C);
                    if (!t.ok) continue;
                    if (!BWD) {
                        const float *f = feat + img + c;
                        acc += t.w00 * f[t.o00] + t.w01 * f[t.o01] + t.w10 * f[t.o10] + t.w11 * f[t.o11];
                    } else {
                        float *d = dfeat + img + c;
                        unsafeAtomicAdd(d + t.o00, go * t.w00);
                        unsafeAtomicAdd(d + t.o01, go * t.w01);
                        unsafeAtomicAdd(d + t.o10, go * t.w10);
                        unsafeAtomicAdd(d + t.o11, go * t.w11);
                    }
                }
            }
            if (!BWD) out[((long)r * PH * PW + bin) * C + c] = acc / count;
        }
    }
}
}  // namespace

extern "C" int rr_roi_align_fwd(const float *feat, const float *rois, int r, int h, int w, int c, int ph, int pw,
                                float spatial_scale, int sampling_ratio, float *out, hipStream_t stream)
{
    RR_CHECK_ARG(h > 0 && w > 0 && c > 0 && ph > 0 && pw > 0 && r >= 0, "rr_roi_align_fwd: bad dims");
    if (r == 0) return RR_OK;
    hipLaunchKernelGGL(roi_align_kernel<false>, dim3(r), dim3(256), 0, stream, feat, (float *)nullptr, rois, out,
                       (const float *)nullptr, h, w, c, ph, pw, spatial_scale, sampling_ratio);
    RR_CHECK_LAUNCH("rr_roi_align_fwd");
    return RR_OK;
}

extern "C" int rr_roi_align_bwd(const float *dout, const float *rois, int r, int b, int h, int w, int c, int ph, int pw,
                                float spatial_scale, int sampling_ratio, float *dfeat, hipStream_t stream)
{
    RR_CHECK_ARG(h > 0 && w > 0 && c > 0 && ph > 0 && pw > 0 && r >= 0 && b > 0, "rr_roi_align_bwd: bad dims");
    hipMemsetAsync(dfeat, 0, sizeof(float) * (size_t)b * h * w * c, stream);
    if (r == 0) return RR_OK;
    hipLaunchKernelGGL(roi_align_kernel<true>, dim3(r), dim3(256), 0, stream, (const float *)nullptr, dfeat, rois,
                       (float *)nullptr, dout, h, w, c, ph, pw, spatial_scale, sampling_ratio);
    RR_CHECK_LAUNCH("rr_roi_align_bwd");
    return RR_OK;
}
