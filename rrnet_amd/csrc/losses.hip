// Fused CenterNet losses for gfx950: one HBM pass + wavefront/block reductions.
//
// Replaces
//   clamp(sigmoid(x)) + focal_loss_for_hm   operators/rrnet_operator.py:55-57, modules/loss/functional.py:25-51
//   RegL1Loss                               modules/loss/regl1loss.py:9-17
//   stage-2 box_iou / targets / smooth-L1   operators/rrnet_operator.py:63-102
// (≈12 elementwise ATen kernels + 3 reductions per stack in the reference).
// Focal: algorithmic bytes = 2 tensors x 4 B x elements forward (x, gt), + 1 write backward.
#include "common.h"
#include "rrnet_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

constexpr int T = 256;

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void block_add3(double a, double b, double c, double *out)
{
    __shared__ double red[3][T / 64];
    a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int w = 0; w < T / 64; ++w) { s0 += red[0][w]; s1 += red[1][w]; s2 += red[2][w]; }
        unsafeAtomicAdd(out + 0, s0);
        unsafeAtomicAdd(out + 1, s1);
        unsafeAtomicAdd(out + 2, s2);
    }
}

// sums[0] = sum log(p)(1-p)^2 [g==1], sums[1] = sum log(1-p) p^2 (1-g)^4 [g<1], sums[2] = #(g==1)
__global__ __launch_bounds__(T) void focal_fwd_kernel(const float *x, const float *gt, long n, double *sums)
{
    double ps = 0.0, ns = 0.0, np = 0.0;
    for (long i = (long)blockIdx.x * T + threadIdx.x; i < n; i += (long)gridDim.x * T) {
        const float g = gt[i];
        float p = sigmoidf_ref(x[i]);
        p = fminf(fmaxf(p, 1e-4f), 1.0f - 1e-4f);
        if (g == 1.0f) {
            const float q = 1.0f - p;
            ps += (double)(logf(p) * (q * q));
            np += 1.0;
        } else if (g < 1.0f) {
            const float w = 1.0f - g;
            const float w2 = w * w;
            ns += (double)(logf(1.0f - p) * (p * p) * (w2 * w2));
        }
    }
    block_add3(ps, ns, np, sums);
}

// loss = -(ps + ns)/np  (np > 0)  |  -ns  (np == 0);   dx = gout * dloss/dx
__global__ __launch_bounds__(T) void focal_bwd_kernel(const float *x, const float *gt, long n, const double *sums,
                                                      const float *gout, float gscale, float *dx)
{
    const double np = sums[2];
    const float coef = -(*gout) * gscale * (np > 0.0 ? (float)(1.0 / np) : 1.0f);
    for (long i = (long)blockIdx.x * T + threadIdx.x; i < n; i += (long)gridDim.x * T) {
        const float g = gt[i];
        const float s = sigmoidf_ref(x[i]);
        float d = 0.f;
        if (s >= 1e-4f && s <= 1.0f - 1e-4f) {   // clamp passes the gradient inside [min, max]
            const float p = s, q = 1.0f - s;
            if (g == 1.0f) {
                d = (q * q) / p - 2.0f * q * logf(p);
            } else if (g < 1.0f) {
                const float w = 1.0f - g, w2 = w * w, w4 = w2 * w2;
                d = w4 * (2.0f * p * logf(q) - (p * p) / q);
            }
            d *= p * q;                           // dsigmoid/dx
        }
        dx[i] = coef * d;
    }
}

// ---- RegL1: pred NHWC [B,H*W,C], ind/mask [B,M] (float), target [B,M,C]
// sums[0] = sum |pred*m - t*m|, sums[1] = sum of the expanded mask
__global__ __launch_bounds__(T) void regl1_fwd_kernel(const float *pred, const float *mask, const float *ind,
                                                      const float *target, int B, int M, int C, long HW, double *sums)
{
    double ls = 0.0, ms = 0.0;
    const long n = (long)B * M * C;
    for (long i = (long)blockIdx.x * T + threadIdx.x; i < n; i += (long)gridDim.x * T) {
        const int c = (int)(i % C);
        const long bm = i / C;
        const int b = (int)(bm / M);
        const float m = mask[bm];
        const long pix = (long)ind[bm];
        const float p = pred[((long)b * HW + pix) * C + c];
        ls += (double)fabsf(p * m - target[i] * m);
        ms += (double)m;
    }
    block_add3(ls, ms, 0.0, sums);
}

__global__ __launch_bounds__(T) void regl1_bwd_kernel(const float *pred, const float *mask, const float *ind,
                                                      const float *target, int B, int M, int C, long HW,
                                                      const double *sums, const float *gout, float gscale, float *dpred)
{
    const float coef = (*gout) * gscale / ((float)sums[1] + 1e-4f);
    const long n = (long)B * M * C;
    for (long i = (long)blockIdx.x * T + threadIdx.x; i < n; i += (long)gridDim.x * T) {
        const int c = (int)(i % C);
        const long bm = i / C;
        const int b = (int)(bm / M);
        const float m = mask[bm];
        const long pix = (long)ind[bm];
        const long off = ((long)b * HW + pix) * C + c;
        const float d = pred[off] * m - target[i] * m;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        if (sg != 0.f && m != 0.f) unsafeAtomicAdd(dpred + off, coef * sg * m);
    }
}

// ---- stage 2: per-RoI max IoU against the image's ground truth, Faster-RCNN targets, smooth-L1
// rois [R,5] = (b, x1,y1,x2,y2) feature coords; gt [B,G,>=4] xyxy (already converted, image coords)
__global__ __launch_bounds__(T) void stage2_match_kernel(const float *rois, int R, const float *gt, int G, int gstride,
                                                         float scale, float *tgt /*[R,4]*/, int *pos /*[R]*/,
                                                         int *npos /*[B]*/)
{
    const int r = blockIdx.x * T + threadIdx.x;
    if (r >= R) return;
    const float *q = rois + (long)r * 5;
    const int b = (int)q[0];
    const float x1 = q[1] * scale, y1 = q[2] * scale, x2 = q[3] * scale, y2 = q[4] * scale;
    const float area = (x2 - x1) * (y2 - y1);
    float best = -__builtin_huge_valf();
    int bi = 0;
    for (int g = 0; g < G; ++g) {
        const float *t = gt + ((long)b * G + g) * gstride;
        const float ga = (t[2] - t[0]) * (t[3] - t[1]);
        const float w = fmaxf(fminf(x2, t[2]) - fmaxf(x1, t[0]), 0.f);
        const float h = fmaxf(fminf(y2, t[3]) - fmaxf(y1, t[1]), 0.f);
        const float inter = w * h;
        const float iou = inter / (area + ga - inter);
        if (iou > best) { best = iou; bi = g; }     // first maximum, as torch.max(dim=1)
    }
    const int p = best > 0.5f ? 1 : 0;
    pos[r] = p;
    if (p) {
        atomicAdd(npos + b, 1);
        const float *t = gt + ((long)b * G + bi) * gstride;
        const float ew = x2 - x1 + 1.0f, eh = y2 - y1 + 1.0f;
        const float ecx = x1 + 0.5f * ew, ecy = y1 + 0.5f * eh;
        const float gw = t[2] - t[0] + 1.0f, gh = t[3] - t[1] + 1.0f;
        const float gcx = t[0] + 0.5f * gw, gcy = t[1] + 0.5f * gh;
        float *o = tgt + (long)r * 4;
        o[0] = (gcx - ecx) / ew;
        o[1] = (gcy - ecy) / eh;
        o[2] = logf(gw / ew);
        o[3] = logf(gh / eh);
    }
}

// loss = sum_b [npos_b > 0] mean_{pos r of b, k} smoothl1(reg[r,k] - tgt[r,k]) / B ; grad likewise
__global__ __launch_bounds__(T) void stage2_loss_kernel(const float *rois, const float *reg, const float *tgt,
                                                        const int *pos, const int *npos, int R, int B, double *loss,
                                                        float *dreg_unit /*[R,4], d loss / d reg*/)
{
    double ls = 0.0;
    for (int i = blockIdx.x * T + threadIdx.x; i < R * 4; i += gridDim.x * T) {
        const int r = i >> 2;
        float g = 0.f;
        if (pos[r]) {
            const int b = (int)rois[(long)r * 5];
            const float inv = 1.0f / (4.0f * (float)npos[b]) / (float)B;
            const float d = reg[i] - tgt[i];
            const float ad = fabsf(d);
            const float l = ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
            ls += (double)(l * inv);
            g = (ad < 1.0f ? d : (d > 0.f ? 1.f : -1.f)) * inv;
        }
        dreg_unit[i] = g;
    }
    block_add3(ls, 0.0, 0.0, loss);
}

// d loss / d roi: the reference lets the smooth-L1 gradient flow into the TARGETS as well
// (rrnet_operator.py:82 with the hard-NMS-selected, still differentiable boxes of models/rrnet.py:70).
// tgt = ((gcx-ecx)/ew, (gcy-ecy)/eh, log(gw/ew), log(gh/eh)), e = roi*scale, ew = ex2-ex1+1, ecx = ex1 + ew/2.
__global__ __launch_bounds__(T) void stage2_droi_kernel(const float *rois, const float *tgt, const int *pos,
                                                        const float *dreg_unit, int R, float scale, float *droi)
{
    const int r = blockIdx.x * T + threadIdx.x;
    if (r >= R) return;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (pos[r]) {
        const float *q = rois + (long)r * 5;
        const float ew = (q[3] - q[1]) * scale + 1.0f, eh = (q[4] - q[2]) * scale + 1.0f;
        const float *t = tgt + (long)r * 4;
        const float *d = dreg_unit + (long)r * 4;
        const float u0 = -d[0], u1 = -d[1], u2 = -d[2], u3 = -d[3];     // dL/dtgt = -dL/dreg
        g0 = (u0 * (t[0] - 0.5f) + u2) / ew;      // d/dex1
        g2 = (u0 * (-t[0] - 0.5f) - u2) / ew;     // d/dex2
        g1 = (u1 * (t[1] - 0.5f) + u3) / eh;
        g3 = (u1 * (-t[1] - 0.5f) - u3) / eh;
    }
    float *o = droi + (long)r * 4;
    o[0] = g0 * scale; o[1] = g1 * scale; o[2] = g2 * scale; o[3] = g3 * scale;
}

}  // namespace

static inline int grid_for(long n) { long b = (n + T - 1) / T; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

extern "C" int rr_focal_loss_fwd(const float *logits, const float *gt, long n, double *sums, hipStream_t stream)
{
    hipMemsetAsync(sums, 0, 3 * sizeof(double), stream);
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(grid_for(n)), dim3(T), 0, stream, logits, gt, n, sums);
    RR_CHECK_LAUNCH("rr_focal_loss_fwd");
    return RR_OK;
}

extern "C" int rr_focal_loss_bwd(const float *logits, const float *gt, long n, const double *sums, const float *gout,
                                 float gscale, float *dlogits, hipStream_t stream)
{
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(grid_for(n)), dim3(T), 0, stream, logits, gt, n, sums, gout, gscale, dlogits);
    RR_CHECK_LAUNCH("rr_focal_loss_bwd");
    return RR_OK;
}

extern "C" int rr_regl1_fwd(const float *pred, const float *mask, const float *ind, const float *target, int b, int m,
                            int c, long hw, double *sums, hipStream_t stream)
{
    hipMemsetAsync(sums, 0, 3 * sizeof(double), stream);
    if ((long)b * m * c > 0)
        hipLaunchKernelGGL(regl1_fwd_kernel, dim3(grid_for((long)b * m * c)), dim3(T), 0, stream, pred, mask, ind, target,
                           b, m, c, hw, sums);
    RR_CHECK_LAUNCH("rr_regl1_fwd");
    return RR_OK;
}

extern "C" int rr_regl1_bwd(const float *pred, const float *mask, const float *ind, const float *target, int b, int m,
                            int c, long hw, const double *sums, const float *gout, float gscale, float *dpred,
                            hipStream_t stream)
{
    hipMemsetAsync(dpred, 0, sizeof(float) * (size_t)b * hw * c, stream);
    if ((long)b * m * c > 0)
        hipLaunchKernelGGL(regl1_bwd_kernel, dim3(grid_for((long)b * m * c)), dim3(T), 0, stream, pred, mask, ind, target,
                           b, m, c, hw, sums, gout, gscale, dpred);
    RR_CHECK_LAUNCH("rr_regl1_bwd");
    return RR_OK;
}

extern "C" int rr_stage2_loss(const float *rois, const float *reg, int r, const float *gt, int b, int g, int gstride,
                              float scale, float *tgt, int *pos, int *npos, double *loss, float *dreg_unit,
                              float *droi_unit, hipStream_t stream)
{
    RR_CHECK_ARG(gstride >= 4 && b > 0, "rr_stage2_loss: bad dims");
    hipMemsetAsync(npos, 0, sizeof(int) * b, stream);
    hipMemsetAsync(loss, 0, 3 * sizeof(double), stream);
    if (r > 0) {
        hipMemsetAsync(tgt, 0, sizeof(float) * 4 * (size_t)r, stream);
        hipLaunchKernelGGL(stage2_match_kernel, dim3(rr_cdiv(r, T)), dim3(T), 0, stream, rois, r, gt, g, gstride, scale, tgt,
                           pos, npos);
        hipLaunchKernelGGL(stage2_loss_kernel, dim3(grid_for((long)r * 4)), dim3(T), 0, stream, rois, reg, tgt, pos, npos, r,
                           b, loss, dreg_unit);
        if (droi_unit)
            hipLaunchKernelGGL(stage2_droi_kernel, dim3(rr_cdiv(r, T)), dim3(T), 0, stream, rois, tgt, pos, dreg_unit, r, scale,
                               droi_unit);
    }
    RR_CHECK_LAUNCH("rr_stage2_loss");
    return RR_OK;
}
