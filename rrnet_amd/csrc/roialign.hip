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

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// ---- forward, separable form ---------------------------------------------------------------
// The samples of a bin form a product grid and the bilinear weights (and the inside-the-map test) factor into a
// row part and a column part, so  sum_{iy,ix} bilinear(y_iy, x_ix) = sum_r sum_c Wy[r] * Wx[c] * f[r][c]  with
// Wy[r] = total weight the bin's sample rows put on pixel row r.  Every footprint pixel is read once per bin
// ((gh+1)*(gw+1) rows of C floats) instead of 4 taps per sample (4*gh*gw): 2.25x fewer reads at a 3x3 grid.
// One wave per RoI, 16 bytes per lane (a 256-channel row is one 1 KB wave access), 4 RoIs per workgroup.
constexpr int SEP_MAXG = 16;           // samples per bin per axis handled here (larger RoIs: generic kernel)
constexpr int SEP_SLOTS = SEP_MAXG + 2;
constexpr int SEP_MAXBINS = 8;

struct AxisW {
    float w[SEP_MAXBINS][SEP_SLOTS];
    int base[SEP_MAXBINS], n[SEP_MAXBINS];
};

// weights of one bin along one axis: `start` = roi start + bin * bin_size, g samples, map extent `size`
__device__ void axis_weights(float start, float bin, int g, int size, float *w, int *base_out, int *n_out)
{
    for (int i = 0; i < SEP_SLOTS; ++i) w[i] = 0.f;
    int base = -1, last = -1;
    for (int i = 0; i < g; ++i) {
        float y = start + ((float)i + 0.5f) * bin / (float)g;
        if (y < -1.0f || y > (float)size) continue;
        if (y <= 0.f) y = 0.f;
        int yl = (int)y, yh;
        if (yl >= size - 1) { yh = yl = size - 1; y = (float)yl; } else yh = yl + 1;
        const float ly = y - yl, hy = 1.f - ly;
        if (base < 0) base = yl;
        w[yl - base] += hy;
        w[yh - base] += ly;
        last = yh - base;
    }
    *base_out = base < 0 ? 0 : base;
    *n_out = last + 1;
}

__global__ __launch_bounds__(256) void roi_align_sep_kernel(const float *feat, const float *rois, float *out, int R, int H,
                                                            int W, int C, int PH, int PW, float scale, int sampling)
{
    __shared__ AxisW ay[4], ax[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    const bool live = r < R;
    const float *q = rois + (long)(live ? r : 0) * 5;
    const int b = (int)q[0];
    const float x1 = q[1] * scale, y1 = q[2] * scale, x2 = q[3] * scale, y2 = q[4] * scale;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bh = rh / (float)PH, bw = rw / (float)PW;
    const int gh = sampling > 0 ? sampling : (int)ceilf(rh / (float)PH);
    const int gw = sampling > 0 ? sampling : (int)ceilf(rw / (float)PW);
    // wave-uniform; samples at most one pixel apart so that a bin's footprint fits the slot arrays (always true for
    // adaptive sampling, g = ceil(bin size))
    const bool sep = gh <= SEP_MAXG && gw <= SEP_MAXG && bh <= (float)gh && bw <= (float)gw;
    if (sep) {
        if (lane < PH) axis_weights(y1 + lane * bh, bh, gh, H, ay[wave].w[lane], &ay[wave].base[lane], &ay[wave].n[lane]);
        else if (lane < PH + PW) {
            const int p = lane - PH;
            axis_weights(x1 + p * bw, bw, gw, W, ax[wave].w[p], &ax[wave].base[p], &ax[wave].n[p]);
        }
    }
    __syncthreads();
    if (!live) return;
    const float count = (float)(gh * gw);
    const float *img = feat + (long)b * H * W * C;
    if (!sep) {   // very large RoI: per-sample taps, as the generic kernel
        for (int c0 = lane * 4; c0 < C; c0 += 256)
            for (int bin = 0; bin < PH * PW; ++bin) {
                const int ph = bin / PW, pw = bin % PW;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int iy = 0; iy < gh; ++iy) {
                    const float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                    for (int ix = 0; ix < gw; ++ix) {
                        const float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                        const Tap t = make_tap(y, x, H, W, C);
                        if (!t.ok) continue;
                        const float *f = img + c0;
                        acc += t.w00 * *reinterpret_cast<const f32x4 *>(f + t.o00) + t.w01 * *reinterpret_cast<const f32x4 *>(f + t.o01) +
                               t.w10 * *reinterpret_cast<const f32x4 *>(f + t.o10) + t.w11 * *reinterpret_cast<const f32x4 *>(f + t.o11);
                    }
                }
                *reinterpret_cast<f32x4 *>(out + ((long)r * PH * PW + bin) * C + c0) = acc / count;
            }
        return;
    }
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        for (int ph = 0; ph < PH; ++ph) {
            const int ny = ay[wave].n[ph], by = ay[wave].base[ph];
            for (int pw = 0; pw < PW; ++pw) {
                const int nx = ax[wave].n[pw], bx = ax[wave].base[pw];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int rr = 0; rr < ny; ++rr) {
                    const float wy = ay[wave].w[ph][rr];
                    const float *row = img + ((long)(by + rr) * W + bx) * C + c0;
                    for (int cc = 0; cc < nx; ++cc) {
                        const float wgt = wy * ax[wave].w[pw][cc];
                        acc += wgt * *reinterpret_cast<const f32x4 *>(row + (long)cc * C);
                    }
                }
                *reinterpret_cast<f32x4 *>(out + (((long)r * PH + ph) * PW + pw) * C + c0) = acc / count;
            }
        }
    }
}
// ---- forward, 3x3 bins, whole-footprint form ---------------------------------------------------
// RRNet's only use (models/rrnet.py:51: output (3,3)).  The three bins of an axis share their border pixels, so the
// RoI's footprint is walked ONCE: every pixel row is read a single time and scattered into the (at most 2 x 2) bins it
// belongs to — (3g+1)^2 row reads instead of 9(g+1)^2 (100 vs 144 at a 3x3 sampling grid).  Axis weights per bin over
// the footprint live in LDS; the bin loops are wave-uniform branches on "weight != 0".
constexpr int FP_MAX = 3 * (SEP_MAXG + 2);
constexpr int ROI_MLP = 8;    // footprint pixels (1 KB each per wave) requested before the first is used

struct AxisFP {
    float w[3][FP_MAX];
    int base, n;
};

// `order` (optional): processing position -> RoI index, a per-frame spatial sort (rr_roi_spatial_order), with the
// blocks mapped XCD-contiguously, so that RoIs whose footprints overlap run close in time on one XCD.  Measured at
// config 5: +0.5 % only — the kernel moves 17.5 GB through the fabric per 128 frames at 5.2 TB/s (PMC FETCH_SIZE), the
// 1024 RoIs an XCD has in flight touch ~90 MB, far beyond its 4 MB L2, so re-reads are served by the Infinity Cache
// either way.
__device__ __forceinline__ int roi_xcd_remap(int bid, int nb)
{
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__global__ __launch_bounds__(256) void roi_align_3x3_kernel(const float *feat, const float *rois, float *out, int R, int H, int W,
                                                            int C, float scale, int sampling, const int *order)
{
    __shared__ AxisFP ay[4], ax[4];
    __shared__ AxisW ty[4], tx[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pos = (order ? roi_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * 4 + wave;
    const bool live = pos < R;
    const int r = live ? (order ? order[pos] : pos) : 0;
    const float *q = rois + (long)(live ? r : 0) * 5;
    const int b = (int)q[0];
    const float x1 = q[1] * scale, y1 = q[2] * scale, x2 = q[3] * scale, y2 = q[4] * scale;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bh = rh / 3.f, bw = rw / 3.f;
    const int gh = sampling > 0 ? sampling : (int)ceilf(rh / 3.f);
    const int gw = sampling > 0 ? sampling : (int)ceilf(rw / 3.f);
    const bool sep = gh <= SEP_MAXG && gw <= SEP_MAXG && bh <= (float)gh && bw <= (float)gw;
    if (sep) {
        if (lane < 3) axis_weights(y1 + lane * bh, bh, gh, H, ty[wave].w[lane], &ty[wave].base[lane], &ty[wave].n[lane]);
        else if (lane < 6) {
            const int p = lane - 3;
            axis_weights(x1 + p * bw, bw, gw, W, tx[wave].w[p], &tx[wave].base[p], &tx[wave].n[p]);
        }
    }
    __syncthreads();
    if (sep && lane < 2) {       // merge the three bins of one axis into footprint-indexed weight rows
        AxisW &t = lane == 0 ? ty[wave] : tx[wave];
        AxisFP &f = lane == 0 ? ay[wave] : ax[wave];
        int lo = 1 << 30, hi = -1;
        for (int p = 0; p < 3; ++p)
            if (t.n[p] > 0) { lo = t.base[p] < lo ? t.base[p] : lo; hi = t.base[p] + t.n[p] > hi ? t.base[p] + t.n[p] : hi; }
        if (hi < 0) { lo = 0; hi = 0; }
        f.base = lo; f.n = hi - lo;
        for (int p = 0; p < 3; ++p) {
            for (int i = 0; i < f.n; ++i) f.w[p][i] = 0.f;
            for (int i = 0; i < t.n[p]; ++i) f.w[p][t.base[p] - lo + i] = t.w[p][i];
        }
    }
    __syncthreads();
    if (!live) return;
    const float count = (float)(gh * gw);
    const float *img = feat + (long)b * H * W * C;
    if (!sep) {   // very large RoI: per-sample taps
        for (int c0 = lane * 4; c0 < C; c0 += 256)
            for (int bin = 0; bin < 9; ++bin) {
                const int ph = bin / 3, pw = bin % 3;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int iy = 0; iy < gh; ++iy) {
                    const float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                    for (int ix = 0; ix < gw; ++ix) {
                        const float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                        const Tap t = make_tap(y, x, H, W, C);
                        if (!t.ok) continue;
                        const float *f = img + c0;
                        acc += t.w00 * *reinterpret_cast<const f32x4 *>(f + t.o00) + t.w01 * *reinterpret_cast<const f32x4 *>(f + t.o01) +
                               t.w10 * *reinterpret_cast<const f32x4 *>(f + t.o10) + t.w11 * *reinterpret_cast<const f32x4 *>(f + t.o11);
                    }
                }
                *reinterpret_cast<f32x4 *>(out + ((long)r * 9 + bin) * C + c0) = acc / count;
            }
        return;
    }
    const AxisFP &fy = ay[wave], &fx = ax[wave];
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        f32x4 acc[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int rr = 0; rr < fy.n; ++rr) {
            const float wy0 = fy.w[0][rr], wy1 = fy.w[1][rr], wy2 = fy.w[2][rr];
            const float *row = img + ((long)(fy.base + rr) * W + fx.base) * C + c0;
            // ROI_MLP pixels of the row are requested before the first is used: with one load in flight per wave (round 2)
            // a RoI took ~145 us (100 footprint pixels x ~1.45 us) and the kernel ran at 5.3 TB/s on occupancy alone;
            // 8 in flight: 2.73 ms instead of 3.43 per 128 frames at config 5 (4 / 12 / 16 in flight: 2.89 / 2.95 / 2.94 —
            // registers cost occupancy).  Capping the occupancy instead (to shorten the time between two overlapping RoIs'
            // reads of a shared line below the caches' residency) only loses: 4.7 ms at 3, 8.9 ms at 2 workgroups per CU.
            for (int cb = 0; cb < fx.n; cb += ROI_MLP) {
                f32x4 vv[ROI_MLP];
#pragma unroll
                for (int u = 0; u < ROI_MLP; ++u)
                    vv[u] = cb + u < fx.n ? *reinterpret_cast<const f32x4 *>(row + (long)(cb + u) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < ROI_MLP; ++u) {
                    const int cc = cb + u;
                    if (cc >= fx.n) break;
                    const f32x4 v = vv[u];
                    const float wx0 = fx.w[0][cc], wx1 = fx.w[1][cc], wx2 = fx.w[2][cc];
                    // wave-uniform: a pixel lies in at most two bins per axis
                    if (wy0 != 0.f) {
                        if (wx0 != 0.f) acc[0][0] += (wy0 * wx0) * v;
                        if (wx1 != 0.f) acc[0][1] += (wy0 * wx1) * v;
                        if (wx2 != 0.f) acc[0][2] += (wy0 * wx2) * v;
                    }
                    if (wy1 != 0.f) {
                        if (wx0 != 0.f) acc[1][0] += (wy1 * wx0) * v;
                        if (wx1 != 0.f) acc[1][1] += (wy1 * wx1) * v;
                        if (wx2 != 0.f) acc[1][2] += (wy1 * wx2) * v;
                    }
                    if (wy2 != 0.f) {
                        if (wx0 != 0.f) acc[2][0] += (wy2 * wx0) * v;
                        if (wx1 != 0.f) acc[2][1] += (wy2 * wx1) * v;
                        if (wx2 != 0.f) acc[2][2] += (wy2 * wx2) * v;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                *reinterpret_cast<f32x4 *>(out + (((long)r * 3 + i) * 3 + j) * C + c0) = acc[i][j] / count;
    }
}
// Per frame: RoIs sorted by (8-pixel row band, x centre) -> order[frame_off[f] + i] = RoI index.  One workgroup per
// frame, LDS bitonic sort of (key << 32 | local index); frames with more RoIs than the LDS sort holds keep their order.
constexpr int ORD_CAP = 4096;

__global__ __launch_bounds__(256) void roi_spatial_order_kernel(const float *rois, const int *frame_off, int *order)
{
    __shared__ unsigned long long keys[ORD_CAP];
    const int f = blockIdx.x;
    const int o0 = frame_off[f], n = frame_off[f + 1] - o0;
    if (n <= 0) return;
    if (n > ORD_CAP) {
        for (int i = threadIdx.x; i < n; i += 256) order[o0 + i] = o0 + i;
        return;
    }
    int np = 2;
    while (np < n) np <<= 1;
    for (int i = threadIdx.x; i < np; i += 256) {
        unsigned long long key = ~0ull;
        if (i < n) {
            const float *q = rois + (long)(o0 + i) * 5;
            const float cy = 0.5f * (q[2] + q[4]), cx = 0.5f * (q[1] + q[3]);
            const unsigned int band = (unsigned int)fminf(fmaxf(cy * 0.125f, 0.f), 65535.f);
            const unsigned int xq = (unsigned int)fminf(fmaxf(cx * 4.f, 0.f), 65535.f);
            key = ((unsigned long long)((band << 16) | xq) << 32) | (unsigned int)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= np; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < np / 2; t += 256) {
                const int lo = ((t / j) * 2 * j) + (t % j);
                const int hi = lo + j;
                const bool asc = ((lo & k2) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == asc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += 256) order[o0 + i] = o0 + (int)(keys[i] & 0xffffffffull);
}
}  // namespace

extern "C" int rr_roi_spatial_order(const float *rois, const int *frame_off, int nframes, int *order, hipStream_t stream)
{
    RR_CHECK_ARG(nframes >= 0, "rr_roi_spatial_order: bad dims");
    if (nframes == 0) return RR_OK;
    hipLaunchKernelGGL(roi_spatial_order_kernel, dim3(nframes), dim3(256), 0, stream, rois, frame_off, order);
    RR_CHECK_LAUNCH("rr_roi_spatial_order");
    return RR_OK;
}

extern "C" int rr_roi_align_fwd(const float *feat, const float *rois, int r, int h, int w, int c, int ph, int pw,
                                float spatial_scale, int sampling_ratio, const int *order, float *out, hipStream_t stream)
{
    RR_CHECK_ARG(h > 0 && w > 0 && c > 0 && ph > 0 && pw > 0 && r >= 0, "rr_roi_align_fwd: bad dims");
    if (r == 0) return RR_OK;
    if (c % 4 == 0 && ph == 3 && pw == 3) {
        hipLaunchKernelGGL(roi_align_3x3_kernel, dim3((r + 3) / 4), dim3(256), 0, stream, feat, rois, out, r, h, w, c,
                           spatial_scale, sampling_ratio, order);
        RR_CHECK_LAUNCH("rr_roi_align_fwd");
        return RR_OK;
    }
    if (c % 4 == 0 && ph <= SEP_MAXBINS && pw <= SEP_MAXBINS) {
        hipLaunchKernelGGL(roi_align_sep_kernel, dim3((r + 3) / 4), dim3(256), 0, stream, feat, rois, out, r, h, w, c, ph, pw,
                           spatial_scale, sampling_ratio);
        RR_CHECK_LAUNCH("rr_roi_align_fwd");
        return RR_OK;
    }
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
