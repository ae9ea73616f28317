// HBM-bound NHWC elementwise / per-channel-reduction kernels around the convolutions:
// BatchNorm (training statistics, apply, backward), ReLU, residual add, fan-out gradient sums,
// nearest-2x upsample + add, bilinear (align_corners) resize, bias gradients, fused Adam.
//
// Replaces what the reference runs as separate ATen kernels:
//   nn.BatchNorm2d / SyncBatchNorm + ReLU + residual add   backbones/hourglass.py:18-19,22,26,34-40,51,59-60
//   nn.Upsample(scale_factor=2) + bilinear resize + add     backbones/hourglass.py:113,121-124
//   optim.Adam                                              operators/rrnet_operator.py:29,138
// All kernels move 16 B per lane (float4 along the channel axis, C % 4 == 0) and are bound by
// HBM bandwidth; algorithmic bytes = 4 B x (tensors read + tensors written) x elements.
#include "common.h"
#include "rrnet_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// four fp32 -> four bf16 (round to nearest even, v_cvt_pk_bf16_f32; NaN stays NaN)
__device__ __forceinline__ u16x4 to_bf16x4(f32x4 v) { return __builtin_bit_cast(u16x4, __builtin_convertvector(v, bf16x4)); }
// four bf16 -> four fp32 (exact)
__device__ __forceinline__ f32x4 from_bf16x4(u16x4 v)
{
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(float, (unsigned)v[e] << 16);
    return r;
}

constexpr int EW_THREADS = 256;
inline int ew_blocks(long n4) { long b = (n4 + EW_THREADS - 1) / EW_THREADS; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

// ---- BatchNorm statistics ------------------------------------------------------------------
// slab [mtiles][2][C] (doubles, written by rr_conv_fprop) -> sums [2][C].
// block = 32 columns x 8 row lanes; grid.y slices the mtiles; one double atomic per column per block
// (sums is zeroed by the caller).
__global__ __launch_bounds__(256) void bn_reduce_slab_kernel(const double *slab, int mtiles, int C2, int rows_per_block,
                                                             double *sums, double count = 0.0, double *count_slot = nullptr)
{
    // (SyncBN: the local sample count travels with the sums through the all-reduce — written here instead of by a fill launch)
    if (count_slot != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *count_slot = count;
    __shared__ double red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + tx;
    const int m0 = blockIdx.y * rows_per_block;
    int m1 = m0 + rows_per_block;
    if (m1 > mtiles) m1 = mtiles;
    double s = 0.0;
    if (col < C2)
        for (int m = m0 + ty; m < m1; m += 8) s += slab[(long)m * C2 + col];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && col < C2) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][tx];
        unsafeAtomicAdd(sums + col, t);
    }
}

// sums [2][C] over `count` samples -> mean / invstd / scale / shift (+ running stats, momentum update
// with the unbiased variance, as nn.BatchNorm2d does in training mode).
__global__ void bn_finalize_kernel(const double *sums, double count_h, const double *count_d, const float *gamma, const float *beta,
                                   float *running_mean, float *running_var, float momentum, float eps,
                                   float *mean, float *invstd, float *scale, float *shift, int C, long *num_batches_tracked,
                                   double *count_out = nullptr)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;   // nn.BatchNorm2d's counter, without its own launch
    const double count = count_d ? *count_d : count_h;   // device count: SyncBN with ragged per-rank sample counts
    if (c == 0 && count_out) *count_out = count;         // (kept for the backward in storage of its own: no clone launch)
    if (c >= C) return;
    const double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = istd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * istd;
    scale[c] = sc;
    shift[c] = b - (float)m * sc;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// Single-process training (no SyncBN exchange between the two): slab -> sums -> finalize in ONE launch.  A workgroup of
// 32 channels x 32 row lanes reduces both statistics of its channels over the mtiles rows in a fixed order (no atomics:
// bit-reproducible, unlike bn_reduce_slab_kernel) and finalizes them.  163 launches per step less.
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const double *slab, int mtiles, double count, const float *gamma,
                                                                 const float *beta, float *running_mean, float *running_var,
                                                                 float momentum, float eps, float *mean, float *invstd,
                                                                 float *scale, float *shift, int C, long *num_batches_tracked)
{
    __shared__ double red[2][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int m = ty; m < mtiles; m += 32) {
            s1 += slab[(long)m * 2 * C + c];
            s2 += slab[(long)m * 2 * C + C + c];
        }
    red[0][ty][tx] = s1;
    red[1][ty][tx] = s2;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (ty != 0 || c >= C) return;
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) { t1 += red[0][k][tx]; t2 += red[1][k][tx]; }
    const double m = t1 / count;
    double var = t2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = istd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * istd;
    scale[c] = sc;
    shift[c] = b - (float)m * sc;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_coeffs_kernel(const float *gamma, const float *beta, const float *running_mean,
                                      const float *running_var, float eps, float *scale, float *shift, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(running_var[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - running_mean[c] * sc;
}

// out = relu?( y*scale+shift  [+ res | + res*res_scale+res_shift] )
// AMAX: also max |out| -> *amax as a bit pattern (one atomicMax per wave) — the next convolution's operand scale when it
// runs on the split-operand kernels (rr_conv_*_f16x3), saved a pass of rr_absmax_bits over the tensor
template <bool AMAX>
__global__ __launch_bounds__(EW_THREADS) void bn_apply_kernel(const f32x4 *y, const float *scale, const float *shift,
                                                              const f32x4 *res, const float *res_scale,
                                                              const float *res_shift, f32x4 *out, long n4, int C4,
                                                              int relu, unsigned *amax, u16x4 *out16 = nullptr,
                                                              const u16x4 *res16 = nullptr, const u16x4 *y16 = nullptr)
{
    float vmax = 0.f;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C4) * 4;
        f32x4 v = y16 ? from_bf16x4(y16[i]) : y[i];          // (y16: the convolution's output exists only as its bf16 image)
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
        v = rr_bn_affine4(v, sc, sh);
        if (res || res16) {
            f32x4 r = res16 ? from_bf16x4(res16[i]) : res[i];     // (res16: the residual exists only as its bf16 image)
            if (res_scale) {
                const f32x4 rs = *reinterpret_cast<const f32x4 *>(res_scale + c);
                const f32x4 rh = *reinterpret_cast<const f32x4 *>(res_shift + c);
                r = r * rs + rh;
            }
            v = v + r;
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (out) out[i] = v;
        if (out16) out16[i] = to_bf16x4(v);      // the bf16 image the next convolution reads (csrc/conv16.hip)
        if constexpr (AMAX) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if constexpr (AMAX) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        // (look first: once the word holds the tensor's maximum, or nearly, almost every wave skips the same-address atomic —
        // 16 k of them per launch serialise at ~15 ns each)
        if ((threadIdx.x & 63) == 0 && __builtin_bit_cast(unsigned, vmax) > *(volatile unsigned *)amax) atomicMax(amax, __builtin_bit_cast(unsigned, vmax));
    }
}

// per-channel sums of dy and dy*xhat, dy = dz * (z > 0 if z given).  Each workgroup walks a slice of
// the pixels with a fixed channel quad per thread, reduces over its pixel lanes in LDS and adds one
// double per channel per workgroup to sums[2][C].
template <bool IMG>      // IMG: y (and z, when given) come as bf16 images — the batched-load main loop below
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_reduce_kernel(const float *dz, const float *z, const float *y,
                                                                   const float *mean, const float *invstd,
                                                                   const float *mscale, const float *mshift,
                                                                   double *sums, long npix, int C, const unsigned short *z16,
                                                                   const unsigned short *y16)
{
    __shared__ double red[2][EW_THREADS * 4];
    const int C4 = C / 4;                      // <= EW_THREADS (checked by the launcher)
    const int lanes = EW_THREADS / C4;         // pixel lanes per pass
    const int t = threadIdx.x;
    const int cq = t % C4, pl = t / C4;
    // fp32 partial sums over at most 8 pixels, then double: the reduction runs over up to 10^6
    // samples per channel and its terms cancel (torch-CPU accumulates in double as well)
    double d1[4] = {0.0, 0.0, 0.0, 0.0}, d2[4] = {0.0, 0.0, 0.0, 0.0};
    if (pl < lanes) {
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + cq * 4);
        const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + cq * 4);
        f32x4 msc = {0.f, 0.f, 0.f, 0.f}, msh = msc;
        if (mscale) { msc = *reinterpret_cast<const f32x4 *>(mscale + cq * 4); msh = *reinterpret_cast<const f32x4 *>(mshift + cq * 4); }
        const long step = (long)gridDim.x * lanes;
        for (long p0 = (long)blockIdx.x * lanes + pl; p0 < npix; p0 += step * 8) {
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
            if (IMG && p0 + 7 * step < npix) {
                // all eight pixels exist: every load first, then the arithmetic.  With the bounds test around each pixel the
                // compiler kept ONE pixel's loads in flight per wave — enough for 16-byte fp32 loads (5.1 TB/s), not for the
                // 8-byte image loads of the bf16-only tensors (same 0.31 ms for two thirds of the bytes; round 5)
                f32x4 gv[8];
                u16x4 yv[8], zv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long off = (p0 + u * step) * C + cq * 4;
                    gv[u] = *reinterpret_cast<const f32x4 *>(dz + off);
                    yv[u] = *reinterpret_cast<const u16x4 *>(y16 + off);
                    if (z16) zv[u] = *reinterpret_cast<const u16x4 *>(z16 + off);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    f32x4 g = gv[u];
                    const f32x4 yy = from_bf16x4(yv[u]);
                    if (z16) {
                        const f32x4 zz = from_bf16x4(zv[u]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
                    } else if (mscale) {
                        const f32x4 zz = rr_bn_affine4(yy, msc, msh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
                    }
                    const f32x4 xh = (yy - mu) * is;
                    s1 += g;
                    s2 += g * xh;
                }
            } else
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long p = p0 + u * step;
                if (p < npix) {
                    const long off = p * C + cq * 4;
                    f32x4 g = *reinterpret_cast<const f32x4 *>(dz + off);
                    const f32x4 yy = y16 ? from_bf16x4(*reinterpret_cast<const u16x4 *>(y16 + off)) : *reinterpret_cast<const f32x4 *>(y + off);
                    if (z || z16) {
                        const f32x4 zz = z16 ? from_bf16x4(*reinterpret_cast<const u16x4 *>(z16 + off)) : *reinterpret_cast<const f32x4 *>(z + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
                    } else if (mscale) {      // ReLU mask recomputed from y: z = relu(y*scale+shift), no residual
                        const f32x4 zz = rr_bn_affine4(yy, msc, msh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
                    }
                    const f32x4 xh = (yy - mu) * is;
                    s1 += g;
                    s2 += g * xh;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { d1[e] += (double)s1[e]; d2[e] += (double)s2[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][t * 4 + e] = d1[e]; red[1][t * 4 + e] = d2[e]; }
    __syncthreads();
    for (int c = t; c < C; c += EW_THREADS) {
        double a = 0.0, b = 0.0;
        for (int l = 0; l < lanes; ++l) {
            a += red[0][(l * C4 + c / 4) * 4 + (c & 3)];
            b += red[1][(l * C4 + c / 4) * 4 + (c & 3)];
        }
        unsafeAtomicAdd(sums + c, a);
        unsafeAtomicAdd(sums + C + c, b);
    }
}

// dx = gamma*invstd*(dy - sum_dy/count - xhat*sum_dy_xhat/count); optional g_out = dy (the masked
// gradient, for the residual branch); workgroup 0 also accumulates dgamma / dbeta.
// AMAX: also max |dx| -> *amax (bit pattern): dx is the output gradient of the convolution in front of this BatchNorm, i.e.
// an operand of its data and weight gradients (rr_conv_*_f16x3 scale their operands by the tensor's maximum)
// HOIST (the bf16-image launches): per-channel constants loaded once per thread when the grid stride allows — the fp32 launches
// keep the plain loop (with the constants in registers they lost a fifth of their rate: 0.42 -> 0.51 ms at the 256^2 layer)
template <bool AMAX, bool HOIST = false>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_apply_kernel(const f32x4 *dz, const f32x4 *z, const f32x4 *y,
                                                                  const float *mean, const float *invstd,
                                                                  const float *gamma, const float *mscale, const float *mshift,
                                                                  const double *sums, double count_h,
                                                                  const double *count_d, f32x4 *dx, f32x4 *g_out,
                                                                  float *dgamma, float *dbeta, long n4, int C, int g_acc,
                                                                  unsigned *amax, u16x4 *dx16 = nullptr, const u16x4 *z16 = nullptr,
                                                                  const u16x4 *y16 = nullptr)
{
    float vmax = 0.f;
    const int C4 = C / 4;
    const double count = count_d ? *count_d : count_h;
    if (blockIdx.x == 0 && dgamma) {
        for (int c = threadIdx.x; c < C; c += EW_THREADS) {
            dbeta[c] += (float)sums[c];
            dgamma[c] += (float)sums[C + c];
        }
    }
    const float inv_count = (float)(1.0 / count);
    if constexpr (HOIST) {
    // the per-channel constants of a channel quad: a = gamma * invstd, and dx = a * (g - sdy - xhat * sdx)
    struct Quad { f32x4 mu, is, a, sdy, sdx, msc, msh; };
    auto quad_of = [&](int c) {
        Quad q;
        q.mu = *reinterpret_cast<const f32x4 *>(mean + c);
        q.is = *reinterpret_cast<const f32x4 *>(invstd + c);
        q.a = *reinterpret_cast<const f32x4 *>(gamma + c) * q.is;
#pragma unroll
        for (int e = 0; e < 4; ++e) { q.sdy[e] = (float)sums[c + e] * inv_count; q.sdx[e] = (float)sums[C + c + e] * inv_count; }
        q.msc = q.msh = f32x4{0.f, 0.f, 0.f, 0.f};
        if (mscale) { q.msc = *reinterpret_cast<const f32x4 *>(mscale + c); q.msh = *reinterpret_cast<const f32x4 *>(mshift + c); }
        return q;
    };
    auto one = [&](long i, const Quad &q) {
        f32x4 g = dz[i];
        const f32x4 yy = y16 ? from_bf16x4(y16[i]) : y[i];
        if (z || z16) {
            const f32x4 zz = z16 ? from_bf16x4(z16[i]) : z[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        } else if (mscale) {
            const f32x4 zz = rr_bn_affine4(yy, q.msc, q.msh);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        }
        if (g_out) g_out[i] = g_acc ? g_out[i] + g : g;     // g_acc: the residual branch's fan-in buffer already holds a gradient
        const f32x4 xh = (yy - q.mu) * q.is;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = q.a[e] * (g[e] - q.sdy[e] - xh[e] * q.sdx[e]);
        if (dx) dx[i] = o;
        if (dx16) dx16[i] = to_bf16x4(o);        // the bf16 image the data / weight gradient read (csrc/conv16.hip)
        if constexpr (AMAX) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    };
    const long stride = (long)gridDim.x * EW_THREADS;
    const long first = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    if (stride % C4 == 0) {
        // the grid stride is a multiple of the channel quads: a thread stays on ONE quad — its constants (eight doubles converted,
        // five vector loads) are loaded once instead of per element (round 5)
        const Quad q = quad_of((int)(first % C4) * 4);
        for (long i = first; i < n4; i += stride) one(i, q);
    } else {
        for (long i = first; i < n4; i += stride) one(i, quad_of((int)(i % C4) * 4));
    }
    } else {
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C4) * 4;
        f32x4 g = dz[i];
        const f32x4 yy = y16 ? from_bf16x4(y16[i]) : y[i];
        if (z || z16) {
            const f32x4 zz = z16 ? from_bf16x4(z16[i]) : z[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        } else if (mscale) {
            const f32x4 zz = rr_bn_affine4(yy, *reinterpret_cast<const f32x4 *>(mscale + c), *reinterpret_cast<const f32x4 *>(mshift + c));
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        }
        if (g_out) g_out[i] = g_acc ? g_out[i] + g : g;     // g_acc: the residual branch's fan-in buffer already holds a gradient
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + c), is = *reinterpret_cast<const f32x4 *>(invstd + c);
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + c);
        const f32x4 xh = (yy - mu) * is;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sdy = (float)sums[c + e] * inv_count, sdx = (float)sums[C + c + e] * inv_count;
            o[e] = ga[e] * is[e] * (g[e] - sdy - xh[e] * sdx);
        }
        if (dx) dx[i] = o;
        if (dx16) dx16[i] = to_bf16x4(o);        // the bf16 image the data / weight gradient read (csrc/conv16.hip)
        if constexpr (AMAX) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    }
    if constexpr (AMAX) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        // (look first: once the word holds the tensor's maximum, or nearly, almost every wave skips the same-address atomic —
        // 16 k of them per launch serialise at ~15 ns each)
        if ((threadIdx.x & 63) == 0 && __builtin_bit_cast(unsigned, vmax) > *(volatile unsigned *)amax) atomicMax(amax, __builtin_bit_cast(unsigned, vmax));
    }
}

// ---- channel padding -----------------------------------------------------------------------
// [npix][k] -> [npix][kp] (kp = k rounded up to 4, the tail zero): one float4 of the output per thread
__global__ __launch_bounds__(EW_THREADS) void pad_channels_kernel(const float *src, f32x4 *dst, long n4, int k, int kp4)
{
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
        const long pix = i / kp4;
        const int c0 = (int)(i - pix * kp4) * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c0 + e < k ? src[pix * k + c0 + e] : 0.f;
        dst[i] = v;
    }
}

// ---- ReLU / sums ---------------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void relu_fwd_kernel(const f32x4 *x, f32x4 *out, long n4)
{
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
        f32x4 v = x[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        out[i] = v;
    }
}

// out = (g0 + g1 + ... ) * (z > 0 if z): gradient fan-in of a tensor with several consumers
struct SumArgs { const f32x4 *g[8]; int n; };
__global__ __launch_bounds__(EW_THREADS) void sum_n_kernel(SumArgs a, const f32x4 *z, f32x4 *out, long n4)
{
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
        f32x4 v = a.g[0][i];
        for (int k = 1; k < a.n; ++k) v += a.g[k][i];
        if (z) {
            const f32x4 zz = z[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = zz[e] > 0.f ? v[e] : 0.f;
        }
        out[i] = v;
    }
}

// dx = dz * (z > 0); dbias[c] += column sums (head convs: conv + bias + ReLU)
__global__ __launch_bounds__(EW_THREADS) void colsum_kernel(const float *dy, const float *z, float *dy_masked,
                                                            float *dbias, long npix, int C)
{
    // generic C (heads have C = 10, 2, 1, 4, 256): thread = channel (strided), block walks pixel slices
    __shared__ float red[EW_THREADS];
    const int t = threadIdx.x;
    const int cpb = C < EW_THREADS ? C : EW_THREADS;          // channels per pass
    const int lanes = EW_THREADS / cpb;                        // pixel lanes
    for (int c0 = 0; c0 < C; c0 += cpb) {
        const int c = c0 + t % cpb, pl = t / cpb;
        float s = 0.f;
        if (c < C && pl < lanes) {
            for (long p = (long)blockIdx.x * lanes + pl; p < npix; p += (long)gridDim.x * lanes) {
                float g = dy[p * C + c];
                if (z) {
                    g = z[p * C + c] > 0.f ? g : 0.f;
                    if (dy_masked) dy_masked[p * C + c] = g;
                }
                s += g;
            }
        }
        red[t] = s;
        __syncthreads();
        if (t < cpb && c0 + t < C) {
            float tot = 0.f;
            for (int l = 0; l < lanes; ++l) tot += red[l * cpb + t];
            if (dbias) unsafeAtomicAdd(dbias + c0 + t, tot);
        }
        __syncthreads();
    }
}

// 16 bytes per lane for the wide heads (C % 4 == 0 and C/4 divides 256): thread = channel quad, the block's 256 / (C/4)
// pixel lanes walk the pixels with two rows in flight; same sums as colsum_kernel up to the summation order.
__global__ __launch_bounds__(EW_THREADS) void colsum4_kernel(const f32x4 *dy, const f32x4 *z, f32x4 *dy_masked, float *dbias,
                                                             long npix, int C4)
{
    __shared__ f32x4 red[EW_THREADS];
    const int t = threadIdx.x;
    const int c4 = t % C4, pl = t / C4, lanes = EW_THREADS / C4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    const long step = (long)gridDim.x * lanes;
    long p = (long)blockIdx.x * lanes + pl;
    for (; p + step < npix; p += 2 * step) {
        f32x4 g0 = dy[p * C4 + c4], g1 = dy[(p + step) * C4 + c4];
        if (z) {
            const f32x4 z0 = z[p * C4 + c4], z1 = z[(p + step) * C4 + c4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { g0[i] = z0[i] > 0.f ? g0[i] : 0.f; g1[i] = z1[i] > 0.f ? g1[i] : 0.f; }
            if (dy_masked) { dy_masked[p * C4 + c4] = g0; dy_masked[(p + step) * C4 + c4] = g1; }
        }
        s0 += g0; s1 += g1;
    }
    if (p < npix) {
        f32x4 g0 = dy[p * C4 + c4];
        if (z) {
            const f32x4 z0 = z[p * C4 + c4];
#pragma unroll
            for (int i = 0; i < 4; ++i) g0[i] = z0[i] > 0.f ? g0[i] : 0.f;
            if (dy_masked) dy_masked[p * C4 + c4] = g0;
        }
        s0 += g0;
    }
    red[t] = s0 + s1;
    __syncthreads();
    if (t < C4 && dbias) {
        f32x4 tot = {0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < lanes; ++l) tot += red[l * C4 + t];
#pragma unroll
        for (int i = 0; i < 4; ++i) unsafeAtomicAdd(dbias + 4 * t + i, tot[i]);
    }
}

// ---- hourglass up path ---------------------------------------------------------------------
// out[n,h,w,:] = up1[n,h,w,:] + low[n,h/2,w/2,:]     (nn.Upsample(scale_factor=2), nearest; the
// bilinear align_corners resize that follows in the reference is the identity at equal sizes)
__global__ __launch_bounds__(EW_THREADS) void upsample2x_add_kernel(const f32x4 *up1, const f32x4 *low, f32x4 *out, int N, int H, int W,
                                                                    int C4, const u16x4 *up1_16 = nullptr, const u16x4 *low_16 = nullptr,
                                                                    u16x4 *out16 = nullptr)
{
    // up1_16 / low_16: the operand exists only as its bf16 image (conv16 activations); out may be null when out16 is given
    const int LH = H / 2, LW = W / 2;
    const long n4 = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const long li = (((long)n * LH + h / 2) * LW + w / 2) * C4 + c;
        const f32x4 a = up1_16 ? from_bf16x4(up1_16[i]) : up1[i];
        const f32x4 b = low_16 ? from_bf16x4(low_16[li]) : low[li];
        const f32x4 v = a + b;
        if (out) out[i] = v;
        if (out16) out16[i] = to_bf16x4(v);
    }
}

// dlow[n,y,x,:] = sum of the 2x2 block of dout
__global__ __launch_bounds__(EW_THREADS) void upsample2x_bwd_kernel(const f32x4 *dout, f32x4 *dlow, int N, int H, int W,
                                                                    int C4)
{
    const int LH = H / 2, LW = W / 2;
    const long n4 = (long)N * LH * LW * C4;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int x = (int)(p % LW); p /= LW;
        const int y = (int)(p % LH);
        const int n = (int)(p / LH);
        const long b = (((long)n * H + 2 * y) * W + 2 * x) * C4 + c;
        dlow[i] = dout[b] + dout[b + C4] + dout[b + (long)W * C4] + dout[b + (long)W * C4 + C4];
    }
}

// General path (odd feature sizes at eval scales): out = up1 + bilinear_align_corners(nearest2x(low)).
// The nearest-2x image is never materialised: sample (yy, xx) of it is low[yy/2, xx/2].
__global__ __launch_bounds__(EW_THREADS) void upsample_bilinear_add_kernel(const float *up1, const float *low, float *out,
                                                                           int N, int H, int W, int LH, int LW, int C)
{
    const long total = (long)N * H * W * C;
    const int UH = 2 * LH, UW = 2 * LW;
    const float sy = H > 1 ? (float)(UH - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(UW - 1) / (float)(W - 1) : 0.f;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C);
        long p = i / C;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const float fy = sy * h, fx = sx * w;
        int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < UH - 1 ? 1 : 0), x1 = x0 + (x0 < UW - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const float *b = low + (long)n * LH * LW * C + c;
        const float v00 = b[((long)(y0 / 2) * LW + x0 / 2) * C], v01 = b[((long)(y0 / 2) * LW + x1 / 2) * C];
        const float v10 = b[((long)(y1 / 2) * LW + x0 / 2) * C], v11 = b[((long)(y1 / 2) * LW + x1 / 2) * C];
        out[i] = up1[i] + (hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11));
    }
}

// Stand-alone bilinear resize, align_corners = True (multi-scale evaluation: operators/rrnet_operator.py:263
// F.interpolate(img, scale_factor=s, mode='bilinear', align_corners=True)); NHWC, any C.
__global__ __launch_bounds__(EW_THREADS) void resize_bilinear_ac_kernel(const float *x, float *out, int N, int H, int W, int OH,
                                                                        int OW, int C)
{
    const long total = (long)N * OH * OW * C;
    const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C);
        long p = i / C;
        const int w = (int)(p % OW); p /= OW;
        const int h = (int)(p % OH);
        const int n = (int)(p / OH);
        const float fy = sy * h, fx = sx * w;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const float *b = x + (long)n * H * W * C + c;
        const float v00 = b[((long)y0 * W + x0) * C], v01 = b[((long)y0 * W + x1) * C];
        const float v10 = b[((long)y1 * W + x0) * C], v11 = b[((long)y1 * W + x1) * C];
        out[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}

__global__ __launch_bounds__(EW_THREADS) void upsample_bilinear_bwd_kernel(const float *dout, float *dlow, int N, int H, int W,
                                                                           int LH, int LW, int C)
{
    const long total = (long)N * H * W * C;
    const int UH = 2 * LH, UW = 2 * LW;
    const float sy = H > 1 ? (float)(UH - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(UW - 1) / (float)(W - 1) : 0.f;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C);
        long p = i / C;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const float fy = sy * h, fx = sx * w;
        int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < UH - 1 ? 1 : 0), x1 = x0 + (x0 < UW - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        float *b = dlow + (long)n * LH * LW * C + c;
        const float g = dout[i];
        unsafeAtomicAdd(b + ((long)(y0 / 2) * LW + x0 / 2) * C, g * hy * hx);
        unsafeAtomicAdd(b + ((long)(y0 / 2) * LW + x1 / 2) * C, g * hy * lx);
        unsafeAtomicAdd(b + ((long)(y1 / 2) * LW + x0 / 2) * C, g * ly * hx);
        unsafeAtomicAdd(b + ((long)(y1 / 2) * LW + x1 / 2) * C, g * ly * lx);
    }
}

// ---- 3x3 avg pool (stage-2 head: adaptive_avg_pool2d(1) on [R,256,3,3]) ---------------------
__global__ __launch_bounds__(EW_THREADS) void avgpool_fwd_kernel(const float *x, float *out, long R, int HW, int C)
{
    const long total = R * C;
    const float inv = 1.f / (float)HW;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
        const long r = i / C;
        const int c = (int)(i % C);
        float s = 0.f;
        for (int k = 0; k < HW; ++k) s += x[(r * HW + k) * C + c];
        out[i] = s * inv;
    }
}
// inference tail of the stage-2 head in one pass: mean over the HW positions of relu(y*scale + shift + res)
// (backbones/resnet.py:48-53 bn3 + residual + relu, then fasterrcnn_detector.py:15 adaptive_avg_pool2d(1)); the
// activated [R,HW,C] tensor is never written
__global__ __launch_bounds__(EW_THREADS) void bn_res_relu_avgpool_kernel(const f32x4 *y, const float *scale, const float *shift,
                                                                         const f32x4 *res, f32x4 *out, long R, int HW, int C4)
{
    const long total = R * C4;
    const float inv = 1.f / (float)HW;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
        const long r = i / C4;
        const int c4 = (int)(i - r * C4);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c4 * 4), sh = *reinterpret_cast<const f32x4 *>(shift + c4 * 4);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < HW; ++k) {
            const long o = (r * HW + k) * C4 + c4;
            f32x4 v = y[o] * sc + sh + res[o];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            s += v;
        }
        out[i] = s * inv;
    }
}
__global__ __launch_bounds__(EW_THREADS) void avgpool_bwd_kernel(const float *dout, float *dx, long R, int HW, int C)
{
    const long total = R * HW * C;
    const float inv = 1.f / (float)HW;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C);
        const long r = i / ((long)HW * C);
        dx[i] = dout[r * C + c] * inv;
    }
}

// ---- WH head (detectors/centernet_detector.py:26-77): the 17x1 and 1x17 convolutions to ONE channel
// are computed as a 1x1 convolution to 2*k per-tap partial products T (MFMA, reads the 256-channel
// map once) followed by this shift-sum: out[...,0] = bW + sum_s T[h, w+s-k/2, k+s],
// out[...,1] = bH + sum_r T[h+r-k/2, w, r].  (Direct 17-tap convolutions to 1 channel run the
// matrix cores at 1/32 occupancy and read the map 17 times.)
__global__ __launch_bounds__(EW_THREADS) void wh_shift_sum_fwd_kernel(const float *T, const float *bw, const float *bh,
                                                                      float *out, int N, int H, int W, int k, int C)
{
    const long total = (long)N * H * W;
    const int half = k / 2;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const float *base = T + (i - (long)h * W - w) * C;     // start of this image
        float sh = bh[0], sw = bw[0];
        for (int r = 0; r < k; ++r) {
            const int hh = h + r - half;
            if (hh >= 0 && hh < H) sh += base[((long)hh * W + w) * C + r];
            const int ww = w + r - half;
            if (ww >= 0 && ww < W) sw += base[((long)h * W + ww) * C + k + r];
        }
        out[i * 2 + 0] = sw;
        out[i * 2 + 1] = sh;
    }
}

__global__ __launch_bounds__(EW_THREADS) void wh_shift_sum_bwd_kernel(const float *dout, float *dT, int N, int H, int W, int k, int C)
{
    const int half = k / 2;
    const long total = (long)N * H * W * C;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        float g = 0.f;
        if (c < k) {                 // T[h, w, r] feeds out_H[h - (r - half), w]
            const int ho = h - (c - half);
            if (ho >= 0 && ho < H) g = dout[(p + (long)(ho - h) * W) * 2 + 1];
        } else if (c < 2 * k) {      // T[h, w, k + s] feeds out_W[h, w - (s - half)]
            const int wo = w - (c - k - half);
            if (wo >= 0 && wo < W) g = dout[(p + (wo - w)) * 2 + 0];
        }
        dT[i] = g;
    }
}

// ---- fused Adam over the flat parameter buffer (torch.optim.Adam defaults, no weight decay,
//      no amsgrad; operators/rrnet_operator.py:29) -------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void adam_kernel(f32x4 *p, const f32x4 *g, f32x4 *m, f32x4 *v, long n4, float lr,
                                                          float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                          float grad_scale)
{
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
        const f32x4 gg = g[i] * grad_scale;
        f32x4 mm = m[i], vv = v[i], pp = p[i];
        mm = mm * b1 + gg * (1.f - b1);
        vv = vv * b2 + gg * gg * (1.f - b2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pp[e] -= step_size * (mm[e] / denom);
        }
        m[i] = mm; v[i] = vv; p[i] = pp;
    }
}

}  // namespace

#define EW_LAUNCH(kern, n4, stream, ...)                                                        \
    do {                                                                                        \
        if ((n4) > 0) hipLaunchKernelGGL(kern, dim3(ew_blocks(n4)), dim3(EW_THREADS), 0, stream, __VA_ARGS__); \
    } while (0)

extern "C" int rr_bn_reduce_slab(const double *slab, int mtiles, int c, double *sums, hipStream_t stream)
{
    RR_CHECK_ARG(mtiles > 0 && c > 0, "rr_bn_reduce_slab: bad dims");
    int parts = rr_cdiv(mtiles, 64);
    if (parts > 64) parts = 64;
    const int rows = rr_cdiv(mtiles, parts);
    hipLaunchKernelGGL(bn_reduce_slab_kernel, dim3(rr_cdiv(2 * c, 32), rr_cdiv(mtiles, rows)), dim3(256), 0, stream, slab,
                       mtiles, 2 * c, rows, sums);
    RR_CHECK_LAUNCH("rr_bn_reduce_slab");
    return RR_OK;
}

// dbeta += sums[0..C), dgamma += sums[C..2C): the affine gradients from the LOCAL BatchNorm-backward sums (SyncBN takes them before
// the sums are exchanged; one launch instead of two casts and two adds)
__global__ void bn_affine_grad_kernel(const double *sums, float *dgamma, float *dbeta, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] += (float)sums[c];
    dgamma[c] += (float)sums[C + c];
}

extern "C" int rr_bn_affine_grad(const double *sums, float *dgamma, float *dbeta, int c, hipStream_t stream)
{
    RR_CHECK_ARG(c > 0 && sums && dgamma && dbeta, "rr_bn_affine_grad: bad arguments");
    hipLaunchKernelGGL(bn_affine_grad_kernel, dim3(rr_cdiv(c, 128)), dim3(128), 0, stream, sums, dgamma, dbeta, c);
    RR_CHECK_LAUNCH("rr_bn_affine_grad");
    return RR_OK;
}

extern "C" int rr_bn_reduce_slab_count(const double *slab, int mtiles, int c, double *sums, double count, double *count_slot,
                                       hipStream_t stream)
{
    RR_CHECK_ARG(mtiles > 0 && c > 0 && count_slot, "rr_bn_reduce_slab_count: bad arguments");
    int parts = rr_cdiv(mtiles, 64);
    if (parts > 64) parts = 64;
    const int rows = rr_cdiv(mtiles, parts);
    hipLaunchKernelGGL(bn_reduce_slab_kernel, dim3(rr_cdiv(2 * c, 32), rr_cdiv(mtiles, rows)), dim3(256), 0, stream, slab,
                       mtiles, 2 * c, rows, sums, count, count_slot);
    RR_CHECK_LAUNCH("rr_bn_reduce_slab_count");
    return RR_OK;
}

extern "C" int rr_bn_finalize_count(const double *sums, const double *count_dev, const float *gamma, const float *beta,
                                    float *running_mean, float *running_var, float momentum, float eps, float *mean,
                                    float *invstd, float *scale, float *shift, int c, long *num_batches_tracked,
                                    double *count_out, hipStream_t stream)
{
    RR_CHECK_ARG(c > 0 && count_dev && count_out, "rr_bn_finalize_count: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(rr_cdiv(c, 128)), dim3(128), 0, stream, sums, 0.0, count_dev, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, invstd, scale, shift, c, num_batches_tracked, count_out);
    RR_CHECK_LAUNCH("rr_bn_finalize_count");
    return RR_OK;
}

extern "C" int rr_bn_finalize(const double *sums, double count, const double *count_dev, const float *gamma, const float *beta,
                              float *running_mean, float *running_var, float momentum, float eps, float *mean,
                              float *invstd, float *scale, float *shift, int c, long *num_batches_tracked,
                              hipStream_t stream)
{
    RR_CHECK_ARG(c > 0 && (count > 0 || count_dev), "rr_bn_finalize: bad dims");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(rr_cdiv(c, 128)), dim3(128), 0, stream, sums, count, count_dev, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, invstd, scale, shift, c, num_batches_tracked);
    RR_CHECK_LAUNCH("rr_bn_finalize");
    return RR_OK;
}

extern "C" int rr_bn_stats_finalize(const double *slab, int mtiles, double count, const float *gamma, const float *beta,
                                    float *running_mean, float *running_var, float momentum, float eps, float *mean,
                                    float *invstd, float *scale, float *shift, int c, long *num_batches_tracked,
                                    hipStream_t stream)
{
    RR_CHECK_ARG(c > 0 && count > 0 && mtiles > 0, "rr_bn_stats_finalize: bad dims");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(rr_cdiv(c, 32)), dim3(1024), 0, stream, slab, mtiles, count, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, invstd, scale, shift, c, num_batches_tracked);
    RR_CHECK_LAUNCH("rr_bn_stats_finalize");
    return RR_OK;
}

extern "C" int rr_bn_eval_coeffs(const float *gamma, const float *beta, const float *running_mean,
                                 const float *running_var, float eps, float *scale, float *shift, int c,
                                 hipStream_t stream)
{
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(rr_cdiv(c, 128)), dim3(128), 0, stream, gamma, beta, running_mean,
                       running_var, eps, scale, shift, c);
    RR_CHECK_LAUNCH("rr_bn_eval_coeffs");
    return RR_OK;
}

extern "C" int rr_bn_apply(const float *y, const float *scale, const float *shift, const float *res,
                           const float *res_scale, const float *res_shift, float *out, long total, int c, int relu,
                           hipStream_t stream)
{
    RR_CHECK_ARG(c % 4 == 0 && total % c == 0, "rr_bn_apply: C=%d must be a multiple of 4", c);
    const long n4 = total / 4;
    EW_LAUNCH(bn_apply_kernel<false>, n4, stream, (const f32x4 *)y, scale, shift, (const f32x4 *)res, res_scale, res_shift,
              (f32x4 *)out, n4, c / 4, relu, (unsigned *)nullptr);
    RR_CHECK_LAUNCH("rr_bn_apply");
    return RR_OK;
}

// ---- data gradient of a head's narrow 1x1 convolution, fused with the producer's ReLU mask and bias gradient ---------------
// dx[m][c] = (z[m][c] > 0) * ( [dx[m][c] +] sum_{k < K} dy[m][k] * w[k][c] ),  sums[c] += sum_m dx[m][c]      (K = 10 / 2 / 34: the
// hm / offset / WH heads' 1x1 layers behind a 3x3 conv + bias + ReLU, /root/reference/detectors/centernet_detector.py:62,73,85-93).
// As a GEMM this is 12 / 4 / 36 reduction indices wide: on the implicit-GEMM kernel (rr_conv_dgrad_s1_relubias) one launch took
// ~1.0 ms at 8 x 256 x 256 pixels — prologue / epilogue bound, 0.2 of what the bytes need.  Here it is what it is: an HBM-bound
// element-wise pass (read dy 21 MB + z 537 MB, write dx 537 MB: ~0.2 ms), K FMAs per output from a filter held in LDS; no channel
// padding of dy or w (rr_pad_channels), optional bf16 image of dx for the 3x3 layer's data / weight gradients (csrc/conv16.hip).
// KMAX: K rounded up to a compile-time bound (4 / 12 / 36): the thread's K x 4 filter values live in registers (zero beyond K).
// UNI: C == 256 — a wave is the 64 channel quads of ONE pixel, so dy[pix][0..K) is wave-uniform and comes through scalar loads.
template <int KMAX, bool UNI>
__global__ __launch_bounds__(EW_THREADS) void head_dgrad_relubias_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                                         f32x4 *__restrict__ dx, u16x4 *__restrict__ dx16,
                                                                         const f32x4 *__restrict__ z, double *__restrict__ sums, long m, int C, int K,
                                                                         int accumulate)
{
    extern __shared__ __align__(16) double red[];          // [PL][C]
    const int C4 = C / 4, cq = threadIdx.x % C4, PL = EW_THREADS / C4;
    int pl = threadIdx.x / C4;
    if constexpr (UNI) pl = __builtin_amdgcn_readfirstlane(pl);
    f32x4 wv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) wv[k] = k < K ? *reinterpret_cast<const f32x4 *>(w + (long)k * C + cq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    // software pipeline: the next pixel's dy row (scalar registers when UNI), mask and old gradient are fetched before the current
    // pixel's K x 4 FMAs — with 144 filter registers per lane only two waves fit a SIMD, and one exposed HBM latency per pixel
    // made the K = 34 layer 0.59 ms
    const long stride = (long)gridDim.x * PL;
    long pix = (long)blockIdx.x * PL + pl;
    float dcur[KMAX];
    f32x4 zcur = {0.f, 0.f, 0.f, 0.f}, ocur = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](long p, float (&d)[KMAX], f32x4 &zz, f32x4 &old) {
        const long q = p < m ? p : m - 1;                      // (past the end: a valid pixel, never stored)
        const float *row = dy + q * K;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) d[k] = row[k < K ? k : K - 1];      // beyond K the filter is zero: stay inside the row
        zz = z[q * C4 + cq];
        if (accumulate) old = dx[q * C4 + cq];
    };
    fetch(pix, dcur, zcur, ocur);
    while (pix < m) {
        float dnext[KMAX];
        f32x4 znext, onext = {0.f, 0.f, 0.f, 0.f};
        fetch(pix + stride, dnext, znext, onext);
        f32x4 acc = ocur;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(dcur[k], wv[k][e], acc[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] = zcur[e] > 0.f ? acc[e] : 0.f;
            s[e] += (double)acc[e];
        }
        const long o = pix * C4 + cq;
        dx[o] = acc;
        if (dx16) dx16[o] = to_bf16x4(acc);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) dcur[k] = dnext[k];
        zcur = znext; ocur = onext;
        pix += stride;
    }
    // column sums: over the workgroup's pixel lanes in LDS, one double atomic per channel and workgroup
#pragma unroll
    for (int e = 0; e < 4; ++e) red[pl * C + cq * 4 + e] = s[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += EW_THREADS) {
        double v = 0.0;
        for (int p = 0; p < PL; ++p) v += red[p * C + c];
        unsafeAtomicAdd(sums + c, v);
    }
}

extern "C" int rr_head_dgrad_relubias(const float *dy, const float *w, float *dx, unsigned short *dx16, const float *prod_z, double *sums,
                                      long m, int c, int k, int accumulate, hipStream_t stream)
{
    RR_CHECK_ARG(m > 0 && k > 0 && k <= 36 && c % 4 == 0 && c >= 4 && EW_THREADS % (c / 4) == 0 && dy && w && dx && prod_z && sums,
                 "rr_head_dgrad_relubias: K=%d (<= 36), C=%d (a divisor of %d x 4)", k, c, EW_THREADS);
    const int pl = EW_THREADS / (c / 4);
    const size_t ldsb = sizeof(double) * (size_t)pl * c;
    long blocks = (m + pl * 16 - 1) / (pl * 16);
    if (blocks > 2048) blocks = 2048;
#define RR_HD(KM, U) hipLaunchKernelGGL((head_dgrad_relubias_kernel<KM, U>), dim3((int)blocks), dim3(EW_THREADS), ldsb, stream, dy, w, \
                                        (f32x4 *)dx, (u16x4 *)dx16, (const f32x4 *)prod_z, sums, m, c, k, accumulate)
    const bool uni = c == 256;
    if (k <= 4) { if (uni) RR_HD(4, true); else RR_HD(4, false); }
    else if (k <= 12) { if (uni) RR_HD(12, true); else RR_HD(12, false); }
    else { if (uni) RR_HD(36, true); else RR_HD(36, false); }
#undef RR_HD
    RR_CHECK_LAUNCH("rr_head_dgrad_relubias");
    return RR_OK;
}

extern "C" int rr_bn_apply_b16(const float *y, const unsigned short *y16, const float *scale, const float *shift, const float *res, const unsigned short *res16,
                               const float *res_scale, const float *res_shift, float *out, unsigned short *out16, long total,
                               int c, int relu, hipStream_t stream)
{
    RR_CHECK_ARG(c % 4 == 0 && total % c == 0 && (out != nullptr || out16 != nullptr) && !(res != nullptr && res16 != nullptr)
                 && (y != nullptr) != (y16 != nullptr),
                 "rr_bn_apply_b16: C=%d must be a multiple of 4, one output required, y and the residual each in ONE precision", c);
    const long n4 = total / 4;
    EW_LAUNCH(bn_apply_kernel<false>, n4, stream, (const f32x4 *)y, scale, shift, (const f32x4 *)res, res_scale, res_shift,
              (f32x4 *)out, n4, c / 4, relu, (unsigned *)nullptr, (u16x4 *)out16, (const u16x4 *)res16, (const u16x4 *)y16);
    RR_CHECK_LAUNCH("rr_bn_apply_b16");
    return RR_OK;
}

__global__ __launch_bounds__(EW_THREADS) void from_bf16_kernel(const u16x4 *x, f32x4 *out, long n4)
{
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) out[i] = from_bf16x4(x[i]);
}

extern "C" int rr_from_bf16(const unsigned short *x, float *out, long total, hipStream_t stream)
{
    RR_CHECK_ARG(total % 4 == 0 && x != nullptr && out != nullptr, "rr_from_bf16: element count must be a multiple of 4");
    EW_LAUNCH(from_bf16_kernel, total / 4, stream, (const u16x4 *)x, (f32x4 *)out, total / 4);
    RR_CHECK_LAUNCH("rr_from_bf16");
    return RR_OK;
}

extern "C" int rr_upsample2x_add_b16(const float *up1, const unsigned short *up1_16, const float *low, const unsigned short *low_16,
                                     float *out, unsigned short *out16, int n, int h, int w, int c, hipStream_t stream)
{
    RR_CHECK_ARG(c % 4 == 0 && h % 2 == 0 && w % 2 == 0 && (up1 != nullptr) != (up1_16 != nullptr) && (low != nullptr) != (low_16 != nullptr)
                 && (out != nullptr || out16 != nullptr), "rr_upsample2x_add_b16: even H, W, C %% 4 == 0, each operand in ONE precision, one output");
    const long n4 = (long)n * h * w * (c / 4);
    EW_LAUNCH(upsample2x_add_kernel, n4, stream, (const f32x4 *)up1, (const f32x4 *)low, (f32x4 *)out, n, h, w, c / 4, (const u16x4 *)up1_16,
              (const u16x4 *)low_16, (u16x4 *)out16);
    RR_CHECK_LAUNCH("rr_upsample2x_add_b16");
    return RR_OK;
}

extern "C" int rr_bn_bwd_reduce_b16(const float *dz, const float *z, const unsigned short *z16, const float *y, const unsigned short *y16,
                                    const float *mean, const float *invstd, const float *mask_scale, const float *mask_shift,
                                    double *sums, long npix, int c, hipStream_t stream)
{
    // rr_bn_bwd_reduce with the ReLU mask's source and / or the pre-BN output read from their bf16 images (sums pre-zeroed)
    RR_CHECK_ARG(c % 4 == 0 && c <= 1024 && !(z != nullptr && z16 != nullptr) && (y != nullptr) != (y16 != nullptr),
                 "rr_bn_bwd_reduce_b16: C=%d must be a multiple of 4 and <= 1024; z and y each in ONE precision", c);
    const int c4 = c / 4;
    const int lanes = EW_THREADS / c4 > 0 ? EW_THREADS / c4 : 1;
    long blocks = (npix + lanes * 8 - 1) / (lanes * 8);
    const long cap = npix >= 400000 ? 1024 : (npix >= 16384 ? 512 : 256);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (y16 != nullptr && z == nullptr)       // (the ReLU mask from its image, from y, or none)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3((int)blocks), dim3(EW_THREADS), 0, stream, dz, z, y, mean, invstd, mask_scale,
                           mask_shift, sums, npix, c, z16, y16);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3((int)blocks), dim3(EW_THREADS), 0, stream, dz, z, y, mean, invstd, mask_scale,
                           mask_shift, sums, npix, c, z16, y16);
    RR_CHECK_LAUNCH("rr_bn_bwd_reduce_b16");
    return RR_OK;
}

__global__ __launch_bounds__(EW_THREADS) void to_bf16_kernel(const f32x4 *x, u16x4 *out, long n4)
{
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) out[i] = to_bf16x4(x[i]);
}

extern "C" int rr_to_bf16(const float *x, unsigned short *out, long total, hipStream_t stream)
{
    RR_CHECK_ARG(total % 4 == 0 && x != nullptr && out != nullptr, "rr_to_bf16: element count must be a multiple of 4");
    EW_LAUNCH(to_bf16_kernel, total / 4, stream, (const f32x4 *)x, (u16x4 *)out, total / 4);
    RR_CHECK_LAUNCH("rr_to_bf16");
    return RR_OK;
}

extern "C" int rr_bn_apply_amax(const float *y, const float *scale, const float *shift, const float *res,
                                const float *res_scale, const float *res_shift, float *out, long total, int c, int relu,
                                unsigned *amax_out, hipStream_t stream)
{
    RR_CHECK_ARG(c % 4 == 0 && total % c == 0 && amax_out != nullptr, "rr_bn_apply_amax: C=%d must be a multiple of 4, amax_out required", c);
    const long n4 = total / 4;
    EW_LAUNCH(bn_apply_kernel<true>, n4, stream, (const f32x4 *)y, scale, shift, (const f32x4 *)res, res_scale, res_shift,
              (f32x4 *)out, n4, c / 4, relu, amax_out);
    RR_CHECK_LAUNCH("rr_bn_apply_amax");
    return RR_OK;
}

extern "C" int rr_bn_bwd_reduce(const float *dz, const float *z, const float *y, const float *mean,
                                const float *invstd, const float *mask_scale, const float *mask_shift, double *sums,
                                long npix, int c, int sums_zeroed, hipStream_t stream)
{
    RR_CHECK_ARG(c % 4 == 0 && c <= 1024, "rr_bn_bwd_reduce: C=%d must be a multiple of 4 and <= 1024", c);
    if (!sums_zeroed) hipMemsetAsync(sums, 0, sizeof(double) * 2 * c, stream);
    const int c4 = c / 4;
    const int lanes = EW_THREADS / c4 > 0 ? EW_THREADS / c4 : 1;
    long blocks = (npix + lanes * 8 - 1) / (lanes * 8);
    // every workgroup ends with 2C double atomics on the same 2C addresses: beyond ~1000 workgroups that tail, not the
    // HBM stream, sets the time of the mid-sized layers (measured: 128x128x256 69 -> 48 us, 64x64x384 56 -> 29 us)
    const long cap = npix >= 400000 ? 1024 : (npix >= 16384 ? 512 : 256);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3((int)blocks), dim3(EW_THREADS), 0, stream, dz, z, y, mean, invstd, mask_scale,
                       mask_shift, sums, npix, c, (const unsigned short *)nullptr, (const unsigned short *)nullptr);
    RR_CHECK_LAUNCH("rr_bn_bwd_reduce");
    return RR_OK;
}

static int bn_bwd_apply_impl(const float *dz, const float *z, const float *y, const float *mean,
                             const float *invstd, const float *gamma, const float *mask_scale,
                             const float *mask_shift, const double *sums, double count,
                             const double *count_dev, float *dx, float *g_out, float *dgamma, float *dbeta,
                             long total, int c, int g_acc, hipStream_t stream, unsigned *amax = nullptr, unsigned short *dx16 = nullptr,
                             const unsigned short *z16 = nullptr, const unsigned short *y16 = nullptr)
{
    RR_CHECK_ARG(c % 4 == 0 && total % c == 0, "rr_bn_bwd_apply: C=%d must be a multiple of 4", c);
    const long n4 = total / 4;
    if (dx16 != nullptr || z16 != nullptr || y16 != nullptr)
        EW_LAUNCH((bn_bwd_apply_kernel<false, true>), n4, stream, (const f32x4 *)dz, (const f32x4 *)z, (const f32x4 *)y, mean, invstd, gamma,
                  mask_scale, mask_shift, sums, count, count_dev, (f32x4 *)dx, (f32x4 *)g_out, dgamma, dbeta, n4, c, g_acc,
                  (unsigned *)nullptr, (u16x4 *)dx16, (const u16x4 *)z16, (const u16x4 *)y16);
    else if (amax != nullptr)
        EW_LAUNCH(bn_bwd_apply_kernel<true>, n4, stream, (const f32x4 *)dz, (const f32x4 *)z, (const f32x4 *)y, mean, invstd, gamma,
                  mask_scale, mask_shift, sums, count, count_dev, (f32x4 *)dx, (f32x4 *)g_out, dgamma, dbeta, n4, c, g_acc, amax);
    else
        EW_LAUNCH(bn_bwd_apply_kernel<false>, n4, stream, (const f32x4 *)dz, (const f32x4 *)z, (const f32x4 *)y, mean, invstd, gamma,
                  mask_scale, mask_shift, sums, count, count_dev, (f32x4 *)dx, (f32x4 *)g_out, dgamma, dbeta, n4, c, g_acc,
                  (unsigned *)nullptr);
    RR_CHECK_LAUNCH("rr_bn_bwd_apply");
    return RR_OK;
}

extern "C" int rr_bn_bwd_apply_b16(const float *dz, const float *z, const unsigned short *z16, const float *y, const unsigned short *y16, const float *mean,
                                   const float *invstd, const float *gamma, const float *mask_scale,
                                   const float *mask_shift, const double *sums, double count,
                                   const double *count_dev, float *dx, unsigned short *dx16, float *g_out, int g_accumulate,
                                   float *dgamma, float *dbeta, long total, int c, hipStream_t stream)
{
    RR_CHECK_ARG((dx != nullptr || dx16 != nullptr) && (!g_accumulate || g_out != nullptr) && !(z != nullptr && z16 != nullptr)
                 && (y != nullptr) != (y16 != nullptr),
                 "rr_bn_bwd_apply_b16: one of dx / dx16 (and the fan-in buffer when accumulating) required; z and y each in ONE precision");
    return bn_bwd_apply_impl(dz, z, y, mean, invstd, gamma, mask_scale, mask_shift, sums, count, count_dev, dx, g_out, dgamma,
                             dbeta, total, c, g_accumulate ? 1 : 0, stream, nullptr, dx16, z16, y16);
}

extern "C" int rr_bn_bwd_apply_amax(const float *dz, const float *z, const float *y, const float *mean,
                                    const float *invstd, const float *gamma, const float *mask_scale,
                                    const float *mask_shift, const double *sums, double count,
                                    const double *count_dev, float *dx, float *g_out, int g_accumulate, float *dgamma,
                                    float *dbeta, long total, int c, unsigned *amax_dx, hipStream_t stream)
{
    RR_CHECK_ARG(amax_dx != nullptr && (!g_accumulate || g_out != nullptr), "rr_bn_bwd_apply_amax: amax_dx (and the fan-in buffer when accumulating) required");
    return bn_bwd_apply_impl(dz, z, y, mean, invstd, gamma, mask_scale, mask_shift, sums, count, count_dev, dx, g_out, dgamma,
                             dbeta, total, c, g_accumulate ? 1 : 0, stream, amax_dx);
}

extern "C" int rr_bn_bwd_apply(const float *dz, const float *z, const float *y, const float *mean,
                               const float *invstd, const float *gamma, const float *mask_scale,
                               const float *mask_shift, const double *sums, double count,
                               const double *count_dev, float *dx, float *g_out, float *dgamma, float *dbeta,
                               long total, int c, hipStream_t stream)
{
    return bn_bwd_apply_impl(dz, z, y, mean, invstd, gamma, mask_scale, mask_shift, sums, count, count_dev, dx, g_out, dgamma,
                             dbeta, total, c, 0, stream);
}

extern "C" int rr_bn_bwd_apply_gacc(const float *dz, const float *z, const float *y, const float *mean,
                                    const float *invstd, const float *gamma, const float *mask_scale,
                                    const float *mask_shift, const double *sums, double count,
                                    const double *count_dev, float *dx, float *g_acc, float *dgamma, float *dbeta,
                                    long total, int c, hipStream_t stream)
{
    RR_CHECK_ARG(g_acc != nullptr, "rr_bn_bwd_apply_gacc: the fan-in buffer is required");
    return bn_bwd_apply_impl(dz, z, y, mean, invstd, gamma, mask_scale, mask_shift, sums, count, count_dev, dx, g_acc, dgamma,
                             dbeta, total, c, 1, stream);
}

extern "C" int rr_relu_fwd(const float *x, float *out, long total, hipStream_t stream)
{
    RR_CHECK_ARG(total % 4 == 0, "rr_relu_fwd: element count must be a multiple of 4");
    EW_LAUNCH(relu_fwd_kernel, total / 4, stream, (const f32x4 *)x, (f32x4 *)out, total / 4);
    RR_CHECK_LAUNCH("rr_relu_fwd");
    return RR_OK;
}

extern "C" int rr_pad_channels(const float *src, float *dst, long npix, int k, int kp, hipStream_t stream)
{
    RR_CHECK_ARG(npix >= 0 && k > 0 && kp >= k && kp % 4 == 0, "rr_pad_channels: k=%d kp=%d (kp >= k, multiple of 4)", k, kp);
    EW_LAUNCH(pad_channels_kernel, npix * (kp / 4), stream, src, (f32x4 *)dst, npix * (kp / 4), k, kp / 4);
    RR_CHECK_LAUNCH("rr_pad_channels");
    return RR_OK;
}

extern "C" int rr_sum_n(const float *const *grads, int n, const float *z, float *out, long total, hipStream_t stream)
{
    RR_CHECK_ARG(n >= 1 && n <= 8 && total % 4 == 0, "rr_sum_n: 1..8 inputs, element count multiple of 4");
    SumArgs a;
    for (int i = 0; i < 8; ++i) a.g[i] = (const f32x4 *)grads[i < n ? i : 0];
    a.n = n;
    EW_LAUNCH(sum_n_kernel, total / 4, stream, a, (const f32x4 *)z, (f32x4 *)out, total / 4);
    RR_CHECK_LAUNCH("rr_sum_n");
    return RR_OK;
}

extern "C" int rr_bias_relu_bwd(const float *dy, const float *z, float *dy_masked, float *dbias, long npix, int c,
                                hipStream_t stream)
{
    RR_CHECK_ARG(c > 0 && npix >= 0, "rr_bias_relu_bwd: bad dims");
    if (npix == 0) return RR_OK;
    if (c % 4 == 0 && c / 4 <= EW_THREADS && EW_THREADS % (c / 4) == 0) {
        const int lanes4 = EW_THREADS / (c / 4);
        long blocks4 = (npix + lanes4 * 8 - 1) / (lanes4 * 8);
        if (blocks4 > 1024) blocks4 = 1024;      // every workgroup ends with C float atomics on the same C addresses
        hipLaunchKernelGGL(colsum4_kernel, dim3((int)blocks4), dim3(EW_THREADS), 0, stream, (const f32x4 *)dy, (const f32x4 *)z,
                           (f32x4 *)dy_masked, dbias, npix, c / 4);
        RR_CHECK_LAUNCH("rr_bias_relu_bwd");
        return RR_OK;
    }
    const int cpb = c < EW_THREADS ? c : EW_THREADS;
    const int lanes = EW_THREADS / cpb;
    long blocks = (npix + lanes * 16 - 1) / (lanes * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(colsum_kernel, dim3((int)blocks), dim3(EW_THREADS), 0, stream, dy, z, dy_masked, dbias, npix, c);
    RR_CHECK_LAUNCH("rr_bias_relu_bwd");
    return RR_OK;
}

extern "C" int rr_upsample_add_fwd(const float *up1, const float *low, float *out, int n, int h, int w, int lh,
                                   int lw, int c, hipStream_t stream)
{
    if (h == 2 * lh && w == 2 * lw && c % 4 == 0) {
        const long n4 = (long)n * h * w * (c / 4);
        EW_LAUNCH(upsample2x_add_kernel, n4, stream, (const f32x4 *)up1, (const f32x4 *)low, (f32x4 *)out, n, h, w, c / 4);
    } else {
        const long tot = (long)n * h * w * c;
        EW_LAUNCH(upsample_bilinear_add_kernel, tot, stream, up1, low, out, n, h, w, lh, lw, c);
    }
    RR_CHECK_LAUNCH("rr_upsample_add_fwd");
    return RR_OK;
}

extern "C" int rr_upsample_add_bwd(const float *dout, float *dlow, int n, int h, int w, int lh, int lw, int c,
                                   hipStream_t stream)
{
    if (h == 2 * lh && w == 2 * lw && c % 4 == 0) {
        const long n4 = (long)n * lh * lw * (c / 4);
        EW_LAUNCH(upsample2x_bwd_kernel, n4, stream, (const f32x4 *)dout, (f32x4 *)dlow, n, h, w, c / 4);
    } else {
        hipMemsetAsync(dlow, 0, sizeof(float) * (size_t)n * lh * lw * c, stream);
        const long tot = (long)n * h * w * c;
        EW_LAUNCH(upsample_bilinear_bwd_kernel, tot, stream, dout, dlow, n, h, w, lh, lw, c);
    }
    RR_CHECK_LAUNCH("rr_upsample_add_bwd");
    return RR_OK;
}

extern "C" int rr_resize_bilinear_ac(const float *x, float *out, int n, int h, int w, int oh, int ow, int c, hipStream_t stream)
{
    RR_CHECK_ARG(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && c > 0, "rr_resize_bilinear_ac: bad dims");
    EW_LAUNCH(resize_bilinear_ac_kernel, (long)n * oh * ow * c, stream, x, out, n, h, w, oh, ow, c);
    RR_CHECK_LAUNCH("rr_resize_bilinear_ac");
    return RR_OK;
}

extern "C" int rr_avgpool_fwd(const float *x, float *out, long r, int hw, int c, hipStream_t stream)
{
    EW_LAUNCH(avgpool_fwd_kernel, r * c, stream, x, out, r, hw, c);
    RR_CHECK_LAUNCH("rr_avgpool_fwd");
    return RR_OK;
}

extern "C" int rr_bn_res_relu_avgpool(const float *y, const float *scale, const float *shift, const float *res, float *out,
                                      long r, int hw, int c, hipStream_t stream)
{
    RR_CHECK_ARG(r >= 0 && hw > 0 && c > 0 && c % 4 == 0, "rr_bn_res_relu_avgpool: bad dims (C multiple of 4)");
    if (r == 0) return RR_OK;
    EW_LAUNCH(bn_res_relu_avgpool_kernel, r * (c / 4), stream, (const f32x4 *)y, scale, shift, (const f32x4 *)res, (f32x4 *)out, r,
              hw, c / 4);
    RR_CHECK_LAUNCH("rr_bn_res_relu_avgpool");
    return RR_OK;
}

extern "C" int rr_avgpool_bwd(const float *dout, float *dx, long r, int hw, int c, hipStream_t stream)
{
    EW_LAUNCH(avgpool_bwd_kernel, r * hw * c, stream, dout, dx, r, hw, c);
    RR_CHECK_LAUNCH("rr_avgpool_bwd");
    return RR_OK;
}

extern "C" int rr_wh_shift_sum_fwd(const float *t, const float *bias_w, const float *bias_h, float *out, int n, int h,
                                   int w, int k, int ct, hipStream_t stream)
{
    RR_CHECK_ARG(n > 0 && h > 0 && w > 0 && k > 0 && (k & 1) && ct >= 2 * k, "rr_wh_shift_sum_fwd: bad dims (k odd, ct >= 2k)");
    const long tot = (long)n * h * w;
    EW_LAUNCH(wh_shift_sum_fwd_kernel, tot, stream, t, bias_w, bias_h, out, n, h, w, k, ct);
    RR_CHECK_LAUNCH("rr_wh_shift_sum_fwd");
    return RR_OK;
}

extern "C" int rr_wh_shift_sum_bwd(const float *dout, float *dt, int n, int h, int w, int k, int ct, hipStream_t stream)
{
    RR_CHECK_ARG(n > 0 && h > 0 && w > 0 && k > 0 && (k & 1) && ct >= 2 * k, "rr_wh_shift_sum_bwd: bad dims (k odd, ct >= 2k)");
    const long tot = (long)n * h * w * ct;
    EW_LAUNCH(wh_shift_sum_bwd_kernel, tot, stream, dout, dt, n, h, w, k, ct);
    RR_CHECK_LAUNCH("rr_wh_shift_sum_bwd");
    return RR_OK;
}

extern "C" int rr_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                            float beta1, float beta2, float eps, int step, float grad_scale, hipStream_t stream)
{
    RR_CHECK_ARG(n % 4 == 0 && step >= 1, "rr_adam_step: n must be a multiple of 4 (pad the flat buffer), step >= 1");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    EW_LAUNCH(adam_kernel, n / 4, stream, (f32x4 *)param, (const f32x4 *)grad, (f32x4 *)exp_avg, (f32x4 *)exp_avg_sq, n / 4,
              lr, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale);
    RR_CHECK_LAUNCH("rr_adam_step");
    return RR_OK;
}
