// Inference tail of the re-regression head for gfx950, fused: conv3 (1x1, planes -> 4*planes) + bn3 (eval, folded) +
// residual + ReLU + global average pool — backbones/resnet.py:46-53 and detectors/fasterrcnn_detector.py:15 of the
// reference — in ONE kernel.  Unfused (round 1) the 1x1 convolution wrote its [R*9, 256] output (1.77 GB per 128
// frames at config 5) only for the next kernel to read it back with the residual and reduce it 9:1; here only the
// pooled [R, 256] tensor is written.  HBM-bound: algorithmic bytes per RoI = 9*(K + N)*4 read + N*4 written
// (K = 64, N = 256: 11.5 KB, 2.2 GB per 128 frames).
// One workgroup per 32 RoIs (288 rows for 3x3 RoIs), four waves = four 64-column slices of the output; a wave keeps
// its slice of W3 (64 x K) in 64 registers for the whole workgroup, streams the rows through in sub-tiles of 32
// (operand fragments straight from global memory: the tile is consumed once), and adds relu(...) of every row into
// the RoI's pooled accumulator in LDS (ds_add_f32).  No global atomics, no zero-fill of the output; several
// workgroups per CU overlap each other's fetches.
#include "common.h"
#include "rrnet_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HT_ROIS = 32, HT_NMAX = 256, ST_LD = 68;

template <int KK>   // KK = K / 8
__global__ __launch_bounds__(256, 2) void head_tail_kernel(const float *h, const float *w, const float *scale, const float *shift,
                                                        const float *res, float *out, long R, int N, int HW)
{
    __shared__ float pooled[HT_ROIS * HT_NMAX];
    __shared__ __align__(16) float stage[4 * 32 * ST_LD];
    constexpr int K = KK * 8;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, lr = lane & 31, lh = lane >> 5;
    const long r0 = (long)blockIdx.x * HT_ROIS;
    const long M = R * HW;
    const long m_begin = r0 * HW;
    long m_end = (r0 + HT_ROIS) * HW;
    if (m_end > M) m_end = M;
    for (int i = t; i < HT_ROIS * HT_NMAX; i += 256) pooled[i] = 0.f;
    // this wave's columns [64 wv, 64 wv + 64): B fragments b[j][kk] = W[n = 64 wv + 32 j + lr][8 kk + 4 lh .. +3]
    f32x4 bfr[2][KK];
    float sc[2], sh[2];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wv * 64 + j * 32 + lr;
        sc[j] = n < N ? scale[n] : 0.f;
        sh[j] = n < N ? shift[n] : 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            bfr[j][kk] = n < N ? *reinterpret_cast<const f32x4 *>(w + (long)n * K + kk * 8 + lh * 4) : z4;
    }
    __syncthreads();
    const int nsub = (int)((m_end - m_begin + 31) / 32);
    float *st = stage + wv * (32 * ST_LD);             // this wave's [32 rows][64 cols] transpose buffer
    for (int sub = 0; sub < nsub; ++sub) {
        const long mb = m_begin + (long)sub * 32;
        // operand rows straight from global memory (16 B per lane) and the residual in row-major order (16 B per lane,
        // 4 rows x 256 B per instruction): the accumulators are transposed through LDS to meet it
        f32x4 afr[KK], rv[8];
        const long mrow = mb + lr;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            afr[kk] = mrow < m_end ? *reinterpret_cast<const f32x4 *>(h + mrow * K + kk * 8 + lh * 4) : z4;
        const int rrow = lane >> 4, rcol = (lane & 15) * 4;      // row-major view: lane group g owns rows 8g .. 8g+7
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            const long m = mb + rrow * 8 + s8;
            const int n = wv * 64 + rcol;
            rv[s8] = (m < m_end && n < N) ? *reinterpret_cast<const f32x4 *>(res + m * N + n) : z4;
        }
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[kk][e], bfr[j][kk][e], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                st[((e & 3) + 8 * (e >> 2) + 4 * lh) * ST_LD + j * 32 + lr] = acc[j][e] * sc[j] + sh[j];
        // the wave reads back, in another layout, what its own lanes just stored: the hardware orders a wave's LDS
        // traffic, the fence + wave barrier tell the COMPILER that these stores and the loads below conflict
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // a lane walks 8 consecutive rows and keeps the running sum of the current RoI in registers: one LDS add per
        // RoI change instead of one per row (and no four-lanes-one-address conflicts)
        f32x4 run = z4;
        int cur = -1;
        auto flush = [&]() {
            if (cur < 0) return;
            float *p = pooled + cur * HT_NMAX + wv * 64 + rcol;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (wv * 64 + rcol + c < N) atomicAdd(p + c, run[c]);   // ds_add_f32
        };
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            const int rl = rrow * 8 + s8;
            const long m = mb + rl;
            if (m >= m_end) break;
            const int roi = (int)((m - m_begin) / HW);
            if (roi != cur) { flush(); cur = roi; run = z4; }
            const f32x4 y = *reinterpret_cast<const f32x4 *>(st + rl * ST_LD + rcol);
#pragma unroll
            for (int c = 0; c < 4; ++c) run[c] += fmaxf(y[c] + rv[s8][c], 0.f);
        }
        flush();
        // the next sub-tile's stores must not move above this sub-tile's loads of `st`
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const float inv = 1.f / (float)HW;
    const int nroi = (int)((m_end - m_begin) / HW);
    for (int i = t; i < nroi * N; i += 256) {
        const int rl = i / N, n = i - rl * N;
        out[(r0 + rl) * N + n] = pooled[rl * HT_NMAX + n] * inv;
    }
}

}  // namespace

extern "C" int rr_conv1x1_bn_res_relu_avgpool(const float *h, const float *w, const float *scale, const float *shift,
                                              const float *res, float *out, long r, int hw, int k, int n,
                                              hipStream_t stream)
{
    RR_CHECK_ARG(r >= 0 && hw > 0 && k > 0 && n > 0, "rr_conv1x1_bn_res_relu_avgpool: bad dims");
    RR_CHECK_ARG((k == 32 || k == 64) && n <= HT_NMAX && n % 4 == 0,
                 "rr_conv1x1_bn_res_relu_avgpool: K=%d (32 or 64), N=%d (multiple of 4, <= %d)", k, n, HT_NMAX);
    if (r == 0) return RR_OK;
    const unsigned blocks = (unsigned)((r + HT_ROIS - 1) / HT_ROIS);
    if (k == 64) hipLaunchKernelGGL(head_tail_kernel<8>, dim3(blocks), dim3(256), 0, stream, h, w, scale, shift, res, out, r, n, hw);
    else hipLaunchKernelGGL(head_tail_kernel<4>, dim3(blocks), dim3(256), 0, stream, h, w, scale, shift, res, out, r, n, hw);
    RR_CHECK_LAUNCH("rr_conv1x1_bn_res_relu_avgpool");
    return RR_OK;
}
