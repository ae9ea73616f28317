// Inference tail of the re-regression head for gfx950, fused: conv3 (1x1, planes -> 4*planes) + bn3 (eval, folded) +
// residual + ReLU + global average pool — backbones/resnet.py:46-53 and detectors/fasterrcnn_detector.py:15 of the
// reference — in ONE kernel.  Unfused (round 1) the 1x1 convolution wrote its [R*9, 256] output (1.77 GB per 128
// frames at config 5) only for the next kernel to read it back with the residual and reduce it 9:1; here only the
// pooled [R, 256] tensor is written.  HBM-bound: algorithmic bytes per RoI = 9*(K + N)*4 read + N*4 written
// (K = 64, N = 256: 11.5 KB, 2.2 GB per 128 frames).
// One workgroup per 32 RoIs, four waves = four 64-column slices of the output; a wave keeps its slice of W3 (64 x K) in
// 64 registers for the whole workgroup.  The rows are taken POSITION-MAJOR: MFMA tile p holds pixel p of each of the 32
// RoIs (row stride HW), so the average pool is a plain sum of the HW tiles' activated accumulators, element by element
// in registers — no transpose through LDS, no pooled buffer, no atomics (the row-major version of round 2 spent ~500
// VALU instructions and 40 LDS operations per tile on exactly that and used 67 KB of LDS per workgroup).  Operand
// fragments come straight from global memory (the tile is consumed once); the residual is read in the accumulator
// layout (lane = column: a wave's load covers two rows x 128 B).  The next tile's residual is requested before, its
// operand rows right after the MFMAs of the current tile.
#include "common.h"
#include "rrnet_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HT_ROIS = 32, HT_NMAX = 256;

template <int KK>   // KK = K / 8
__global__ __launch_bounds__(256, 2) void head_tail_kernel(const float *h, const float *w, const float *scale, const float *shift,
                                                        const float *res, float *out, long R, int N, int HW)
{
    constexpr int K = KK * 8;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, lr = lane & 31, lh = lane >> 5;
    const long r0 = (long)blockIdx.x * HT_ROIS;
    const int nroi = R - r0 < HT_ROIS ? (int)(R - r0) : HT_ROIS;
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
    // Both streams go through buffer descriptors that cover exactly this workgroup's RoIs: rows past the last RoI read
    // 0 from the hardware's range check, per-lane state is ONE 32-bit offset per stream, and the tile / row / K-step
    // part of every address travels in the instruction's scalar offset (no 64-bit address per load: 32 of them per
    // tile spilled).  Descriptor words pass through readfirstlane so that the compiler sees them wave-uniform.
    auto make_srd = [](const float *p, long bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_h = make_srd(h + r0 * HW * K, (long)nroi * HW * K * 4);
    const __amdgpu_buffer_rsrc_t rs_r = make_srd(res + r0 * HW * N, (long)nroi * HW * N * 4);
    // A fragment: lane (lr, lh) = RoI lr, reduction indices 8 kk + 4 lh .. +3 of pixel p
    const unsigned a_voff = (unsigned)((lr * HW * K + lh * 4) * 4);
    // D layout: col = lr, row (= RoI) = (e & 3) + 8 * (e >> 2) + 4 * lh
    unsigned r_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        r_voff[j] = wv * 64 + j * 32 + lr < N ? (unsigned)((4 * lh * HW * N + wv * 64 + j * 32 + lr) * 4) : 0xFFFFFFF0u;
    f32x4 afr[KK];
    float rv[2][16], rv_n[2][16];
    auto fetch_a = [&](int p) {
        const unsigned v = p < HW ? a_voff : 0xFFFFFFF0u;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            afr[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, v, (p * K + kk * 8) * 4, 0));
    };
    auto fetch_r = [&](int p, float (&fr)[2][16]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned v = p < HW ? r_voff[j] : 0xFFFFFFF0u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2);
                fr[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, v, (row * HW + p) * N * 4, 0));
            }
        }
    };
    f32x16 pool[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) pool[j][e] = 0.f;
    fetch_a(0);
    fetch_r(0, rv);
    for (int p = 0; p < HW; ++p) {
        fetch_r(p + 1, rv_n);
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
        // the MFMAs have read their operands at issue: the next tile's rows are fetched into the same registers
        __builtin_amdgcn_sched_barrier(0);
        fetch_a(p + 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                pool[j][e] += fmaxf(__builtin_fmaf(acc[j][e], sc[j], sh[j]) + rv[j][e], 0.f);
                rv[j][e] = rv_n[j][e];
            }
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * lh, n = wv * 64 + j * 32 + lr;
            if (row < nroi && n < N) out[(r0 + row) * N + n] = pool[j][e] * inv;
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
