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
    // Both streams go through buffer descriptors based at this workgroup's first RoI: per-lane state is ONE 32-bit offset
    // per stream, and the tile / row / K-step part of every address travels in the instruction's scalar offset (no 64-bit
    // address per load: 32 of them per tile spilled).  The hardware's range check covers the per-lane offset only (not the
    // scalar one), so rows past the last RoI (last workgroup) get the out-of-range per-lane offset explicitly and read 0.
    // Descriptor words pass through readfirstlane so that the compiler sees them wave-uniform.
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
    const unsigned a_voff = lr < nroi ? (unsigned)((lr * HW * K + lh * 4) * 4) : 0xFFFFFFF0u;
    // D layout: col = lr, row (= RoI) = (e & 3) + 8 * (e >> 2) + 4 * lh
    unsigned rmask = 0;                                   // bit e: accumulator row e of this lane is a real RoI
#pragma unroll
    for (int e = 0; e < 16; ++e) rmask |= ((e & 3) + 8 * (e >> 2) + 4 * lh < nroi ? 1u : 0u) << e;
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
                fr[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, ((rmask >> e) & 1u) ? v : 0xFFFFFFF0u,
                                                                                          (row * HW + p) * N * 4, 0));
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

// ---- 1x1 convolution to FEW output channels on MANY rows (the head's conv1: 256 -> 64 on R*9 rows) --------------------------
// y[m][n] = relu?(x[m][:] . w[n][:] + bias[n]).  The implicit-GEMM kernel runs this shape at 79 TFLOP/s / 3.0 TB/s: a 128 x 64
// tile has only 8 K-steps, so its prologue (first loads exposed) and epilogue are a third of the tile's time, and the two
// floors (0.37 ms of HBM, 0.36 ms of MFMA per 128 frames at config 5) do not overlap inside one short launch.  Here a wave
// keeps its 32-column slice of W (32 x K floats) in registers for the whole launch and streams 32-row tiles through:
// operand fragments straight from global memory (buffer loads, the tile / chunk part of the address in the scalar offset),
// 64-deep K chunks double-buffered in registers, and the workgroups are persistent — the first chunk of the next tile is
// requested under the MFMAs of the current tile's last chunk, so there is no per-tile prologue.  No LDS, no barriers.
// Four waves = 2 column tiles x 2 row tiles (64 rows per workgroup step).
template <int KK>   // K = 8 * KK, KK a multiple of 16
__global__ __launch_bounds__(256, 2) void rows_gemm_kernel(const float *x, const float *w, const float *bias, float *y, long M, int N,
                                                        int relu, int ntiles)
{
    // chunks of 4 fragment registers (32 reduction indices, 16 MFMAs); four register buffers in rotation, requests THREE
    // chunks ahead of their use (two waves per SIMD x 3 x 0.43 us of MFMA work cover ~2.5 us of fetch latency; one chunk
    // ahead left the waves waiting: 0.70 ms, the sum of the HBM and the MFMA time)
    constexpr int K = KK * 8, CH = KK / 4;
    static_assert(CH % 4 == 0, "four register buffers in rotation");
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, lr = lane & 31, lh = lane >> 5;
    const int ct = wv & 1, rg = wv >> 1;
    const int n = ct * 32 + lr;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bfr[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) bfr[kk] = n < N ? *reinterpret_cast<const f32x4 *>(w + (long)n * K + kk * 8 + lh * 4) : z4;
    const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
    auto make_srd = [](const float *p, long bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_x = make_srd(x, M * K * 4);          // (the host keeps M*K*4 < 2^31)
    const unsigned a_voff = (unsigned)((lr * K + lh * 4) * 4);
    f32x4 a[4][4];
    // tile = 64 rows; this wave's 32 rows start at tile * 64 + rg * 32.  `c` counts chunks from the start of `tile`
    // (c >= CH: the next tile of this workgroup)
    auto fetch = [&](f32x4 (&buf)[4], int tile, int c) {
        if (c >= CH) { c -= CH; tile += gridDim.x; }
        // rows past M: out-of-range per-lane offset (the scalar offset is not part of the hardware's range check)
        const unsigned v = (tile < ntiles && (long)tile * 64 + rg * 32 + lr < M) ? a_voff : 0xFFFFFFF0u;
        const int soff = __builtin_amdgcn_readfirstlane(tile < ntiles ? ((tile * 64 + rg * 32) * K + c * 32) * 4 : 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            buf[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, v, soff + kk * 32, 0));
    };
    f32x16 acc;
    auto mma = [&](const f32x4 (&buf)[4], int c) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[kk][e], bfr[c * 4 + kk][e], acc, 0, 0, 0);
    };
    int tile = blockIdx.x;
    fetch(a[0], tile, 0); fetch(a[1], tile, 1); fetch(a[2], tile, 2);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            fetch(a[(c + 3) & 3], tile, c + 3);
            mma(a[c & 3], c);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long row0 = (long)tile * 64 + rg * 32 + 4 * lh;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long m = row0 + (e & 3) + 8 * (e >> 2);
            float v = acc[e] + bv;
            if (relu) v = fmaxf(v, 0.f);
            if (m < M && n < N) y[m * N + n] = v;
        }
    }
}

}  // namespace

extern "C" int rr_conv1x1_rows(const float *x, const float *w, const float *bias, float *y, long m, int k, int n, int relu,
                               hipStream_t stream)
{
    RR_CHECK_ARG(m >= 0 && (k == 128 || k == 256) && n > 0 && n <= 64 && m * k * 4 < (1l << 31),
                 "rr_conv1x1_rows: K=%d (128 or 256), N=%d (<= 64), M*K*4 below 2 GiB", k, n);
    if (m == 0) return RR_OK;
    const int ntiles = (int)((m + 63) / 64);
    const int blocks = ntiles < 512 ? ntiles : 512;          // two workgroups per CU, persistent
    if (k == 256) hipLaunchKernelGGL(rows_gemm_kernel<32>, dim3(blocks), dim3(256), 0, stream, x, w, bias, y, m, n, relu, ntiles);
    else hipLaunchKernelGGL(rows_gemm_kernel<16>, dim3(blocks), dim3(256), 0, stream, x, w, bias, y, m, n, relu, ntiles);
    RR_CHECK_LAUNCH("rr_conv1x1_rows");
    return RR_OK;
}

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
