// Modulated deformable convolution (DCNv2) for gfx950 — BASELINE.json config 4, SURVEY §8 row a22.
//
// Replaces ext/dcn of the reference: `dcn_v2_cuda_forward/backward` (src/cuda/dcn_v2_cuda.cu:42-172,
// 206-335) and kernels K2-K5 (src/cuda/dcn_v2_im2col_cuda.cu:125-327), bound by ext/dcn/dcn_v2.py:16-52.
//   out[n,o,p] = b_o + sum_{c,i,j} W[o,c,i,j] * m[n,g,ij,p] * bilinear(x[n,c], p*s - pad + ij*dil + d[n,g,ij,p])
// All three GEMMs of the layer are gather-GEMMs on MFMA: the reference's im2col kernel + column buffer (9x the
// input) + batched cuBLAS GEMM become the operand load of an implicit GEMM — per (pixel, tap) four NHWC corner rows are
// read as float4s along C and blended with the bilinear x mask weights on their way into LDS.
//   * window kernels (dcn_fprop_win_kernel, dcn_wgrad_win_kernel, dcn_dgrad_win_kernel; stride 1, C % 32 == 0): a
//     workgroup owns an 8x16 block of output pixels and stages the input block it can reach in LDS once per 32-channel
//     chunk; templates over the operand precision (bf16 on v_mfma_f32_32x32x16_bf16, fp32 on v_mfma_f32_32x32x2_f32);
//     d input is pre-summed in a fixed-point LDS image before it goes out as global atomics;
//   * L2-gather kernels (dcn_fprop_kernel, dcn_fprop_bf16_kernel, dcn_wgrad_kernel, dcn_dgrad_kernel): the same GEMMs
//     with the corners fetched from global memory per K-step — every other layer shape, and RR_DCN_WINDOW=0;
//   * column path (dcn_im2col_kernel / dcn_col2im_kernel): the reference's structure ([M, R*S*C] columns materialised,
//     GEMMs on the conv kernels), kept as the A/B reference of the fused backward and for the layouts it does not take.
// Layouts: x NHWC; offset NHWC [N,P,Q,2*dg*R*S] (per group: interleaved (dh,dw) per tap, as the
// reference's channel order); mask NHWC [N,P,Q,dg*R*S]; weight OHWI.
#include "common.h"
#include "rrnet_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BK = 32, LDK = 36;

struct DcnArgs {
    const float *x, *offset, *mask, *w, *bias;
    float *y;
    const float *zero;
    int N, H, W, C, K, R, S, P, Q, stride, pad_h, pad_w, dil, dg;
    int M;
};

struct Tap4 {           // bilinear sample of one (pixel, tap): 4 corner element offsets (pixel base, no channel) + weights
    long o[4];
    float w[4];         // already multiplied by the modulation mask; 0 for corners / samples outside the image
};

__device__ __forceinline__ Tap4 make_tap(const DcnArgs &a, int n, int p, int q, int i, int j, float dh, float dw, float m)
{
    Tap4 t;
    const float h = (float)(p * a.stride - a.pad_h + i * a.dil) + dh;
    const float w = (float)(q * a.stride - a.pad_w + j * a.dil) + dw;
    const bool inside = h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;
    const float hf = floorf(h), wf = floorf(w);
    const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
    const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
    const bool v0 = inside && h0 >= 0 && w0 >= 0, v1 = inside && h0 >= 0 && w1 <= a.W - 1;
    const bool v2 = inside && h1 <= a.H - 1 && w0 >= 0, v3 = inside && h1 <= a.H - 1 && w1 <= a.W - 1;
    const long base = (long)n * a.H * a.W;
    t.o[0] = v0 ? (base + (long)h0 * a.W + w0) * a.C : -1;
    t.o[1] = v1 ? (base + (long)h0 * a.W + w1) * a.C : -1;
    t.o[2] = v2 ? (base + (long)h1 * a.W + w0) * a.C : -1;
    t.o[3] = v3 ? (base + (long)h1 * a.W + w1) * a.C : -1;
    t.w[0] = v0 ? hh * hw * m : 0.f;
    t.w[1] = v1 ? hh * lw * m : 0.f;
    t.w[2] = v2 ? lh * hw * m : 0.f;
    t.w[3] = v3 ? lh * lw * m : 0.f;
    return t;
}

template <int BN>
__global__ __launch_bounds__(256) void dcn_fprop_kernel(const DcnArgs a)
{
    constexpr int WN = BN / 64 ? BN / 64 : 1, WM = 4 / WN;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_ELEMS = BM * LDK, B_ELEMS = BN * LDK;
    constexpr int AJ = BM / 32, BJ = BN / 32;
    extern __shared__ __align__(16) float lds[];
    float *As = lds, *Bs = lds + 2 * A_ELEMS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
    const int ntiles = (a.K + BN - 1) / BN;
    const int n_tile = blockIdx.x % ntiles, m_tile = blockIdx.x / ntiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int RS = a.R * a.S;
    const int cpt = (a.C + BK - 1) / BK;          // channel chunks per tap
    const int nk = RS * cpt;                      // tap outer, channel chunk inner
    const int cpg = a.C / a.dg;
    const int a_col = (t & 7) * 4, a_row = t >> 3;

    int rn[AJ], rp[AJ], rq[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int m = m0 + a_row + 32 * j;
        if (m < a.M) {
            const int pq = a.P * a.Q;
            rn[j] = m / pq;
            const int rem = m - rn[j] * pq;
            rp[j] = rem / a.Q;
            rq[j] = rem - rp[j] * a.Q;
        } else {
            rn[j] = -1; rp[j] = 0; rq[j] = 0;
        }
    }

    f32x4 rv[AJ][4], rb[BJ];
    float rw[AJ][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // the sample geometry of a (row, tap) pair is shared by the channel chunks of the tap (K-steps run tap outer,
    // chunk inner): recomputed only when the tap or the deformable group changes — a wave-uniform branch
    Tap4 tc[AJ];
    int tc_tap = -1, tc_g = -1;
    auto issue = [&](int kc) {
        const int tap = kc / cpt, cch = kc - tap * cpt;
        const int i = tap / a.S, jx = tap - i * a.S;
        const int c0 = cch * BK;
        const int g = c0 / cpg;
        const bool c_ok = c0 + a_col < a.C;
        if (tap != tc_tap || g != tc_g) {
            tc_tap = tap; tc_g = g;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                if (rn[j] >= 0) {
                    const long m = (long)m0 + a_row + 32 * j;
                    const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                    const float mk = a.mask[m * (a.dg * RS) + g * RS + tap];
                    tc[j] = make_tap(a, rn[j], rp[j], rq[j], i, jx, po[0], po[1], mk);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { tc[j].o[e] = -1; tc[j].w[e] = 0.f; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float *src = (tc[j].o[e] >= 0 && c_ok) ? a.x + tc[j].o[e] + c0 + a_col : a.zero;
                rv[j][e] = *reinterpret_cast<const f32x4 *>(src);
                rw[j][e] = tc[j].w[e];
            }
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int ko = n0 + a_row + 32 * j;
            const bool ok = ko < a.K && c0 + a_col < a.C;
            const float *src = ok ? a.w + ((long)ko * RS + tap) * a.C + c0 + a_col : a.zero;
            rb[j] = *reinterpret_cast<const f32x4 *>(src);
        }
    };
    auto commit = [&](int buf) {      // blend the four corners and stage both tiles
        float *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const f32x4 v = rv[j][0] * rw[j][0] + rv[j][1] * rw[j][1] + rv[j][2] * rw[j][2] + rv[j][3] * rw[j][3];
            *reinterpret_cast<f32x4 *>(A + (a_row + 32 * j) * LDK + a_col) = v;
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4 *>(B + (a_row + 32 * j) * LDK + a_col) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;

    issue(0);
    commit(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) issue(kc + 1);
        const float *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const f32x4 *>(A + ((wm * TM + i) * 32 + lr) * LDK + kk * 8 + lh * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const f32x4 *>(B + ((wn * TN + j) * 32 + lr) * LDK + kk * 8 + lh * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < nk) commit(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ko = n0 + (wn * TN + j) * 32 + lr;
        if (ko >= a.K) continue;
        const float bv = a.bias ? a.bias[ko] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < a.M) a.y[(long)m * a.K + ko] = acc[i][j][e] + bv;
            }
    }
}

// ---- bf16-operand forward (BASELINE config 4) --------------------------------------------------------------------
// Same gather-GEMM; the blended samples and the weights are rounded to bf16 (round-to-nearest-even) on their way into
// LDS and multiplied on v_mfma_f32_32x32x16_bf16 (fp32 accumulation): 8 MFMAs per 32-channel K-step instead of 64,
// which moves the bound from the matrix pipe to the four-corner gather.  LDS tiles are [row][k] with 40 bf16 per
// row (16-byte aligned rows, one ds_read_b128 = the 8 k values a lane feeds to one MFMA).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDKH = BK + 8;

// float -> bf16, round to nearest even, on the hardware converter (v_cvt_pk_bf16_f32: two values per instruction).  The
// integer emulation this replaces (add 0x7fff + lsb, shift: four VALU operations per value) was a third of the VALU work of
// every staging thread — 192 of ≈330 operations per thread and K-step in the forward.
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned short f2bf(float f)
{
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ u16x4 f2bf4(f32x4 v)
{
    return __builtin_bit_cast(u16x4, __builtin_convertvector(v, bf16x4v));
}

template <int BN>
__global__ __launch_bounds__(256) void dcn_fprop_bf16_kernel(const DcnArgs a)
{
    constexpr int WN = BN / 64 ? BN / 64 : 1, WM = 4 / WN;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_ELEMS = BM * LDKH, B_ELEMS = BN * LDKH;      // in bf16 elements
    constexpr int AJ = BM / 32, BJ = BN / 32;
    extern __shared__ __align__(16) unsigned short ldsh[];
    unsigned short *As = ldsh, *Bs = ldsh + 2 * A_ELEMS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
    const int ntiles = (a.K + BN - 1) / BN;
    const int n_tile = blockIdx.x % ntiles, m_tile = blockIdx.x / ntiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int RS = a.R * a.S;
    const int cpt = (a.C + BK - 1) / BK;
    const int nk = RS * cpt;
    const int cpg = a.C / a.dg;
    const int a_col = (t & 7) * 4, a_row = t >> 3;

    int rn[AJ], rp[AJ], rq[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int m = m0 + a_row + 32 * j;
        if (m < a.M) {
            const int pq = a.P * a.Q;
            rn[j] = m / pq;
            const int rem = m - rn[j] * pq;
            rp[j] = rem / a.Q;
            rq[j] = rem - rp[j] * a.Q;
        } else {
            rn[j] = -1; rp[j] = 0; rq[j] = 0;
        }
    }
    f32x4 rv[AJ][4], rb[BJ];
    float rw[AJ][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // the sample geometry of a (row, tap) pair is shared by the channel chunks of the tap (K-steps run tap outer,
    // chunk inner): recomputed only when the tap or the deformable group changes — a wave-uniform branch
    Tap4 tc[AJ];
    int tc_tap = -1, tc_g = -1;
    auto issue = [&](int kc) {
        const int tap = kc / cpt, cch = kc - tap * cpt;
        const int i = tap / a.S, jx = tap - i * a.S;
        const int c0 = cch * BK;
        const int g = c0 / cpg;
        const bool c_ok = c0 + a_col < a.C;
        if (tap != tc_tap || g != tc_g) {
            tc_tap = tap; tc_g = g;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                if (rn[j] >= 0) {
                    const long m = (long)m0 + a_row + 32 * j;
                    const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                    const float mk = a.mask[m * (a.dg * RS) + g * RS + tap];
                    tc[j] = make_tap(a, rn[j], rp[j], rq[j], i, jx, po[0], po[1], mk);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { tc[j].o[e] = -1; tc[j].w[e] = 0.f; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float *src = (tc[j].o[e] >= 0 && c_ok) ? a.x + tc[j].o[e] + c0 + a_col : a.zero;
                rv[j][e] = *reinterpret_cast<const f32x4 *>(src);
                rw[j][e] = tc[j].w[e];
            }
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int ko = n0 + a_row + 32 * j;
            const bool ok = ko < a.K && c0 + a_col < a.C;
            const float *src = ok ? a.w + ((long)ko * RS + tap) * a.C + c0 + a_col : a.zero;
            rb[j] = *reinterpret_cast<const f32x4 *>(src);
        }
    };
    auto commit = [&](int buf) {
        unsigned short *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const f32x4 v = rv[j][0] * rw[j][0] + rv[j][1] * rw[j][1] + rv[j][2] * rw[j][2] + rv[j][3] * rw[j][3];
            *reinterpret_cast<u16x4 *>(A + (a_row + 32 * j) * LDKH + a_col) = f2bf4(v);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<u16x4 *>(B + (a_row + 32 * j) * LDKH + a_col) = f2bf4(rb[j]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;

    issue(0);
    commit(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) issue(kc + 1);
        const unsigned short *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            // both operands use the same (lane half, element) -> k map, so the MFMA's own k ordering is immaterial
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const bf16x8 *>(A + ((wm * TM + i) * 32 + lr) * LDKH + kk * 16 + lh * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const bf16x8 *>(B + ((wn * TN + j) * 32 + lr) * LDKH + kk * 16 + lh * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < nk) commit(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ko = n0 + (wn * TN + j) * 32 + lr;
        if (ko >= a.K) continue;
        const float bv = a.bias ? a.bias[ko] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < a.M) a.y[(long)m * a.K + ko] = acc[i][j][e] + bv;
            }
    }
}

// ---- forward with an LDS-staged input window (round 2; bf16 or fp32 operands) -------------------------------------------------
// The gather was the bound of dcn_fprop_bf16_kernel: every (pixel, tap) fetched its four corner rows from global
// memory, 19.4 GB per call through the fabric for a 0.54 GB input (each input row is wanted ~36 times, minutes apart in
// L2 terms).  Here a workgroup owns a 2-D block of TH x TW = 8 x 16 output pixels, and for every 32-channel chunk the
// input window that block can reach — the block, the filter's extent and a margin of RW pixels for the offsets — is
// staged ONCE into LDS (coalesced rows, prefetched into registers under the previous chunk's MFMAs); the 9 taps of
// the chunk gather their corners from LDS.  Corners that fall outside the window (|offset| beyond the margin) are
// fetched from global memory by the lanes concerned.  The sample geometry (4 corner indices + 4 mask-weighted bilinear
// weights per pixel and tap) is computed once per workgroup into LDS tables.  K-steps run chunk outer / tap inner.
// stride 1 only (the window of a strided layer is not compact); other layers take dcn_fprop_bf16_kernel.
constexpr int WIN_TH = 8, WIN_TW = 16;

struct DcnWinArgs {
    DcnArgs a;
    int RW, WH, WW;        // margin, window height / width in pixels
    int tiles_y, tiles_x;  // pixel blocks per image
    // bf16 forward: the weights pre-packed to bf16 [tap][32-channel chunk][filter][32] (rr_dcn_pack_weights_bf16), or null.
    // With them a K-step's B tile (256 filters x 32 channels = 16 KB, contiguous) goes global -> LDS by
    // buffer_load ... lds: no staging registers, no converts, no ds_write for the weight operand.
    const unsigned short *wpk;
};

// F32: fp32 matrix operands (v_mfma_f32_32x32x2_f32); a K-step is then HALF a window chunk (16 channels of one tap),
// which keeps the operand images at the bf16 kernel's bytes.
template <int BN, bool F32>      // 128 or 256 output channels per workgroup: at 256 one gather feeds twice the MFMAs
__global__ __launch_bounds__(512) void dcn_fprop_win_kernel(const DcnWinArgs wa)
{
    const DcnArgs &a = wa.a;
    // 512 threads: the window leaves room for one workgroup per CU, and two waves per SIMD hide the gather's LDS latency
    constexpr int NT = 512, WN = 4, WM = 2, TM = 2, TN = BN / 128, BJ = BN / 64;
    constexpr int A_ELEMS = BM * LDKH, B_ELEMS = BN * LDKH;      // 2-byte units (an fp32 row is 20 floats = 40 units)
    constexpr int H = F32 ? 2 : 1, LDF = 20, BJF = BN / 128;
    extern __shared__ __align__(16) unsigned char smem[];
    const int RS = a.R * a.S;
    const int npx = wa.WH * wa.WW;
    float *win = reinterpret_cast<float *>(smem);                               // [npx][32]
    unsigned int *geo_o = reinterpret_cast<unsigned int *>(win + (size_t)npx * BK);   // [BM][RS][4]
    float *geo_w = reinterpret_cast<float *>(geo_o + BM * RS * 4);              // [BM][RS][4]
    unsigned short *As = reinterpret_cast<unsigned short *>(geo_w + BM * RS * 4);
    unsigned short *Bs = As + 2 * A_ELEMS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
    const int ntiles = (a.K + BN - 1) / BN;
    int bid = blockIdx.x;
    const int n_tile = bid % ntiles; bid /= ntiles;
    const int txi = bid % wa.tiles_x; bid /= wa.tiles_x;
    const int tyi = bid % wa.tiles_y;
    const int n = bid / wa.tiles_y;
    const int y0 = tyi * WIN_TH, x0 = txi * WIN_TW, n0 = n_tile * BN;
    const int wy0 = y0 - a.pad_h - wa.RW, wx0 = x0 - a.pad_w - wa.RW;   // image coordinates of window pixel (0, 0)
    const int cpt = a.C / BK;                                            // channel chunks (C % 32 == 0 on this path)
    const int cpg = a.C / a.dg;
    const int a_col = (t & 7) * 4, a_row = t >> 3;
    const int fa_col = (t & 3) * 4, fa_row = t >> 2;                     // F32 staging: one 16-channel row quarter per thread
    const long img = (long)n * a.H * a.W;

    // ---- geometry tables for deformable group g
    auto build_geo = [&](int g) {
        for (int it = t; it < BM * RS; it += NT) {
            const int r = it / RS, tap = it - r * RS;
            const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
            unsigned int o[4] = {0u, 0u, 0u, 0u};
            float w4[4] = {0.f, 0.f, 0.f, 0.f};
            if (p < a.P && q < a.Q) {
                const long m = ((long)n * a.P + p) * a.Q + q;
                const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                const float mk = a.mask[m * (a.dg * RS) + g * RS + tap];
                const int ti = tap / a.S, tj = tap - ti * a.S;
                const float h = (float)(p - a.pad_h + ti * a.dil) + po[0];
                const float w = (float)(q - a.pad_w + tj * a.dil) + po[1];
                const bool inside = h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;
                const float hf = floorf(h), wf = floorf(w);
                const int h0 = (int)hf, w0 = (int)wf;
                const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
                const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int hy = h0 + (e >> 1), wx = w0 + (e & 1);
                    if (inside && hy >= 0 && hy <= a.H - 1 && wx >= 0 && wx <= a.W - 1) {
                        w4[e] = cw[e] * mk;
                        const int ly = hy - wy0, lx = wx - wx0;
                        o[e] = (ly >= 0 && ly < wa.WH && lx >= 0 && lx < wa.WW) ? (unsigned int)(ly * wa.WW + lx)
                                                                               : (0x80000000u | (unsigned int)(hy * a.W + wx));
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { geo_o[it * 4 + e] = o[e]; geo_w[it * 4 + e] = w4[e]; }
        }
    };
    // ---- window of channel chunk `cch`: global -> registers (prefetch) -> LDS
    constexpr int WREG = 7;                        // float4 per thread: windows up to 448 pixels
    f32x4 wreg[WREG];
    auto fetch_window = [&](int cch) {
        const int c0 = cch * BK;
#pragma unroll
        for (int u = 0; u < WREG; ++u) {
            const int i = t + u * NT;
            const int px = i >> 3, c4 = (i & 7) * 4;
            const int ly = px / wa.WW, lx = px - ly * wa.WW;
            const int gy = wy0 + ly, gx = wx0 + lx;
            const bool ok = px < npx && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            wreg[u] = *reinterpret_cast<const f32x4 *>(ok ? a.x + (img + (long)gy * a.W + gx) * a.C + c0 + c4 : a.zero);
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int u = 0; u < WREG; ++u) {
            const int i = t + u * NT;
            if ((i >> 3) < npx) *reinterpret_cast<f32x4 *>(win + (size_t)i * 4) = wreg[u];
        }
    };
    f32x4 rv[2], rb[BJ];
    // Operand staging in two phases so that a K-step's MFMAs run UNDER the next step's gather instead of behind it:
    // gather_a issues the loads (geometry words, then the four corner rows from the window; corners beyond the window
    // margin come from global memory through an unconditional load that reads the zero page when the corner is inside —
    // no divergent branch between the LDS reads and the matrix instructions), blend_a does the arithmetic afterwards.
    constexpr int GJ = F32 ? 1 : 2;
    f32x4 gxl[4], gwt;
    u32x4 gof;
    int g_c0 = 0, g_col = 0;
    auto gather_a = [&](int cch, int tap, int half, int j) {      // j: which of the thread's (bf16: two) staged rows
        g_c0 = cch * BK;
        g_col = F32 ? half * 16 + fa_col : a_col;
        const int r = F32 ? fa_row : a_row + 64 * j;
        gof = *reinterpret_cast<const u32x4 *>(geo_o + (r * RS + tap) * 4);
        gwt = *reinterpret_cast<const f32x4 *>(geo_w + (r * RS + tap) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int o = gof[e];
            gxl[e] = *reinterpret_cast<const f32x4 *>(win + (size_t)((o & 0x80000000u) ? 0u : o) * BK + g_col);
        }
    };
    auto blend_a = [&](int j) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        // Two versions behind a WAVE-UNIFORM branch: the common one has no global load in it.  With the per-lane fetch of
        // far corners inline, the wait for those (rare) loads sat behind the join — an unconditional s_waitcnt vmcnt(0) in
        // the middle of every K-step, which also waited for the weight tile's DMA issued just before (in-order counter):
        // 1.62 -> 1.44 ms at the bench layer (round 5).  (Tried and dropped: a ring of three weight images filled two steps
        // ahead by inline-assembly DMA — 1.54 ms; and producer / consumer waves (four waves gather + blend + DMA, four waves MFMA on
        // 64 x 128 tiles) — 1.52 ms: on this part one wave's MFMAs and another wave's VALU work on the same SIMD do not overlap,
        // the two roles' times add, as the split-operand kernels found in round 4.  Of the 1.44 ms about 0.5 are
        // not the K-steps: window fetch / store 0.27, output stores 0.17, geometry 0.03 (variant builds); workgroup turnover itself
        // is nothing — 4096 workgroups of this shape with 88 barriers each take 0.10 ms in all, tools/wg_turnover_probe.hip.)
        if (__builtin_amdgcn_ballot_w64(((gof[0] | gof[1] | gof[2] | gof[3]) & 0x80000000u) != 0) == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v += gxl[e] * gwt[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 xv = gxl[e];
                const unsigned int o = gof[e];
                if (o & 0x80000000u)       // a corner beyond the window margin (rare): straight from global memory
                    xv = *reinterpret_cast<const f32x4 *>(a.x + (img + (long)(o & 0x7fffffffu)) * a.C + g_c0 + g_col);
                v += xv * gwt[e];
            }
        }
        rv[j] = v;
    };
    auto build_a = [&](int cch, int tap, int half) {
#pragma unroll
        for (int j = 0; j < GJ; ++j) { gather_a(cch, tap, half, j); blend_a(j); }
    };
    auto issue_b = [&](int cch, int tap, int half) {
        const int c0 = cch * BK;
        if constexpr (F32) {
#pragma unroll
            for (int j = 0; j < BJF; ++j) {
                const int ko = n0 + fa_row + 128 * j;
                rb[j] = *reinterpret_cast<const f32x4 *>(ko < a.K ? a.w + ((long)ko * RS + tap) * a.C + c0 + half * 16 + fa_col : a.zero);
            }
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int ko = n0 + a_row + 64 * j;
                rb[j] = *reinterpret_cast<const f32x4 *>(ko < a.K ? a.w + ((long)ko * RS + tap) * a.C + c0 + a_col : a.zero);
            }
        }
    };
    auto commit = [&](int buf) {
        unsigned short *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
        if constexpr (F32) {
            *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(A) + fa_row * LDF + fa_col) = rv[0];
#pragma unroll
            for (int j = 0; j < BJF; ++j) *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(B) + (fa_row + 128 * j) * LDF + fa_col) = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<u16x4 *>(A + (a_row + 64 * j) * LDKH + a_col) = f2bf4(rv[j]);
            if (!(BN == 256 && wa.wpk != nullptr)) {
#pragma unroll
                for (int j = 0; j < BJ; ++j) *reinterpret_cast<u16x4 *>(B + (a_row + 64 * j) * LDKH + a_col) = f2bf4(rb[j]);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- pre-packed bf16 weights: B tile by LDS-DMA.  Image = [256 filters][32 k] bf16, 64-byte rows WITHOUT padding (a
    // wave-instruction's 64 x 16 B land contiguously), 16-byte chunks XOR-swizzled with (row >> 2) & 3 so that the 16 lanes
    // of a ds_read_b128 group (rows r, r+4, r+8, r+12 share a 16-bank quadrant) hit four different chunks.
    constexpr bool CAN_DMA = !F32 && BN == 256;
    const bool dma = CAN_DMA && wa.wpk != nullptr;
    typedef __attribute__((address_space(3))) void lds_void;
    __amdgpu_buffer_rsrc_t rs_wpk;
    {
        const unsigned long long u = reinterpret_cast<unsigned long long>(dma ? (const void *)wa.wpk : (const void *)a.w);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
        rs_wpk = __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)((long)RS * cpt * a.K * BK * 2)), 0x00020000);
    }
    auto dma_b = [&](int cch, int tap, int buf) {
        if constexpr (CAN_DMA) {
            unsigned short *B = Bs + buf * B_ELEMS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wave * 32 + i * 16 + (lane >> 2);           // filter row inside the tile
                const int chunk = (lane & 3) ^ ((row >> 2) & 3);             // global chunk that belongs in this lane's slot
                const int ko = n0 + row;
                const unsigned off = ko < a.K ? (unsigned)(((((long)tap * cpt + cch) * a.K + ko) * BK + chunk * 8) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wpk, (lds_void *)(B + (wave * 32 + i * 16) * BK), 16, off, 0, 0, 0);
            }
        }
    };

    int g_cur = 0;
    build_geo(0);
    fetch_window(0);
    store_window();
    __syncthreads();
    build_a(0, 0, 0);
    if (dma) dma_b(0, 0, 0); else issue_b(0, 0, 0);
    commit(0);
    if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int spc = RS * H;                                      // K-steps per window chunk: (tap, half) pairs
    const int nk = cpt * spc;
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        const int cch = kc / spc, rem = kc - cch * spc;
        const bool more = kc + 1 < nk;
        const bool new_chunk = more && rem == spc - 1;           // the next K-step opens chunk cch + 1
        if (rem == 0 && cch + 1 < cpt) fetch_window(cch + 1);    // lands under this chunk's K-steps
        const bool stage = more && !new_chunk;
        const unsigned short *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
        if constexpr (F32) {
            const float *Af = reinterpret_cast<const float *>(A), *Bf = reinterpret_cast<const float *>(B);
            // lane half lh takes channels 8 lh .. +7 of the 16-channel step (A and B agree on the pairing)
            f32x4 fa[TM][2], fb[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    fa[i][u] = *reinterpret_cast<const f32x4 *>(Af + ((wm * TM + i) * 32 + lr) * LDF + 8 * lh + 4 * u);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    fb[j][u] = *reinterpret_cast<const f32x4 *>(Bf + ((wn * TN + j) * 32 + lr) * LDF + 8 * lh + 4 * u);
            if (stage) {       // next step's operand loads go out behind this step's fragment reads, ahead of its MFMAs
                issue_b(cch, (rem + 1) / H, (rem + 1) % H);
                gather_a(cch, (rem + 1) / H, (rem + 1) % H, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][u][e], fb[j][u][e], acc[i][j], 0, 0, 0);
        } else {
            bf16x8 fa[BK / 16][TM], fb[BK / 16][TN];
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[kk][i] = *reinterpret_cast<const bf16x8 *>(A + ((wm * TM + i) * 32 + lr) * LDKH + kk * 16 + lh * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = (wn * TN + j) * 32 + lr;
                    fb[kk][j] = dma ? *reinterpret_cast<const bf16x8 *>(B + row * BK + (((kk * 2 + lh) ^ ((row >> 2) & 3)) * 8))
                                    : *reinterpret_cast<const bf16x8 *>(B + row * LDKH + kk * 16 + lh * 8);
                }
            }
            // the next step's operand loads go out behind this step's fragment reads and are consumed one MFMA group
            // later: row 0 of the thread's two staged rows under the first half of the K-step, row 1 under the second
            if (stage) {
                if (dma) dma_b(cch, (rem + 1) / H, buf ^ 1); else issue_b(cch, (rem + 1) / H, (rem + 1) % H);
                gather_a(cch, (rem + 1) / H, (rem + 1) % H, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (stage) {
                blend_a(0);
                gather_a(cch, (rem + 1) / H, (rem + 1) % H, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (stage) blend_a(1);
        }
        if constexpr (F32) {
            __builtin_amdgcn_sched_barrier(0);
            if (stage) blend_a(0);
        }
        if (new_chunk) {
            // every wave is past its last gather from the old window (it built the chunk's last step one K-step ago)
            __syncthreads();
            store_window();
            const int g = ((cch + 1) * BK) / cpg;
            if (g != g_cur) { g_cur = g; build_geo(g); }
            __syncthreads();
            if (dma) dma_b(cch + 1, 0, buf ^ 1); else issue_b(cch + 1, 0, 0);
            build_a(cch + 1, 0, 0);
        }
        if (more) commit(buf ^ 1);
        if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next B tile has landed
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ko = n0 + (wn * TN + j) * 32 + lr;
        if (ko >= a.K) continue;
        const float bv = a.bias ? a.bias[ko] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
                if (p < a.P && q < a.Q) a.y[(((long)n * a.P + p) * a.Q + q) * a.K + ko] = acc[i][j][e] + bv;
            }
    }
}

// columns [M][R*S*C] = mask * bilinear samples (only the backward needs them materialised)
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const DcnArgs a, float *col)
{
    const int RS = a.R * a.S, C4 = a.C / 4, cpg = a.C / a.dg;
    const long total = (long)a.M * RS * C4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C4) * 4;
        const long mt = idx / C4;
        const int tap = (int)(mt % RS);
        const long m = mt / RS;
        const int pq = a.P * a.Q;
        const int n = (int)(m / pq), rem = (int)(m - (long)n * pq), p = rem / a.Q, q = rem - p * a.Q;
        const int g = c / cpg, i = tap / a.S, j = tap - i * a.S;
        const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
        const Tap4 tp = make_tap(a, n, p, q, i, j, po[0], po[1], a.mask[m * (a.dg * RS) + g * RS + tap]);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (tp.o[e] >= 0) v += *reinterpret_cast<const f32x4 *>(a.x + tp.o[e] + c) * tp.w[e];
        *reinterpret_cast<f32x4 *>(col + (m * RS + tap) * a.C + c) = v;
    }
}

// one wave per (pixel, tap, group): dcol -> d input (atomics), d offset (h, w), d mask
// (dcn_v2_im2col_cuda.cu:197-327: col2im + col2im_coord)
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const DcnArgs a, const float *dcol, float *dx, float *doffset,
                                                         float *dmask)
{
    const int RS = a.R * a.S, cpg = a.C / a.dg;
    const long items = (long)a.M * RS * a.dg;
    const int lane = threadIdx.x & 63;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += (long)gridDim.x * 4) {
        const int g = (int)(it % a.dg);
        const long mt = it / a.dg;
        const int tap = (int)(mt % RS);
        const long m = mt / RS;
        const int pq = a.P * a.Q;
        const int n = (int)(m / pq), rem = (int)(m - (long)n * pq), p = rem / a.Q, q = rem - p * a.Q;
        const int i = tap / a.S, j = tap - i * a.S;
        const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
        const float mk = a.mask[m * (a.dg * RS) + g * RS + tap];
        const float h = (float)(p * a.stride - a.pad_h + i * a.dil) + po[0];
        const float w = (float)(q * a.stride - a.pad_w + j * a.dil) + po[1];
        const bool inside = h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;
        const float hf = floorf(h), wf = floorf(w);
        const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
        const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool v[4] = {inside && h0 >= 0 && w0 >= 0, inside && h0 >= 0 && w1 <= a.W - 1,
                           inside && h1 <= a.H - 1 && w0 >= 0, inside && h1 <= a.H - 1 && w1 <= a.W - 1};
        const long base = (long)n * a.H * a.W;
        const long o[4] = {(base + (long)h0 * a.W + w0) * a.C, (base + (long)h0 * a.W + w1) * a.C,
                           (base + (long)h1 * a.W + w0) * a.C, (base + (long)h1 * a.W + w1) * a.C};
        const float wt[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
        const float dh_w[4] = {-hw, -lw, hw, lw};      // d wt / d h
        const float dw_w[4] = {-hh, hh, -lh, lh};      // d wt / d w
        float s_mask = 0.f, s_h = 0.f, s_w = 0.f;
        for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
            const float gcol = dcol[(m * RS + tap) * a.C + c];
            const float gval = gcol * mk;
            float val = 0.f, gh = 0.f, gw = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (v[e]) {
                    const float xv = a.x[o[e] + c];
                    val += wt[e] * xv;
                    gh += dh_w[e] * xv;
                    gw += dw_w[e] * xv;
                    unsafeAtomicAdd(dx + o[e] + c, gval * wt[e]);
                }
            }
            s_mask += gcol * val;
            s_h += gval * gh;
            s_w += gval * gw;
        }
        s_mask = wave_sum(s_mask); s_h = wave_sum(s_h); s_w = wave_sum(s_w);
        if (lane == 0) {
            doffset[m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap] = s_h;
            doffset[m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap + 1] = s_w;
            dmask[m * (a.dg * RS) + g * RS + tap] = s_mask;
        }
    }
}

// ---- fused backward ---------------------------------------------------------------------------------------------
// The reference materialises the columns twice (dcn_v2_cuda.cu:206-335: im2col -> GEMM for dW; GEMM -> column
// gradient -> col2im / col2im_coord): 2 x 4.8 GB written and read at the config-4 layer.  Here neither buffer exists:
//   dcn_wgrad_kernel: dW[ko][tap][c] += sum_m dY[m][ko] * col[m][tap,c] with the column tile produced in registers
//                     from the four bilinear corners on its way into LDS (the forward's A-operand load, as B operand);
//   dcn_dgrad_kernel: per 128-pixel tile and tap, dcol[m][c] = sum_ko dY[m][ko] W[ko][tap][c] stays in the MFMA
//                     accumulators, goes through LDS once, and the epilogue turns it into d input (float atomics on
//                     the four corners), d offset and d mask (reduced over the channels in registers / LDS, stored
//                     once: no atomics, no memset for those two).
struct DcnBwdArgs {
    DcnArgs a;          // a.y unused
    const float *dy;    // [M][K]
    float *dw;          // [K][R][S][C], accumulated with float atomics
    float *dx, *doffset, *dmask;
    int chunks_per_split, mt, nt;
};

__device__ __forceinline__ int dcn_xcd_remap(int bid, int nb)
{
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__global__ __launch_bounds__(256, 2) void dcn_wgrad_kernel(const DcnBwdArgs b)
{
    const DcnArgs &a = b.a;
    constexpr int BMW = 128, BN = 128, TM = 2, TN = 2, WN = 2;
    constexpr int A_ELEMS = BK * BMW, B_ELEMS = BK * BN;
    extern __shared__ __align__(16) float lds[];
    float *As = lds, *Bs = lds + 2 * A_ELEMS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
    const int RS = a.R * a.S;
    int logical = dcn_xcd_remap(blockIdx.x, gridDim.x);
    const int tap = logical % RS; logical /= RS;
    const int n_tile = logical % b.nt; logical /= b.nt;
    const int m_tile = logical % b.mt;
    const int split = logical / b.mt;
    const int ti = tap / a.S, tj = tap - ti * a.S;
    const int ko0 = m_tile * BMW, c0 = n_tile * BN;
    const int total_chunks = (a.M + BK - 1) / BK;
    const int kc_begin = split * b.chunks_per_split;
    int kc_end = kc_begin + b.chunks_per_split;
    if (kc_end > total_chunks) kc_end = total_chunks;
    if (kc_begin >= kc_end) return;
    const int g = c0 / (a.C / a.dg);                  // one deformable group per channel tile (checked by the host)

    const int row = t >> 5, col4 = (t & 31) * 4;      // 8 pixel rows per pass, 4 passes; 4 ko / 4 channels per thread
    const bool ko_ok = ko0 + col4 < a.K, c_ok = c0 + col4 < a.C;
    // Register budget (two workgroups per CU need <= 256 VGPR + AGPR per lane): the column rows of a K-step are
    // fetched in two halves (rows 0-1 under the first 8 MFMA steps, rows 2-3 under the last 8), so only 2 x 4 corner
    // vectors are in flight at a time.
    f32x4 ra[4], rv[2][4];
    float rw[2][4];
    float om[4][3];                                   // (dh, dw, mask) of the rows of the K-step after next
    const int pq = a.P * a.Q;

    auto load_om = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long m = (long)kc * BK + row + 8 * j;
            if (kc < kc_end && m < a.M) {
                const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                om[j][0] = po[0]; om[j][1] = po[1];
                om[j][2] = a.mask[m * (a.dg * RS) + g * RS + tap];
            } else {
                om[j][0] = 0.f; om[j][1] = 0.f; om[j][2] = 0.f;
            }
        }
    };
    auto issue_a = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long m = (long)kc * BK + row + 8 * j;
            ra[j] = *reinterpret_cast<const f32x4 *>(m < a.M && ko_ok ? b.dy + m * a.K + ko0 + col4 : a.zero);
        }
    };
    auto issue_b = [&](int kc, int half) {            // the four corners of rows 2*half, 2*half+1
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * half + jj;
            const long m = (long)kc * BK + row + 8 * j;
            Tap4 tp;
            if (m < a.M) {
                const int n = (int)(m / pq), rem = (int)(m - (long)n * pq), p = rem / a.Q, q = rem - p * a.Q;
                tp = make_tap(a, n, p, q, ti, tj, om[j][0], om[j][1], om[j][2]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { tp.o[e] = -1; tp.w[e] = 0.f; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rv[jj][e] = *reinterpret_cast<const f32x4 *>(tp.o[e] >= 0 && c_ok ? a.x + tp.o[e] + c0 + col4 : a.zero);
                rw[jj][e] = tp.w[e];
            }
        }
    };
    auto commit_a = [&](int buf) {
        float *A = As + buf * A_ELEMS;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(A + (row + 8 * j) * BMW + col4) = ra[j];
    };
    auto commit_b = [&](int buf, int half) {
        float *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const f32x4 v = rv[jj][0] * rw[jj][0] + rv[jj][1] * rw[jj][1] + rv[jj][2] * rw[jj][2] + rv[jj][3] * rw[jj][3];
            *reinterpret_cast<f32x4 *>(B + (row + 8 * (2 * half + jj)) * BN + col4) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;
    auto mma = [&](int buf, int k2lo) {
        const float *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int k2 = k2lo; k2 < k2lo + BK / 4; ++k2) {
            const int kr = 2 * k2 + lh;
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = A[kr * BMW + (wm * TM + i) * 32 + lr];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = B[kr * BN + (wn * TN + j) * 32 + lr];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    load_om(kc_begin);
    issue_a(kc_begin);
    issue_b(kc_begin, 0);
    commit_b(0, 0);
    issue_b(kc_begin, 1);
    commit_b(0, 1);
    commit_a(0);
    load_om(kc_begin + 1);
    __syncthreads();
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int buf = (kc - kc_begin) & 1;
        const bool more = kc + 1 < kc_end;
        if (more) {
            issue_a(kc + 1);
            issue_b(kc + 1, 0);                       // geometry from the (dh, dw, mask) fetched one iteration ago
        }
        mma(buf, 0);
        if (more) {
            commit_b(buf ^ 1, 0);
            issue_b(kc + 1, 1);
        }
        mma(buf, BK / 4);
        if (more) {
            commit_b(buf ^ 1, 1);
            commit_a(buf ^ 1);
            load_om(kc + 2);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c0 + (wn * TN + j) * 32 + lr;
        if (c >= a.C) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ko = ko0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (ko < a.K) unsafeAtomicAdd(b.dw + ((long)ko * RS + tap) * a.C + c, acc[i][j][e]);
            }
    }
}

__global__ __launch_bounds__(256) void dcn_dgrad_kernel(const DcnBwdArgs b)
{
    const DcnArgs &a = b.a;
    constexpr int BN = 128, TM = 2, TN = 2, WN = 2, SST = 132;
    constexpr int A_ELEMS = BM * LDK, B_ELEMS = BK * BN;
    extern __shared__ __align__(16) float lds[];
    float *As = lds, *Bs = lds + 2 * A_ELEMS;
    float *stage = lds;                                // [BM][SST], aliases the GEMM tiles between two K loops
    float *geo_w = lds + 2 * A_ELEMS + 2 * B_ELEMS;    // [BM][8]: wt0..3 * 1, lh, lw, mask, unused
    int *geo_o = reinterpret_cast<int *>(geo_w + BM * 8);   // [BM][4] corner pixel index or -1
    float *red = reinterpret_cast<float *>(geo_o + BM * 4); // [BM][3] d mask, d offset h, d offset w

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int RS = a.R * a.S;
    const int m0 = dcn_xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int a_col = (t & 7) * 4, a_row = t >> 3;     // A: 32 pixel rows per pass x 32 ko
    const int b_row = t >> 5, b_col = (t & 31) * 4;    // B: 8 ko rows per pass x 128 channels
    const int nkc = (a.K + BK - 1) / BK;
    const int nct = (a.C + BN - 1) / BN;
    const int cpg = a.C / a.dg;
    const int pq = a.P * a.Q;
    f32x4 ra[4], rb[4];

    for (int tap = 0; tap < RS; ++tap) {
        const int ti = tap / a.S, tj = tap - ti * a.S;
        for (int ct = 0; ct < nct; ++ct) {
            const int c0 = ct * BN;
            const int g = c0 / cpg;
            if (ct == 0 || a.dg > 1) {                 // sample geometry of the tile's pixels for this tap (and group)
                __syncthreads();
                if (t < BM) {
                    const long m = (long)m0 + t;
                    int o[4] = {-1, -1, -1, -1};
                    float w4[4] = {0.f, 0.f, 0.f, 0.f}, flh = 0.f, flw = 0.f, mk = 0.f;
                    if (m < a.M) {
                        const int n = (int)(m / pq), rem = (int)(m - (long)n * pq), p = rem / a.Q, q = rem - p * a.Q;
                        const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                        mk = a.mask[m * (a.dg * RS) + g * RS + tap];
                        const float h = (float)(p * a.stride - a.pad_h + ti * a.dil) + po[0];
                        const float w = (float)(q * a.stride - a.pad_w + tj * a.dil) + po[1];
                        const bool inside = h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;
                        const float hf = floorf(h), wf = floorf(w);
                        const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
                        flh = h - hf; flw = w - wf;
                        const float hh = 1.f - flh, hw = 1.f - flw;
                        const int base = n * a.H * a.W;
                        if (inside && h0 >= 0 && w0 >= 0) { o[0] = base + h0 * a.W + w0; w4[0] = hh * hw; }
                        if (inside && h0 >= 0 && w1 <= a.W - 1) { o[1] = base + h0 * a.W + w1; w4[1] = hh * flw; }
                        if (inside && h1 <= a.H - 1 && w0 >= 0) { o[2] = base + h1 * a.W + w0; w4[2] = flh * hw; }
                        if (inside && h1 <= a.H - 1 && w1 <= a.W - 1) { o[3] = base + h1 * a.W + w1; w4[3] = flh * flw; }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { geo_o[t * 4 + e] = o[e]; geo_w[t * 8 + e] = w4[e]; }
                    geo_w[t * 8 + 4] = flh; geo_w[t * 8 + 5] = flw; geo_w[t * 8 + 6] = mk;
                    if (ct == 0) { red[t * 3] = 0.f; red[t * 3 + 1] = 0.f; red[t * 3 + 2] = 0.f; }
                }
                __syncthreads();
            }
            // ---- dcol tile = dY[m0.., :] x W[:, tap, c0..]
            auto issue = [&](int kc) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long m = (long)m0 + a_row + 32 * j;
                    const int ko = kc * BK + a_col;
                    ra[j] = *reinterpret_cast<const f32x4 *>(m < a.M && ko < a.K ? b.dy + m * a.K + ko : a.zero);
                    const int kb = kc * BK + b_row + 8 * j;
                    rb[j] = *reinterpret_cast<const f32x4 *>(kb < a.K && c0 + b_col < a.C
                                                                 ? a.w + ((long)kb * RS + tap) * a.C + c0 + b_col : a.zero);
                }
            };
            auto commit = [&](int buf) {
                float *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *reinterpret_cast<f32x4 *>(A + (a_row + 32 * j) * LDK + a_col) = ra[j];
                    *reinterpret_cast<f32x4 *>(B + (b_row + 8 * j) * BN + b_col) = rb[j];
                }
            };
            f32x16 acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            issue(0);
            commit(0);
            __syncthreads();
            for (int kc = 0; kc < nkc; ++kc) {
                const int buf = kc & 1;
                if (kc + 1 < nkc) issue(kc + 1);
                const float *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
                for (int kk = 0; kk < BK / 8; ++kk) {
                    f32x4 fa[TM];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[i] = *reinterpret_cast<const f32x4 *>(A + ((wm * TM + i) * 32 + lr) * LDK + kk * 8 + lh * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float fb[TN];
#pragma unroll
                        for (int j = 0; j < TN; ++j) fb[j] = B[(kk * 8 + lh * 4 + e) * BN + (wn * TN + j) * 32 + lr];
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j], acc[i][j], 0, 0, 0);
                    }
                }
                if (kc + 1 < nkc) commit(buf ^ 1);
                __syncthreads();
            }
            // ---- epilogue: accumulators -> LDS -> (pixel, channel) threads
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        stage[((wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * SST + (wn * TN + j) * 32 + lr] = acc[i][j][e];
            __syncthreads();
            // 16 passes of 8 pixel rows; a thread owns channels l32 + 32 i of its row.  The corner values of pass p+1
            // are fetched before pass p is processed (the passes are latency-bound gathers otherwise).
            const int r8 = t >> 5, l32 = t & 31;
            struct PassRegs { int o[4]; float w[4], flh, flw, mk; float xv[4][4]; };
            auto fetch = [&](int pass, PassRegs &R) {
                const int ml = pass * 8 + r8;
#pragma unroll
                for (int e = 0; e < 4; ++e) { R.o[e] = geo_o[ml * 4 + e]; R.w[e] = geo_w[ml * 8 + e]; }
                R.flh = geo_w[ml * 8 + 4]; R.flw = geo_w[ml * 8 + 5]; R.mk = geo_w[ml * 8 + 6];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + l32 + 32 * i;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        R.xv[e][i] = (R.o[e] >= 0 && c < a.C) ? a.x[(long)R.o[e] * a.C + c] : 0.f;
                }
            };
            auto process = [&](int pass, const PassRegs &R) {
                const int ml = pass * 8 + r8;
                const float hh = 1.f - R.flh, hw = 1.f - R.flw;
                const float dhw[4] = {-hw, -R.flw, hw, R.flw};      // d weight / d h of the four corners
                const float dww[4] = {-hh, hh, -R.flh, R.flh};      // d weight / d w
                float s_m = 0.f, s_h = 0.f, s_w = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + l32 + 32 * i;
                    const float gcol = stage[ml * SST + l32 + 32 * i];
                    const float gval = gcol * R.mk;
                    float val = 0.f, gh = 0.f, gw = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xv = R.xv[e][i];                // 0 for an invalid corner
                        val += R.w[e] * xv;
                        gh += dhw[e] * xv;
                        gw += dww[e] * xv;
                        // (a corner of weight 0 — integer sample positions, e.g. the zero-initialised offsets of a fresh DCN
                        // layer — adds nothing: no atomic for it)
                        if (R.o[e] >= 0 && R.w[e] != 0.f && c < a.C)
                            unsafeAtomicAdd(b.dx + (long)R.o[e] * a.C + c, gval * R.w[e]);
                    }
                    s_m += gcol * val;
                    s_h += gval * gh;
                    s_w += gval * gw;
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {
                    s_m += __shfl_xor(s_m, off, 64);
                    s_h += __shfl_xor(s_h, off, 64);
                    s_w += __shfl_xor(s_w, off, 64);
                }
                if (l32 == 0) { red[ml * 3] += s_m; red[ml * 3 + 1] += s_h; red[ml * 3 + 2] += s_w; }
            };
            PassRegs R0, R1;
            fetch(0, R0);
            for (int pass = 0; pass < BM / 8; pass += 2) {
                fetch(pass + 1, R1);
                process(pass, R0);
                if (pass + 2 < BM / 8) fetch(pass + 2, R0);
                process(pass + 1, R1);
            }
            __syncthreads();
            if ((ct == nct - 1 || a.dg > 1) && t < BM && (long)m0 + t < a.M) {
                // all channels of the group are in: store (dg > 1: one group per channel tile, cpg == BN multiples)
                const bool last_of_group = ((c0 + BN) % cpg) == 0 || ct == nct - 1;
                if (last_of_group) {
                    const long m = (long)m0 + t;
                    b.dmask[m * (a.dg * RS) + g * RS + tap] = red[t * 3];
                    b.doffset[m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap] = red[t * 3 + 1];
                    b.doffset[m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap + 1] = red[t * 3 + 2];
                    red[t * 3] = 0.f; red[t * 3 + 1] = 0.f; red[t * 3 + 2] = 0.f;
                }
            }
        }
    }
}

// ---- DCN module glue (ext/dcn/dcn_v2.py:117-121): the offset/mask convolution's 3*dg*R*S output channels are
// chunked in three; offset = cat(o1, o2) = the first two thirds unchanged, mask = sigmoid(last third).  One pass
// instead of torch.chunk + cat + sigmoid (two copies and an elementwise kernel); backward likewise.
__global__ void dcn_split_fwd_kernel(const float *om, long M, int third, float *offset, float *mask)
{
    const int ch = 3 * third;
    const long total = M * ch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / ch;
        const int c = (int)(i - m * ch);
        const float v = om[i];
        if (c < 2 * third) offset[m * 2 * third + c] = v;
        else mask[m * third + (c - 2 * third)] = 1.f / (1.f + expf(-v));
    }
}
__global__ void dcn_split_bwd_kernel(const float *doffset, const float *dmask, const float *mask, long M, int third, float *dom)
{
    const int ch = 3 * third;
    const long total = M * ch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / ch;
        const int c = (int)(i - m * ch);
        if (c < 2 * third) {
            dom[i] = doffset[m * 2 * third + c];
        } else {
            const float s = mask[m * third + (c - 2 * third)];
            dom[i] = dmask[m * third + (c - 2 * third)] * s * (1.f - s);
        }
    }
}

// ---- fused backward, data side, with on-chip pre-summation of d input (bf16 or fp32 operands) -------------------------
// dcn_dgrad_kernel adds every (pixel, tap, corner, channel) contribution to d input with its own global float atomic:
// 19.3 GB of atomic bytes at the config-4 layer, i.e. 14.8 ms at the chip's ≈1.3 TB/s atomic rate whatever the schedule.
// Here a workgroup owns a 2-D block of 8 x 16 output pixels; for each 32-channel chunk it computes the nine taps' column
// gradients in ONE sweep over K (dcol_tap[128 px][32 ch] = dY[128][K] x W[K][tap][32 ch]: nine accumulator tiles in
// registers, bf16 operands, fp32 accumulation), scatters the corner contributions into an LDS image of the input window the block
// can reach (fixed-point ds_add_u32 — ds_add_f32 is 26x slower on this part; pixel stride 33 words against bank
// conflicts) and flushes that window ONCE per chunk with global atomics: ≈3.3 window pixels per output pixel instead of 36 corner adds, 1.8 GB of atomic bytes instead of 19.3.
// Corners beyond the window margin go to global memory directly.  d offset / d mask are summed over the chunks in LDS and
// stored once.  Sample geometry per (pixel, tap) in LDS: (h0, w0) of the first corner, the fractions, the mask.
// MFMA 32x32x16 bf16 operand fragment (8 consecutive reduction indices k = k0 + 8 (lane / 32) .. +7 of column
// lane % 32) out of a ROW-MAJOR [k][32 columns] bf16 LDS image with 64-byte rows, through the hardware transpose read
// ds_read_b64_tr_b16 (checked lane by lane on the device: tools/tr_probe.hip): per 16-lane group a 4 x 16 block, lane
// 4q + p supplies the address of row q, columns 4p .. 4p+3, lane i receives column i.  Operands that arrive k-major from
// memory (weights [ko][c], pixels [px][c]) are stored with plain 8-byte writes instead of transposed 2-byte ones.  All
// 64 lanes must be active.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_tr_frag(const unsigned short *img, int k0, int lane)
{
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned short *a = img + (k0 + 8 * (g >> 1) + q) * 32 + 16 * (g & 1) + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a + 4 * 32));
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}

// Sum over the 8 lanes of an aligned lane group, on the VALU (DPP row_half_mirror, then the two quad permutes): every
// lane ends with the total.  __shfl_xor compiles to ds_bpermute — LDS traffic the epilogue below cannot afford.
__device__ __forceinline__ float dpp_sum8(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // lane i <- lane 7 - i
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4d __attribute__((ext_vector_type(4)));
// 64 lanes x 16 bytes global -> LDS (lane-linear from the wave-uniform byte address lds_addr), outside the compiler's
// memory model: the caller orders it with its own s_waitcnt vmcnt / s_barrier
__device__ __forceinline__ void dma16(i32x4d rsrc, unsigned lds_addr, unsigned voff, unsigned soff)
{
    // M0 (the LDS base of the transfer) is a register the compiler reserves for itself and cannot be named as clobbered: it is
    // saved and restored around the load, so nothing the compiler may keep there is lost
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
// float -> int, round to nearest with ties toward +inf (floor(x + 0.5)): ONE instruction where __float2int_rn takes two
// (v_rndne_f32 + v_cvt_i32_f32); the fixed-point window's rounding bound (half a unit per add) is the same.
__device__ __forceinline__ int cvt_rpi(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

struct DcnWinBwdArgs {
    DcnWinArgs w;
    const float *dy;
    float *dx, *doffset, *dmask;
    // bf16 data gradient: dY already rounded to bf16 (rr_dcn_dgrad_bf16_ws converts it once per call), or null.  The sweep
    // re-reads a block's dY tile once per 32-channel chunk; as 128 KB of fp32 per workgroup those re-reads miss the XCD's
    // 4 MB L2 (32 workgroups x 128 KB) and come from HBM eight times (4.3 of the 5.9 GB the kernel fetched in round 2);
    // as 64 KB of bf16 they stay in L2, and the staging needs no convert.
    const unsigned short *dyb;
    // DMA sweep: the weights packed to bf16 [tap][32-channel chunk][filter][32] (dcn_pack_weights_kernel, the forward's layout)
    const unsigned short *wpk;
    int goff_at;        // byte offset of the window-pixel offset table in dynamic LDS (behind everything else)
};

// F32: the matrix operands stay fp32 (v_mfma_f32_32x32x2_f32, K-steps of 16 filters: the same LDS bytes as 32 in bf16);
// everything after the GEMM sweep is shared.
// DMA (round 5; bf16 only, K % 32 == 0, dY given as a bf16 image, weights pre-packed): the weight operand of the K sweep
// and dY go global -> LDS by buffer_load ... lds into a ring of THREE operand images (weights [tap][32 ko][32 ch] exactly
// as packed: the transpose read needs no swizzle; dY [128 px][32 ko], 64-byte rows without padding, 16-byte chunks
// XOR-swizzled with (row >> 2) & 3 on the SOURCE side): no staging registers, converts or ds_writes, one barrier per
// K-step, loads two steps ahead.  Most of the ring lies over the d-input window, which holds nothing during the sweep (it
// is flushed and zero at every chunk boundary) and is zeroed again after it.  Before: 1.8 of the kernel's 5.3 ms were
// this sweep's staging (fp32 weight loads, converts, ds_writes, two barriers per step around 0.25 ms of MFMA).
template <int RS, bool F32, bool DMA = false>      // RS taps (9 for 3x3): one accumulator tile per tap lives in registers through a chunk
__global__ __launch_bounds__(512) void dcn_dgrad_win_kernel(const DcnWinBwdArgs wb)
{
    static_assert(!(DMA && F32), "the DMA sweep is bf16 only");
    // 512 threads = 8 waves, two per SIMD: the window leaves room for ONE workgroup per CU, and the epilogue is a chain
    // of dependent LDS / L2 round trips that a single wave per SIMD cannot hide.  Wave w computes pixel rows
    // 32 (w & 3) .. +31 of the block for taps 5 (w >> 2) .. +4 (five accumulator tiles; the ninth..tenth slot idles).
    const DcnWinArgs &wa = wb.w;
    const DcnArgs &a = wa.a;
    constexpr int CW = 32, WSTR = 33, SST = 36, NT = 512, TG = 5, GEO_SLOW = 1 << 20;
    constexpr int BKW = F32 ? 16 : 32;                        // filters per K-step
    constexpr int LDF = 20;                                   // fp32 operand rows: 16 + 4 floats (80 B: aligned, conflict-free b128 reads)
    constexpr int A_ELEMS = BM * LDKH;                        // in 2-byte units for both precisions (BM * LDF * 2 == BM * LDKH)
    static_assert(LDF * 2 == LDKH, "fp32 and bf16 operand images have the same size");
    extern __shared__ __align__(16) unsigned char smem[];
    const int npx = wa.WH * wa.WW;
    int *dxw = reinterpret_cast<int *>(smem);                                       // [npx][33] fixed point (see below)
    // [BM][RS] x {lh, lw, mask, window pixel of corner 0 | corner-valid bits << 16 | SLOW}: one ds_read_b128 per sample
    f32x4 *geo4 = reinterpret_cast<f32x4 *>(dxw + (size_t)((npx * WSTR + 3) & ~3));
    float *red = reinterpret_cast<float *>(geo4 + BM * RS);                         // [BM][RS][3]: d mask, d off h, d off w
    float *xw = red + BM * RS * 3;                                                  // [npx][32]: the input window of this chunk
    unsigned short *As = reinterpret_cast<unsigned short *>(xw + (size_t)npx * CW); // [BM][LDKH]   (single image: operands
    unsigned short *Bs = As + A_ELEMS;                                              // bf16: [RS][32 ko][32 ch]; fp32: [RS][32 ch][LDF] (prefetched in registers)
    float *stage = reinterpret_cast<float *>(As);                                   // [BM][SST]: epilogue only, over the idle operand images
    int *gofft = reinterpret_cast<int *>(smem + wb.goff_at);                        // [npx] (DMA sweep only): d-input element offset of a window pixel
    float *Af = reinterpret_cast<float *>(As), *Bf = reinterpret_cast<float *>(Bs);  // F32: [BM][LDF], [RS][32][LDF]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh_ = lane >> 5;
    const int wpx = wave & 3, wtg = wave >> 2;
    int bid = dcn_xcd_remap(blockIdx.x, gridDim.x);
    const int txi = bid % wa.tiles_x; bid /= wa.tiles_x;
    const int tyi = bid % wa.tiles_y;
    const int n = bid / wa.tiles_y;
    const int y0 = tyi * WIN_TH, x0 = txi * WIN_TW;
    const int wy0 = y0 - a.pad_h - wa.RW, wx0 = x0 - a.pad_w - wa.RW;
    const int cpt = a.C / CW, cpg = a.C / a.dg;
    const int nkc = (a.K + BKW - 1) / BKW;
    const int a_col = (t & 7) * 4, a_row = t >> 3;          // a_row 0..63
    const int b_row = a_row & 31, b_tg = a_row >> 5;        // weight rows: ko row b_row, taps 5 b_tg .. +4
    // F32 staging: dY 128 px x 16 ko = one float4 per thread; W 16 ko x 9 taps x 32 ch: ko row (t >> 3) & 15, taps 3 (t >> 7) .. +2
    const int fa_row = t >> 2, fa_col = (t & 3) * 4, fb_row = (t >> 3) & 15, fb_tg = t >> 7;
    const long img = (long)n * a.H * a.W;

    __shared__ int wmaxc[8][32];     // bits of max |dcol| per wave and channel
    __shared__ __align__(16) float fxs[32], fxi[32];   // fixed-point scale of the chunk's channels and its inverse (NaN: non-finite)
    __shared__ int mkmax_bits;       // bits of max |mask| seen by this block (over all groups: a bound is all that is needed)
    __shared__ int maxhits;          // most corner contributions any window pixel can receive from this block's samples
    // build_geo(g): sample geometry of deformable group g.  Also COUNTS, per window pixel, the corners that land on it
    // (in the window's unused pad word, column 32 of the 33-word pixel rows): the fixed-point scale below must hold
    // however many samples the offsets pile onto one input pixel (up to 128 x 9 x 4 in principle, ~36 for small offsets).
    // Called by all threads, between barriers of the caller; contains two barriers of its own.
    auto build_geo = [&](int g) {
        for (int px = t; px < npx; px += NT) dxw[px * WSTR + CW] = 0;
        if (t == 0) maxhits = 1;
        __syncthreads();
        for (int it = t; it < BM * RS; it += NT) {
            const int r = it / RS, tap = it - r * RS;
            const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
            int packed = 0;                       // no valid corner: nothing to add, zero gradients
            float flh = 0.f, flw = 0.f, mk = 0.f;
            if (p < a.P && q < a.Q) {
                const long m = ((long)n * a.P + p) * a.Q + q;
                const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                const int ti = tap / a.S, tj = tap - ti * a.S;
                const float h = (float)(p - a.pad_h + ti * a.dil) + po[0];
                const float w = (float)(q - a.pad_w + tj * a.dil) + po[1];
                if (h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W) {
                    const float hf = floorf(h), wf = floorf(w);
                    const int h0 = (int)hf, w0 = (int)wf;
                    int valid = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int hy = h0 + (e >> 1), wx = w0 + (e & 1);
                        if (hy >= 0 && hy <= a.H - 1 && wx >= 0 && wx <= a.W - 1) valid |= 1 << e;
                    }
                    // fast samples have their whole 2x2 footprint inside the window: corner e is window pixel
                    // base + (e >> 1) * WW + (e & 1); the others recompute their position and go corner by corner
                    const int ly = h0 - wy0, lx = w0 - wx0;
                    const bool fast = ly >= 0 && ly + 1 < wa.WH && lx >= 0 && lx + 1 < wa.WW;
                    packed = (valid << 16) | (fast ? (ly * wa.WW + lx) : GEO_SLOW);
                    flh = h - hf; flw = w - wf;
                    mk = a.mask[m * (a.dg * RS) + g * RS + tap];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int cy = ly + (e >> 1), cx = lx + (e & 1);
                        if (((valid >> e) & 1) && cy >= 0 && cy < wa.WH && cx >= 0 && cx < wa.WW)
                            atomicAdd(dxw + (cy * wa.WW + cx) * WSTR + CW, 1);
                    }
                }
            }
            geo4[it] = f32x4{flh, flw, mk, __int_as_float(packed)};
            atomicMax(&mkmax_bits, (int)(__float_as_uint(mk) & 0x7fffffffu));      // |mask| bound of the block (non-negative floats order as ints)
        }
        __syncthreads();
        int hits = 0;
        for (int px = t; px < npx; px += NT) hits = max(hits, dxw[px * WSTR + CW]);
        if (hits > 1) atomicMax(&maxhits, hits);
    };
    auto store_red = [&](int g) {
        for (int it = t; it < BM * RS; it += NT) {
            const int r = it / RS, tap = it - r * RS;
            const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
            if (p < a.P && q < a.Q) {
                const long m = ((long)n * a.P + p) * a.Q + q;
                wb.dmask[m * (a.dg * RS) + g * RS + tap] = red[it * 3];
                wb.doffset[m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap] = red[it * 3 + 1];
                wb.doffset[m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap + 1] = red[it * 3 + 2];
            }
            red[it * 3] = 0.f; red[it * 3 + 1] = 0.f; red[it * 3 + 2] = 0.f;
        }
    };
    for (int i = t; i < npx * WSTR; i += NT) dxw[i] = 0;
    for (int i = t; i < BM * RS * 3; i += NT) red[i] = 0.f;
    if (t == 0) mkmax_bits = 0;
    __syncthreads();
    int g_cur = 0;
    build_geo(0);
    __syncthreads();

    f32x4 ra[2], rb[TG];
    u16x4 rah[2];
    i32x4d rs_dy, rs_w, rs_x;
    unsigned dma_offa = 0;
    constexpr int XP = 6;                  // window pieces per wave: up to 6 x 8 x 8 = 384 window pixels
    constexpr int FP = 23;                 // flush pixels per thread: 16 x 23 = 368 (two fp32 windows of more pixels do not fit LDS anyway)
    const int wtg_u = __builtin_amdgcn_readfirstlane(wtg);
    if constexpr (DMA) {
        auto rsrc = [](const void *ptr, long bytes) {        // raw buffer descriptor: base, stride 0, bytes, raw 32-bit data format
            const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
            return i32x4d{(int)__builtin_amdgcn_readfirstlane((unsigned)u), (int)(__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) & 0xffffu),
                          __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
        };
        rs_x = rsrc(a.x, (long)a.N * a.H * a.W * a.C * 4);
        // window pixel -> element offset of its first channel in d input (-1 outside the image): the flush's table
        for (int px = t; px < npx; px += NT) {
            const int ly = px / wa.WW, lx = px - ly * wa.WW;
            const int gy = wy0 + ly, gx = wx0 + lx;
            gofft[px] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (int)((img + (long)gy * a.W + gx) * a.C) : -1;
        }
        rs_dy = rsrc(wb.dyb, (long)a.M * a.K * 2);
        rs_w = rsrc(wb.wpk, (long)RS * a.C * a.K * 2);
        // this lane's 16 bytes of the wave's dY piece: pixel row 16 wave + lane / 4, slot lane % 4 <- chunk slot ^ swizzle(row)
        // (64-byte rows without padding: the swizzle (row >> 2) & 3 spreads a ds_read_b128 group over the banks)
        const int row = wave * 16 + (lane >> 2);
        const int p = y0 + row / WIN_TW, q = x0 + row % WIN_TW;
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        dma_offa = (p < a.P && q < a.Q) ? (unsigned)(((((long)n * a.P + p) * a.Q + q) * a.K + chunk * 8) * 2) : 0x80000000u;   // beyond the buffer: zeros
    }
    for (int cch = 0; cch < cpt; ++cch) {
        const int c0 = cch * CW;
        const int g = c0 / cpg;
        if (g != g_cur) {                       // (dg > 1) new deformable group: store its predecessor's sums, new geometry
            __syncthreads();
            store_red(g_cur);
            g_cur = g;
            build_geo(g);
            __syncthreads();
        }
        // ---- this chunk's input window (d offset / d mask re-read the four corners of every sample: from L2 that is
        //      19 GB per call, the same gather the windowed forward removed)
        if constexpr (DMA) {
            // asynchronously, 8 pixels x 128 bytes per wave-instruction; lands under the sweep (whose last step waits for everything)
            typedef __attribute__((address_space(3))) void lds_void;
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            {
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int piece = wv + 8 * i;
                const int px = piece * 8 + (lane >> 3);
                if (piece * 8 < npx && px < npx) {          // lanes past the window's last pixel stay out (exec mask)
                    const int go = gofft[px];                // (written before the first chunk's barriers)
                    dma16(rs_x, __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)(xw + piece * 256)),
                          go >= 0 ? (unsigned)go * 4u + (unsigned)(lane & 7) * 16u : 0x80000000u, (unsigned)(c0 * 4));      // outside the image: zeros
                }
            }
            }
        } else
        for (int i = t; i < npx * 8; i += NT) {
            const int px = i >> 3, c4 = (i & 7) * 4;
            const int ly = px / wa.WW, lx = px - ly * wa.WW;
            const int gy = wy0 + ly, gx = wx0 + lx;
            const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            *reinterpret_cast<f32x4 *>(xw + (size_t)i * 4) =
                *reinterpret_cast<const f32x4 *>(ok ? a.x + (img + (long)gy * a.W + gx) * a.C + c0 + c4 : a.zero);
        }
        // ---- the nine column-gradient tiles of this chunk in ONE sweep over K:
        //      dcol_tap[128 px][32 ch] = dY[128][K] x W[K][tap][c0 .. c0+31]
        auto issue = [&](int kc) {
            if constexpr (F32) {
                const int p = y0 + fa_row / WIN_TW, q = x0 + fa_row % WIN_TW;
                const int ko = kc * BKW + fa_col;
                ra[0] = *reinterpret_cast<const f32x4 *>((p < a.P && q < a.Q && ko < a.K)
                                                             ? wb.dy + (((long)n * a.P + p) * a.Q + q) * a.K + ko : a.zero);
                const int kb = kc * BKW + fb_row;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int tap = fb_tg * 3 + i;
                    rb[i] = *reinterpret_cast<const f32x4 *>((kb < a.K && tap < RS) ? a.w + ((long)kb * RS + tap) * a.C + c0 + a_col : a.zero);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = a_row + 64 * j;
                    const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
                    const int ko = kc * BKW + a_col;
                    const bool ok = p < a.P && q < a.Q && ko < a.K;
                    const long idx = (((long)n * a.P + p) * a.Q + q) * a.K + ko;
                    if (wb.dyb != nullptr)
                        rah[j] = *reinterpret_cast<const u16x4 *>(ok ? (const void *)(wb.dyb + idx) : (const void *)a.zero);
                    else
                        ra[j] = *reinterpret_cast<const f32x4 *>(ok ? wb.dy + idx : a.zero);
                }
                const int kb = kc * BKW + b_row;         // one ko row per 8 threads, 4 channels each, five taps
#pragma unroll
                for (int i = 0; i < TG; ++i) {
                    const int tap = b_tg * TG + i;
                    rb[i] = *reinterpret_cast<const f32x4 *>((kb < a.K && tap < RS) ? a.w + ((long)kb * RS + tap) * a.C + c0 + a_col : a.zero);
                }
            }
        };
        auto commit = [&]() {
            if constexpr (F32) {
                *reinterpret_cast<f32x4 *>(Af + fa_row * LDF + fa_col) = ra[0];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int tap = fb_tg * 3 + i;
                    if (tap < RS) {                  // B images are [tap][n = channel][k = ko]: transposed 4-byte stores
#pragma unroll
                        for (int c = 0; c < 4; ++c) Bf[(tap * CW + a_col + c) * LDF + fb_row] = rb[i][c];
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<u16x4 *>(As + (a_row + 64 * j) * LDKH + a_col) = wb.dyb != nullptr ? rah[j] : f2bf4(ra[j]);
#pragma unroll
                for (int i = 0; i < TG; ++i) {
                    const int tap = b_tg * TG + i;
                    if (tap < RS) {
                        // B images are [tap][k = ko][n = channel], as the weights lie in memory: the MFMA fragment
                        // comes out of the transpose read
                        *reinterpret_cast<u16x4 *>(Bs + (tap * 32 + b_row) * CW + a_col) = f2bf4(rb[i]);
                    }
                }
            }
        };
        f32x16 acc[TG];
#pragma unroll
        for (int i = 0; i < TG; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        if constexpr (DMA) {
            typedef __attribute__((address_space(3))) void lds_void;
            // Operand images: a ring of THREE (dY [128 px][32 ko] 8 KB + weights [tap][32 ko][32 ch] 18 KB each), filled two
            // K-steps ahead.  Image 0 and the third dY image lie in the operand area, the others over the empty d-input
            // window.  The loads are inline assembly: as builtins every later LDS read would wait for them (the compiler
            // cannot tell that the images do not alias) and __syncthreads would wait for everything.
            unsigned short *const dxw16 = reinterpret_cast<unsigned short *>(dxw);
            unsigned short *const Aimg[3] = {As, dxw16, As + (BM * 32 + RS * 32 * CW)};
            unsigned short *const Bimg[3] = {As + BM * 32, dxw16 + BM * 32, dxw16 + (BM * 32 + RS * 32 * CW)};
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            auto lds_addr = [](const unsigned short *ptr) { return (unsigned)(size_t)(lds_void *)ptr; };
            // per K-step: piece w of dY (16 pixel rows), weight pieces w, w + 8, w + 16 (half a tap's tile each; 18 in all)
            auto issue = [&](int kc, int st) {
                dma16(rs_dy, __builtin_amdgcn_readfirstlane(lds_addr(Aimg[st] + wv * 512)), dma_offa, (unsigned)(kc * 64));
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int b = wv + 8 * i;
                    if (b < 2 * RS) {
                        const unsigned so = (unsigned)((((((long)(b >> 1)) * cpt + cch) * a.K + kc * 32 + (b & 1) * 16) * 32) * 2);
                        dma16(rs_w, __builtin_amdgcn_readfirstlane(lds_addr(Bimg[st] + b * 512)), (unsigned)lane * 16u,
                              __builtin_amdgcn_readfirstlane(so));
                    }
                }
            };
            const int nk = nkc;
            if (nk > 0) issue(0, 0);
            if (nk > 1) issue(1, 1);
            const int row = wpx * 32 + lr;
            for (int kc0 = 0; kc0 < nk; kc0 += 3) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int kc = kc0 + u;
                    if (kc < nk) {                       // uniform
                        // step kc's pieces landed: only step kc + 1's (3 or 4 per wave) may still fly
                        if (kc + 1 < nk) {
                            if (wv < 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                        __builtin_amdgcn_s_barrier();    // images kc % 3 visible; images (kc + 2) % 3 = (kc - 1) % 3 free again
                        if (kc + 2 < nk) issue(kc + 2, (u + 2) % 3);
                        const unsigned short *A = Aimg[u], *B = Bimg[u];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(A + row * 32 + (((kk * 2 + lh_) ^ ((row >> 2) & 3)) * 8));
                            // taps i < RS - TG exist in both wave groups: their reads go out together, the MFMAs back to back
                            constexpr int NV = RS - TG;
                            bf16x8 fb[TG];
#pragma unroll
                            for (int i = 0; i < NV; ++i) fb[i] = lds_tr_frag(B + (wtg_u * TG + i) * 32 * CW, kk * 16, lane);
                            if (wtg_u == 0) {
#pragma unroll
                                for (int i = NV; i < TG; ++i) fb[i] = lds_tr_frag(B + i * 32 * CW, kk * 16, lane);
                            }
#pragma unroll
                            for (int i = 0; i < NV; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[i], acc[i], 0, 0, 0);
                            if (wtg_u == 0) {
#pragma unroll
                                for (int i = NV; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[i], acc[i], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
        issue(0);
        for (int kc = 0; kc < nkc; ++kc) {
            commit();
            __syncthreads();
            if (kc + 1 < nkc) issue(kc + 1);         // lands under this K-step's MFMAs
            if constexpr (F32) {
                // lane half lh_ takes ko 8 lh_ .. 8 lh_ + 7 of the step (any pairing of ko values across the halves is a
                // valid 32x32x2 product as long as A and B agree): two ds_read_b128 per operand
                const f32x4 fa0 = *reinterpret_cast<const f32x4 *>(Af + (wpx * 32 + lr) * LDF + 8 * lh_);
                const f32x4 fa1 = *reinterpret_cast<const f32x4 *>(Af + (wpx * 32 + lr) * LDF + 8 * lh_ + 4);
#pragma unroll
                for (int i = 0; i < TG; ++i) {
                    const int tap = wtg * TG + i;
                    if (tap < RS) {                   // wave-uniform
                        const f32x4 fb0 = *reinterpret_cast<const f32x4 *>(Bf + (tap * CW + lr) * LDF + 8 * lh_);
                        const f32x4 fb1 = *reinterpret_cast<const f32x4 *>(Bf + (tap * CW + lr) * LDF + 8 * lh_ + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb0[e], acc[i], 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb1[e], acc[i], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < BK / 16; ++kk) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(As + (wpx * 32 + lr) * LDKH + kk * 16 + lh_ * 8);
#pragma unroll
                    for (int i = 0; i < TG; ++i) {
                        const int tap = wtg * TG + i;
                        if (tap < RS) {                   // wave-uniform
                            const bf16x8 fb = lds_tr_frag(Bs + tap * 32 * CW, kk * 16, lane);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
        }
        // ---- the window accumulates in FIXED POINT: ds_add_f32 costs ~190 cycles per wave-instruction on this part
        //      (3 cycles per lane, serialised across the waves of a CU; measured), ds_add_u32 ~7.  Scale (per channel) =
        //      the power of two that keeps `maxhits` contributions (the most any window pixel receives, counted in
        //      build_geo) of the block's largest |dcol| x |mask| of that channel below 2^30: absolute rounding <= 2^-31 of that bound per add, exact scaling back at the flush.
        // One scale PER CHANNEL (a lane of the accumulator tiles is one channel): a channel with small gradients keeps its
        // own 25 bits next to a loud one in the same chunk, as it would with float atomics.
        // max |dcol| on the BIT patterns (non-negative floats order as integers, and NaN > inf > finite: fmaxf would
        // drop a NaN silently)
        int amax = 0;
#pragma unroll
        for (int i = 0; i < TG; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) amax = max(amax, (int)(__float_as_uint(acc[i][e]) & 0x7fffffffu));
        amax = max(amax, __shfl_xor(amax, 32, 64));          // the two lane halves hold the same channel
        if (lh_ == 0) wmaxc[wave][lr] = amax;
        __syncthreads();
        if constexpr (DMA) {                // every wave is past the sweep: the window under the second operand image back to zero
            constexpr int WORDS4 = (BM * 32 + 2 * RS * 32 * CW) * 2 / 16;           // dY image 1, weight images 1 and 2
            for (int i = t; i < WORDS4; i += NT) reinterpret_cast<f32x4 *>(dxw)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (t < CW) {
            int bbits = wmaxc[0][t];
#pragma unroll
            for (int i = 1; i < 8; ++i) bbits = max(bbits, wmaxc[i][t]);
            const float bound = (float)maxhits * __int_as_float(bbits) * fmaxf(__int_as_float(mkmax_bits), 1e-30f);
            // a non-finite column gradient or mask (diverged training) must stay visible: fixed point cannot carry it,
            // so the channel's whole window receives NaN at the flush (inverse scale = NaN marks it); incl. an
            // overflowing bound
            const bool nonfinite = bbits > 0x7f7fffff || mkmax_bits > 0x7f7fffff || !(bound <= 3.0e38f);
            int ex = 0;
            frexpf(bound, &ex);
            // the scale is a power of two 2^sh with sh <= 126: for bounds below 2^-96 (vanishing gradients) 2^(30 - ex)
            // itself would overflow to inf and the integer adds would saturate
            const int sh = 30 - ex > 126 ? 126 : 30 - ex;
            fxs[t] = (bound > 0.f && !nonfinite) ? ldexpf(1.f, sh) : 0.f;
            fxi[t] = nonfinite ? __int_as_float(0x7fc00000) : ldexpf(1.f, -sh);
        }
        __syncthreads();
        const f32x4 fx4 = *reinterpret_cast<const f32x4 *>(fxs + a_col);      // this thread's four channels
        // ---- epilogue: accumulators -> LDS -> (pixel row, 4 channels) threads, TWO taps per barrier pair (round 5): at
        //      step i the waves of tap group 0 stage tap i and those of group 1 tap TG + i — every wave writes, five
        //      barrier pairs per chunk instead of nine.  Stage rows are 32 floats without padding (a wave's b128 reads
        //      cover 1 KB contiguously; the two lane halves of a write hit the same banks, which a 64-lane b32 write
        //      pays anyway): two images fit the idle operand area.
        constexpr int SST2 = 32;
        for (int step = 0; step < TG; ++step) {        // a real loop: only the accumulator -> LDS copy is per-step code
            if (wtg * TG + step < RS) {
                float *sp = stage + wtg * (BM * SST2) + (wpx * 32 + 4 * lh_) * SST2 + lr;
#define RR_PUT(T)                                                                                    \
    case T:                                                                                          \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) sp[((e & 3) + 8 * (e >> 2)) * SST2] = acc[T][e]; \
        break;
                switch (step) {
                    RR_PUT(0) RR_PUT(1) RR_PUT(2) RR_PUT(3) RR_PUT(4)
                    default: break;
                }
#undef RR_PUT
            }
            __syncthreads();
            // Items of this step: (tap u TG + step, pixel row a_row + 64 j), k = 2 u + j.  Software-pipelined: the sample
            // geometry of all items first, then item k + 1's LDS reads (column gradient, four corner rows of x) go out
            // BEFORE item k's sixteen ds_adds — a read issued behind them waits until the LDS queue has drained them.
            const int nit = (TG + step < RS) ? 4 : 2;
            struct ItemLd { f32x4 gcol; f32x4 xc[4]; };
            f32x4 gq4[4];                                          // flh, flw, mask, packed corner word
            auto load_geo = [&](int k) { gq4[k] = geo4[(a_row + 64 * (k & 1)) * RS + (k >> 1) * TG + step]; };
            auto load_item = [&](int k, ItemLd &L) {
                const int r = a_row + 64 * (k & 1);
                const int base = __float_as_int(gq4[k][3]) & 0xffff;          // slow samples: 0 (their reads are not used)
                L.gcol = *reinterpret_cast<const f32x4 *>(stage + (k >> 1) * (BM * SST2) + r * SST2 + a_col);
                const float *xb = xw + (size_t)base * CW + a_col;
#pragma unroll
                for (int e = 0; e < 4; ++e) L.xc[e] = *reinterpret_cast<const f32x4 *>(xb + ((e >> 1) * wa.WW + (e & 1)) * CW);
            };
            auto process = [&](int k, const ItemLd &L) {
                    const int tap = (k >> 1) * TG + step;
                    const int r = a_row + 64 * (k & 1);
                    const f32x4 gq = gq4[k];
                    const int gi = __float_as_int(gq[3]);
                    const float flh = gq[0], flw = gq[1], mk = gq[2];
                    const f32x4 gcol = L.gcol;
                    float s_m = 0.f, s_h = 0.f, s_w = 0.f;
                    if (!(gi & GEO_SLOW)) {
                        // Fast samples (the whole 2x2 footprint inside the window; also samples without any valid corner:
                        // gi == 0, mask 0): BRANCH-FREE and, since round 5, MASK-FREE.  A corner outside the image reads a
                        // window pixel that was filled with zeros (its dot product is 0) and adds into a window pixel the
                        // flush discards — no per-corner validity selects.  Two channels per packed-fp32 instruction.
                        const float hh = 1.f - flh, hw = 1.f - flw;
                        const f32x2 wcol = {hw, flw};
                        const f32x2 w01 = wcol * hh, w23 = wcol * flh;       // corner weights (top row, bottom row)
                        const f32x2 g01 = {gcol[0], gcol[1]}, g23 = {gcol[2], gcol[3]};
                        const f32x2 q01 = g01 * (f32x2{fx4[0], fx4[1]} * mk);   // column gradient x mask in each channel's fixed-point unit
                        const f32x2 q23 = g23 * (f32x2{fx4[2], fx4[3]} * mk);
                        int *db = dxw + (gi & 0xffff) * WSTR + a_col;
                        float d[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int po = (e >> 1) * wa.WW + (e & 1);
                            const f32x4 xv = L.xc[e];
                            const f32x2 dd = g01 * f32x2{xv[0], xv[1]} + g23 * f32x2{xv[2], xv[3]};
                            d[e] = dd[0] + dd[1];
                            const float we = e == 0 ? w01[0] : e == 1 ? w01[1] : e == 2 ? w23[0] : w23[1];
                            const f32x2 c01 = q01 * we, c23 = q23 * we;
                            atomicAdd(db + po * WSTR + 0, cvt_rpi(c01[0]));      // ds_add_u32
                            atomicAdd(db + po * WSTR + 1, cvt_rpi(c01[1]));
                            atomicAdd(db + po * WSTR + 2, cvt_rpi(c23[0]));
                            atomicAdd(db + po * WSTR + 3, cvt_rpi(c23[1]));
                        }
                        // d mask = sum_e wt_e <gcol, x_e>, d offset = mask * sum_e dwt_e <gcol, x_e>: one dot product per corner
                        s_m = w01[0] * d[0] + w01[1] * d[1] + w23[0] * d[2] + w23[1] * d[3];
                        s_m = gi ? s_m : 0.f;            // a sample wholly outside the image (gi == 0, mask 0) sits on window pixel 0, which holds data
                        s_h = mk * (hw * (d[2] - d[0]) + flw * (d[3] - d[1]));
                        s_w = mk * (hh * (d[1] - d[0]) + flh * (d[3] - d[2]));
                    } else if ((gi >> 16) & 15) {
                        const int valid = (gi >> 16) & 15;
                        const float hh = 1.f - flh, hw = 1.f - flw;
                        const float wt[4] = {hh * hw, hh * flw, flh * hw, flh * flw};
                        const float dhw[4] = {-hw, -flw, hw, flw};
                        const float dww[4] = {-hh, hh, -flh, flh};
                                                const f32x4 gs = gcol * fx4;                           // column gradient in each channel's fixed-point unit
                        // footprint not inside the window (offset beyond the margin): position again from the offsets
                        const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
                        const float *po = a.offset + (((long)n * a.P + p) * a.Q + q) * (2 * a.dg * RS) + g_cur * 2 * RS + 2 * tap;
                        const int ti = tap / a.S, tj = tap - ti * a.S;
                        const int h0 = (int)floorf((float)(p - a.pad_h + ti * a.dil) + po[0]);
                        const int w0 = (int)floorf((float)(q - a.pad_w + tj * a.dil) + po[1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (!((valid >> e) & 1)) continue;
                            const int hy = h0 + (e >> 1), wx = w0 + (e & 1);
                            const int ly = hy - wy0, lx = wx - wx0;
                            const bool inwin = ly >= 0 && ly < wa.WH && lx >= 0 && lx < wa.WW;
                            f32x4 xv;
                            if (inwin) xv = *reinterpret_cast<const f32x4 *>(xw + (size_t)(ly * wa.WW + lx) * CW + a_col);
                            else xv = *reinterpret_cast<const f32x4 *>(a.x + (img + (long)hy * a.W + wx) * a.C + c0 + a_col);
                            const float dd = gcol[0] * xv[0] + gcol[1] * xv[1] + gcol[2] * xv[2] + gcol[3] * xv[3];
                            s_m += wt[e] * dd; s_h += dhw[e] * dd; s_w += dww[e] * dd;
                            if (wt[e] != 0.f) {
                                if (inwin) {
                                    int *d4 = dxw + (ly * wa.WW + lx) * WSTR + a_col;
#pragma unroll
                                    for (int c = 0; c < 4; ++c) atomicAdd(d4 + c, cvt_rpi(gs[c] * (mk * wt[e])));
                                } else {
                                    float *d4 = wb.dx + (img + (long)hy * a.W + wx) * a.C + c0 + a_col;
#pragma unroll
                                    for (int c = 0; c < 4; ++c) unsafeAtomicAdd(d4 + c, gcol[c] * (mk * wt[e]));
                                }
                            }
                        }
                        s_h *= mk; s_w *= mk;
                    }
                    s_m = dpp_sum8(s_m); s_h = dpp_sum8(s_h); s_w = dpp_sum8(s_w);      // the 8 threads of a pixel row
                    if ((t & 7) < 3) {                  // lanes 0 / 1 / 2 of the row add d mask / d offset h / d offset w
                        const float v = (t & 7) == 0 ? s_m : (t & 7) == 1 ? s_h : s_w;
                        red[(r * RS + tap) * 3 + (t & 7)] += v;
                    }
            };
            if constexpr (DMA) {
                if (nit > 0) {
                    ItemLd L0, L1;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < nit) load_geo(k);
                    load_item(0, L0);
                    load_item(1, L1);
                    process(0, L0);
                    if (nit > 2) load_item(2, L0);
                    process(1, L1);
                    if (nit > 2) {
                        load_item(3, L1);
                        process(2, L0);
                        process(3, L1);
                    }
                }
            } else {                        // (the staging registers of the plain sweep leave no room for a second item in flight)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < nit) {
                        ItemLd L;
                        load_geo(k);
                        load_item(k, L);
                        process(k, L);
                    }
                }
            }
            __syncthreads();        // the stage images are rewritten by the next step
        }
        // ---- flush this chunk's window: lane <-> channel, 128-byte row segments, one global atomic per touched element
        {
            const int c = t & 31;
            const float fx_inv = fxi[c];
            const bool nonfinite = fx_inv != fx_inv;
            if constexpr (DMA) {
                // the pixel positions do not depend on the chunk: element offsets computed once per workgroup (goff), LDS
                // addresses with immediate offsets
                {
                float *const dxc = wb.dx + c0 + c;
                int *const dw = dxw + (t >> 5) * WSTR + c;
#pragma unroll
                for (int i = 0; i < FP; ++i) {
                    if ((t >> 5) + 16 * i < npx) {
                        const int iv = dw[i * 16 * WSTR];
                        const int go = gofft[(t >> 5) + 16 * i];
                        if (iv != 0) {
                            if (go >= 0) unsafeAtomicAdd(dxc + go, (float)iv * fx_inv);
                            dw[i * 16 * WSTR] = 0;
                        } else if (nonfinite && go >= 0) {
                            unsafeAtomicAdd(dxc + go, __int_as_float(0x7fc00000));
                        }
                    }
                }
                }
            } else
            for (int px = t >> 5; px < npx; px += NT / 32) {
                const int iv = dxw[px * WSTR + c];
                const int ly = px / wa.WW, lx = px - ly * wa.WW;
                const int gy = wy0 + ly, gx = wx0 + lx;
                const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                if (iv != 0) {                                   // window pixels outside the image collect sums nobody wants (mask-free fast path)
                    if (inside) unsafeAtomicAdd(wb.dx + (img + (long)gy * a.W + gx) * a.C + c0 + c, (float)iv * fx_inv);
                    dxw[px * WSTR + c] = 0;
                } else if (nonfinite && inside) {
                    unsafeAtomicAdd(wb.dx + (img + (long)gy * a.W + gx) * a.C + c0 + c, __int_as_float(0x7fc00000));
                }
            }
        }
        __syncthreads();
    }
    store_red(g_cur);
}

// ---- fused backward, weight side, on the forward's LDS window (bf16 or fp32 operands) ---------------------------------
// dW[ko][tap][c] = sum_px dY[px][ko] * col[px][tap, c].  A workgroup owns ONE 32-channel chunk and 256 filters — the
// whole [256 filters][9 taps x 32 channels] block of dW in its accumulators (8 waves x 9 tiles of 32x32) — and walks
// over its share of the 8x16 pixel blocks: per block the input window of the chunk goes to LDS once (as in the
// forward), per half block (64 pixels) dY is rounded to bf16 and written TRANSPOSED ([filter][pixel]: the pixel is the
// reduction index, 8 consecutive pixels = one MFMA operand register pair) and the masked bilinear samples likewise
// ([tap][channel][pixel]).  36 MFMAs per wave and half block; the blocks of a split are shared by the C/32 chunk
// workgroups, which sit on one XCD so that dY is read from HBM once.  dW leaves with one float atomic per element and
// workgroup (75 MB per call at the config-4 layer).
struct DcnWinWgradArgs {
    DcnWinArgs w;
    const float *dy;
    float *dw;
    int splits, ktiles;
    int dma_window;     // x is addressable with 32-bit byte offsets: the window goes global -> LDS by DMA
    const unsigned short *dyb;      // DYDMA: dY's bf16 image
};

// F32: fp32 operands (v_mfma_f32_32x32x2_f32) in sub-blocks of 32 pixels — the same LDS bytes as 64 pixels of bf16.
// DYDMA (round 5; bf16, K % 32 == 0, dY given as a bf16 image): wave w's dY operand (filters 32 w .. +31 of the tile) goes
// global -> LDS by buffer_load ... lds into a PRIVATE ring of six 16-pixel slices (1 KB each, [pixel][32 filters] as dY lies
// in memory: the transpose read makes the operand) — no staging registers, converts, ds_writes, and no barrier for this
// operand: the next sub-block's four slices are issued while this one's are consumed.
template <int RS, bool F32, bool DYDMA = false>
__global__ __launch_bounds__(512) void dcn_wgrad_win_kernel(const DcnWinWgradArgs wb)
{
    static_assert(!(DYDMA && F32), "the dY ring is bf16 only");
    const DcnWinArgs &wa = wb.w;
    const DcnArgs &a = wa.a;
    constexpr int CW = 32, NT = 512, GEO_SLOW = 1 << 20, KT = 256;
    constexpr int HP = F32 ? 32 : 64;                          // pixels per sub-block
    constexpr int LDP = F32 ? HP + 4 : HP + 8;                 // operand row length in elements (144 B rows either way)
    constexpr int NSUB = BM / HP;
    extern __shared__ __align__(16) unsigned char smem[];
    const int npx = wa.WH * wa.WW;
    float *xw = reinterpret_cast<float *>(smem);                                    // [npx][32]
    f32x4 *geo4 = reinterpret_cast<f32x4 *>(xw + (size_t)npx * CW);                 // [BM][RS] x {lh, lw, mask, packed corner word}: as in dcn_dgrad_win_kernel
    unsigned short *dyT = reinterpret_cast<unsigned short *>(geo4 + BM * RS);       // [256 filters][LDP pixels]
    unsigned short *colT = dyT + (DYDMA ? 8 * 6 * 512 : KT * 72);                   // [RS][32 channels][LDP pixels]; DYDMA: dyT = 8 waves x 6 slices x [16 px][32 filters]
    float *dyF = reinterpret_cast<float *>(dyT), *colF = reinterpret_cast<float *>(colT);   // F32 images, same bytes

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh_ = lane >> 5;
    const int cpt = a.C / CW, cpg = a.C / a.dg;
    // blockIdx -> (split, chunk, filter tile): the chunk workgroups of one split share an XCD (blockIdx & 7)
    int split, rest;
    if ((wb.splits & 7) == 0) { split = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / (cpt * wb.ktiles)); rest = (blockIdx.x >> 3) % (cpt * wb.ktiles); }
    else { split = blockIdx.x / (cpt * wb.ktiles); rest = blockIdx.x % (cpt * wb.ktiles); }
    const int cch = rest % cpt, kt = rest / cpt;
    const int c0 = cch * CW, g = c0 / cpg, k0 = kt * KT;
    // sample builder: pixel a_row of the sub-block, 4 channels; bf16: all 9 taps, F32: taps 5 (t >> 8) .. +4
    const int a_col = (t & 7) * 4, a_row = F32 ? (t >> 3) & 31 : t >> 3, tap0 = F32 ? 5 * (t >> 8) : 0;
    constexpr int NTAP = F32 ? 5 : RS, PG = F32 ? 4 : 8;
    const int fq = t & 63, pg = t >> 6;                     // dY stager: filters 4 fq .. +3, pixels PG pg .. +PG-1 of the sub-block
    const int ntiles = a.N * wa.tiles_y * wa.tiles_x;

    f32x16 acc[RS];
#pragma unroll
    for (int tap = 0; tap < RS; ++tap)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tap][e] = 0.f;

    f32x4 rdy[DYDMA ? 1 : PG];
    auto issue_dy = [&](int tile, int sub) {                 // PG pixels x 4 filters, a 1 KB row segment per wave and pixel
        if constexpr (DYDMA) return;
        int bid = tile;
        const int txi = bid % wa.tiles_x; bid /= wa.tiles_x;
        const int tyi = bid % wa.tiles_y;
        const int n = bid / wa.tiles_y;
        const int r0 = sub * HP + PG * pg;
        const int p = tyi * WIN_TH + r0 / WIN_TW, q0 = txi * WIN_TW + r0 % WIN_TW;
        const int f = k0 + 4 * fq;
#pragma unroll
        for (int i = 0; i < (DYDMA ? 1 : PG); ++i)
            rdy[i] = *reinterpret_cast<const f32x4 *>((p < a.P && q0 + i < a.Q && f < a.K)
                                                          ? wb.dy + (((long)n * a.P + p) * a.Q + q0 + i) * a.K + f : a.zero);
    };
    auto commit_dy = [&]() {
        if constexpr (DYDMA) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (F32) {
                const f32x4 v = {rdy[0][j], rdy[1][j], rdy[2][j], rdy[3][j]};
                *reinterpret_cast<f32x4 *>(dyF + (4 * fq + j) * LDP + 4 * pg) = v;
            }
        }
        if constexpr (!F32) {
            // bf16: [filter slab of 32][pixel][32 filters], as dY lies in memory; the MFMA fragment (8 consecutive
            // pixels of one filter) comes out of the transpose read
#pragma unroll
            for (int i = 0; i < (DYDMA ? 1 : 8); ++i)
                *reinterpret_cast<u16x4 *>(dyT + ((fq >> 3) * HP + 8 * pg + i) * 32 + 4 * (fq & 7)) = f2bf4(rdy[i]);
        }
    };

    // Input window of a block and this workgroup's chunk: global -> LDS by buffer_load ... lds (8 pixels x 128 bytes per
    // wave-instruction, zeros outside the image), and the sample geometry of deformable group g.  Both are prepared for the
    // NEXT block as soon as the last samples of the current one are built — under its last MFMA phase (round 5; before,
    // every block started with a synchronous window copy and the geometry's dependent global reads: ~0.9 of 2.34 ms).
    typedef __attribute__((address_space(3))) void lds_void;
    const i32x4d rs_x = [&] {
        const unsigned long long u = reinterpret_cast<unsigned long long>(a.x);
        return i32x4d{(int)__builtin_amdgcn_readfirstlane((unsigned)u), (int)(__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) & 0xffffu),
                      __builtin_amdgcn_readfirstlane((int)((long)a.N * a.H * a.W * a.C * 4)), 0x00020000};
    }();
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    constexpr int GI = (BM * RS + NT - 1) / NT;            // geometry items per thread (3)
    float g_oh[GI], g_ow[GI], g_mk[GI];                    // offsets and mask of the next block's items, loaded under the MFMAs
    auto prepare_issue = [&](int tl) {
        int bid = tl;
        const int txi = bid % wa.tiles_x; bid /= wa.tiles_x;
        const int tyi = bid % wa.tiles_y;
        const int n = bid / wa.tiles_y;
        const int y0 = tyi * WIN_TH, x0 = txi * WIN_TW;
        const int wy0 = y0 - a.pad_h - wa.RW, wx0 = x0 - a.pad_w - wa.RW;
        const long img = (long)n * a.H * a.W;
        if (wb.dma_window) {
            for (int piece = wv; piece * 8 < npx; piece += 8) {
                const int px = piece * 8 + (lane >> 3);
                if (px < npx) {                                  // lanes past the window's last pixel stay out (exec mask)
                    const int ly = px / wa.WW, lx = px - ly * wa.WW;
                    const int gy = wy0 + ly, gx = wx0 + lx;
                    const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    dma16(rs_x, __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)(xw + piece * 256)),
                          ok ? (unsigned)(((img + (long)gy * a.W + gx) * a.C) * 4) + (unsigned)(lane & 7) * 16u : 0x80000000u, (unsigned)(c0 * 4));
                }
            }
        } else {
            for (int i = t; i < npx * 8; i += NT) {
                const int px = i >> 3, c4 = (i & 7) * 4;
                const int ly = px / wa.WW, lx = px - ly * wa.WW;
                const int gy = wy0 + ly, gx = wx0 + lx;
                const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                *reinterpret_cast<f32x4 *>(xw + (size_t)i * 4) =
                    *reinterpret_cast<const f32x4 *>(ok ? a.x + (img + (long)gy * a.W + gx) * a.C + c0 + c4 : a.zero);
            }
        }
        if constexpr (DYDMA)        // (the other variants read them in prepare_finish, which follows at once: no registers to park them)
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int it = t + i * NT;
            const int r = it / RS, tap = it - r * RS;
            const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
            g_oh[i] = 0.f; g_ow[i] = 0.f; g_mk[i] = 0.f;
            if (it < BM * RS && p < a.P && q < a.Q) {
                const long m = ((long)n * a.P + p) * a.Q + q;
                const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                g_oh[i] = po[0]; g_ow[i] = po[1];
                g_mk[i] = a.mask[m * (a.dg * RS) + g * RS + tap];
            }
        }
    };
    auto prepare_finish = [&](int tl) {
        int bid = tl;
        const int txi = bid % wa.tiles_x; bid /= wa.tiles_x;
        const int tyi = bid % wa.tiles_y;
        const int y0 = tyi * WIN_TH, x0 = txi * WIN_TW;
        const int wy0 = y0 - a.pad_h - wa.RW, wx0 = x0 - a.pad_w - wa.RW;
        const int n = bid / wa.tiles_y;
#pragma unroll(DYDMA ? GI : 1)
        for (int i = 0; i < GI; ++i) {
            const int it = t + i * NT;
            if (it >= BM * RS) continue;
            const int r = it / RS, tap = it - r * RS;
            const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
            int packed = 0;
            float flh = 0.f, flw = 0.f, mk = 0.f;
            if (p < a.P && q < a.Q) {
                const int ti = tap / a.S, tj = tap - ti * a.S;
                float oh, ow, omk;
                if constexpr (DYDMA) { oh = g_oh[i]; ow = g_ow[i]; omk = g_mk[i]; }
                else {
                    const long m = ((long)n * a.P + p) * a.Q + q;
                    const float *po = a.offset + m * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                    oh = po[0]; ow = po[1];
                    omk = a.mask[m * (a.dg * RS) + g * RS + tap];
                }
                const float h = (float)(p - a.pad_h + ti * a.dil) + oh;
                const float w = (float)(q - a.pad_w + tj * a.dil) + ow;
                if (h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W) {
                    const float hf = floorf(h), wf = floorf(w);
                    const int h0 = (int)hf, w0 = (int)wf;
                    int valid = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int hy = h0 + (e >> 1), wx = w0 + (e & 1);
                        if (hy >= 0 && hy <= a.H - 1 && wx >= 0 && wx <= a.W - 1) valid |= 1 << e;
                    }
                    const int ly = h0 - wy0, lx = w0 - wx0;
                    const bool fast = ly >= 0 && ly + 1 < wa.WH && lx >= 0 && lx + 1 < wa.WW;
                    packed = (valid << 16) | (fast ? (ly * wa.WW + lx) : GEO_SLOW);
                    flh = h - hf; flw = w - wf;
                    mk = omk;
                }
            }
            geo4[it] = f32x4{flh, flw, mk, __int_as_float(packed)};
        }
    };

    // DYDMA: slice kk (16 pixels = one row of the 8 x 16 block) of sub-block `sub` of block tl -> ring slot `slot` of this wave
    const i32x4d rs_dy = [&] {
        const unsigned long long u = reinterpret_cast<unsigned long long>(wb.dyb);
        return i32x4d{(int)__builtin_amdgcn_readfirstlane((unsigned)u), (int)(__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) & 0xffffu),
                      __builtin_amdgcn_readfirstlane((int)((long)a.M * a.K * 2)), 0x00020000};
    }();
    unsigned short *const ring = dyT + wv * (6 * 512);
    auto issue_slice = [&](int tl, int sub, int kk, int slot) {
        int bid = tl;
        const int txi = bid % wa.tiles_x; bid /= wa.tiles_x;
        const int tyi = bid % wa.tiles_y;
        const int n = bid / wa.tiles_y;
        const int p = tyi * WIN_TH + sub * (HP / WIN_TW) + kk, q = txi * WIN_TW + (lane >> 2);
        const unsigned off = (p < a.P && q < a.Q) ? (unsigned)(((((long)n * a.P + p) * a.Q + q) * a.K + (lane & 3) * 8) * 2) : 0x80000000u;
        dma16(rs_dy, __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)(ring + slot * 512)), off, (unsigned)((k0 + wv * 32) * 2));
    };
    int slot = 0;                                            // ring slot of the current sub-block's first slice

    int tile = split;
    if (tile < ntiles) {
        prepare_issue(tile); prepare_finish(tile); issue_dy(tile, 0);
        if constexpr (DYDMA) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) issue_slice(tile, 0, kk, kk);
        }
    }
    for (; tile < ntiles; tile += wb.splits) {
        int bid = tile;
        const int txi = bid % wa.tiles_x; bid /= wa.tiles_x;
        const int tyi = bid % wa.tiles_y;
        const int n = bid / wa.tiles_y;
        const int y0 = tyi * WIN_TH, x0 = txi * WIN_TW;
        const long img = (long)n * a.H * a.W;
        for (int half = 0; half < NSUB; ++half) {
            if (half == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's window pieces landed
            __syncthreads();                                 // window + geometry visible; dyT / colT free again
            commit_dy();
            // ---- masked bilinear samples of 64 pixels x 9 taps x 32 channels, bf16, pixel-minor
            const int r = half * HP + a_row;
#pragma unroll(DYDMA ? 3 : 1)
            for (int ti_ = 0; ti_ < NTAP; ++ti_) {
                const int tap = tap0 + ti_;
                if (tap >= RS) break;
                const f32x4 gq = geo4[r * RS + tap];
                const int gi = __float_as_int(gq[3]);
                const int valid = (gi >> 16) & 15;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (valid) {
                    const float flh = gq[0], flw = gq[1], mk = gq[2];
                    const float hh = 1.f - flh, hw = 1.f - flw;
                    const float wt[4] = {hh * hw * mk, hh * flw * mk, flh * hw * mk, flh * flw * mk};     // as make_tap: the forward's samples exactly
                    if (!(gi & GEO_SLOW)) {
                        // a corner outside the image reads a window pixel filled with zeros: no validity selects (adding
                        // weight x 0 leaves the forward's sum bit for bit)
                        const float *xb = xw + (size_t)(gi & 0xffff) * CW + a_col;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v += *reinterpret_cast<const f32x4 *>(xb + ((e >> 1) * wa.WW + (e & 1)) * CW) * wt[e];
                    } else {
                        const int p = y0 + r / WIN_TW, q = x0 + r % WIN_TW;
                        const float *po = a.offset + (((long)n * a.P + p) * a.Q + q) * (2 * a.dg * RS) + g * 2 * RS + 2 * tap;
                        const int ti = tap / a.S, tj = tap - ti * a.S;
                        const int h0 = (int)floorf((float)(p - a.pad_h + ti * a.dil) + po[0]);
                        const int w0 = (int)floorf((float)(q - a.pad_w + tj * a.dil) + po[1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if ((valid >> e) & 1)
                                v += *reinterpret_cast<const f32x4 *>(a.x + (img + (long)(h0 + (e >> 1)) * a.W + w0 + (e & 1)) * a.C + c0 + a_col) * wt[e];
                    }
                }
                if constexpr (F32) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) colF[(tap * CW + a_col + c) * LDP + a_row] = v[c];
                } else {
                    *reinterpret_cast<u16x4 *>(colT + (tap * HP + a_row) * CW + a_col) = f2bf4(v);   // [tap][pixel][channel]
                }
            }
            __syncthreads();
            // next half block's dY lands under the MFMAs
            if constexpr (DYDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this sub-block's dY slices (issued a phase ago)
            if (half + 1 < NSUB) issue_dy(tile, half + 1);
            else if (tile + wb.splits < ntiles) {            // the window is free: every sample of this block is built
                prepare_issue(tile + wb.splits);
                if constexpr (!DYDMA) prepare_finish(tile + wb.splits);       // (no registers to hold the offsets through the MFMA phase next to the dY staging)
                issue_dy(tile + wb.splits, 0);
            }
            // DYDMA: the sub-block after this one, whose slices go out while this one's are consumed
            const bool nx_ok = half + 1 < NSUB || tile + wb.splits < ntiles;
            const int nx_tile = half + 1 < NSUB ? tile : tile + wb.splits, nx_sub = half + 1 < NSUB ? half + 1 : 0;
            if constexpr (DYDMA) {
                if (nx_ok) {
                    issue_slice(nx_tile, nx_sub, 0, (slot + 4) % 6);
                    issue_slice(nx_tile, nx_sub, 1, (slot + 5) % 6);
                }
            }
#pragma unroll
            for (int kk = 0; kk < HP / 16; ++kk) {
                if constexpr (F32) {
                    // lane half lh_ takes pixels 8 lh_ .. +7 of the 16-pixel step (A and B agree on the pairing)
                    const f32x4 fa0 = *reinterpret_cast<const f32x4 *>(dyF + (wave * 32 + lr) * LDP + kk * 16 + 8 * lh_);
                    const f32x4 fa1 = *reinterpret_cast<const f32x4 *>(dyF + (wave * 32 + lr) * LDP + kk * 16 + 8 * lh_ + 4);
#pragma unroll
                    for (int tap = 0; tap < RS; ++tap) {
                        const f32x4 fb0 = *reinterpret_cast<const f32x4 *>(colF + (tap * CW + lr) * LDP + kk * 16 + 8 * lh_);
                        const f32x4 fb1 = *reinterpret_cast<const f32x4 *>(colF + (tap * CW + lr) * LDP + kk * 16 + 8 * lh_ + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb0[e], acc[tap], 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb1[e], acc[tap], 0, 0, 0);
                    }
                } else {
                    bf16x8 fa;
                    if constexpr (DYDMA) {
                        const int sl = (slot + kk) % 6;
                        fa = lds_tr_frag(ring + sl * 512, 0, lane);
                        if (kk < 2 && nx_ok) {               // its slot takes slice kk + 2 of the next sub-block once the read is through
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            issue_slice(nx_tile, nx_sub, kk + 2, sl);
                        }
                    } else {
                        fa = lds_tr_frag(dyT + wave * HP * 32, kk * 16, lane);
                    }
#pragma unroll
                    for (int tap = 0; tap < RS; ++tap) {
                        const bf16x8 fb = lds_tr_frag(colT + tap * HP * CW, kk * 16, lane);
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[tap], 0, 0, 0);
                    }
                }
            }
            if constexpr (DYDMA) slot = (slot + 4) % 6;
            // the next block's geometry from the offsets / masks that arrived under the MFMAs (the table is free since the last samples)
            if constexpr (DYDMA)
                if (half + 1 == NSUB && tile + wb.splits < ntiles) prepare_finish(tile + wb.splits);
        }
    }
    // ---- dW += this workgroup's partial sums: lane <-> channel, 128-byte row segments
#pragma unroll
    for (int tap = 0; tap < RS; ++tap)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int f = k0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh_;
            if (f < a.K && acc[tap][e] != 0.f) unsafeAtomicAdd(wb.dw + ((long)f * RS + tap) * a.C + c0 + lr, acc[tap][e]);
        }
}

__device__ float rr_dcn_zero16[4] = {0.f, 0.f, 0.f, 0.f};

int fill_args(DcnArgs &a, const float *x, const float *offset, const float *mask, const float *w, int n, int h, int wd,
              int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dil, int dg)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && k > 0 && r > 0 && s > 0 && stride > 0 && dil > 0 && dg > 0, "rr_dcn: bad dims");
    RR_CHECK_ARG(c % 4 == 0 && c % dg == 0, "rr_dcn: C=%d must be a multiple of 4 and of deformable_groups=%d", c, dg);
    RR_CHECK_ARG(dg == 1 || (c / dg) % BK == 0, "rr_dcn: channels per deformable group (%d) must be a multiple of 32", c / dg);
    a.x = x; a.offset = offset; a.mask = mask; a.w = w;
    a.N = n; a.H = h; a.W = wd; a.C = c; a.K = k; a.R = r; a.S = s;
    a.P = (h + 2 * pad_h - (dil * (r - 1) + 1)) / stride + 1;
    a.Q = (wd + 2 * pad_w - (dil * (s - 1) + 1)) / stride + 1;
    RR_CHECK_ARG(a.P > 0 && a.Q > 0, "rr_dcn: empty output");
    a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil = dil; a.dg = dg;
    const long M = (long)n * a.P * a.Q;
    RR_CHECK_ARG(M < (1l << 31), "rr_dcn: too many output pixels");
    a.M = (int)M;
    static float *zp[64] = {};
    int dev = 0;
    hipGetDevice(&dev);
    dev &= 63;
    if (!zp[dev]) hipGetSymbolAddress(reinterpret_cast<void **>(&zp[dev]), HIP_SYMBOL(rr_dcn_zero16));
    a.zero = zp[dev];
    return RR_OK;
}

}  // namespace

static int dcn_win_margin()
{
    static int v = -2;
    if (v == -2) {
        const char *e = getenv("RR_DCN_WINDOW");      // margin in pixels around the filter's reach; -1 / 0 disables the window kernel
        v = e ? atoi(e) : 3;
    }
    return v;
}

// LDS-window forward for either operand precision; -1 when the layer does not qualify.
// wpk[((tap * C/32 + cch) * K + ko) * 32 + cl] = bf16(w[(ko * RS + tap) * C + cch * 32 + cl]): one thread = 4 values
__global__ __launch_bounds__(256) void dcn_pack_weights_kernel(const float *w, unsigned short *wpk, int K, int C, int RS)
{
    const long total = (long)K * RS * C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % (C / 4));
        long rest = i / (C / 4);
        const int tap = (int)(rest % RS);
        const int ko = (int)(rest / RS);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w + ((long)ko * RS + tap) * C + c4 * 4);
        const int cch = (c4 * 4) / BK, cl = (c4 * 4) % BK;
        *reinterpret_cast<u16x4 *>(wpk + ((((long)tap * (C / BK) + cch) * K + ko) * BK + cl)) = f2bf4(v);
    }
}

static int dcn_fwd_win(const DcnArgs &a, int n, int k, int r, int s, int stride, int dilation, int c, int h, int wd, int bf16,
                       hipStream_t stream, const char *name, unsigned short *wpk = nullptr)
{
    const int rw = dcn_win_margin();
    if (!(rw > 0 && stride == 1 && k > 32 && c % BK == 0 && (long)h * wd < (1l << 30))) return -1;
    // the input block a pixel tile can reach is staged once per channel chunk
    DcnWinArgs wa{};
    wa.a = a;
    wa.tiles_y = rr_cdiv(a.P, WIN_TH);
    wa.tiles_x = rr_cdiv(a.Q, WIN_TW);
    const int wbn = (k % 256 == 0 || k > 384) ? 256 : 128;
    size_t lds = 0;
    bool fits = false;
    for (int m = rw; m >= 1 && !fits; --m) {          // the largest margin <= the requested one that fits
        wa.RW = m;
        wa.WH = WIN_TH + (r - 1) * dilation + 2 * m + 1;
        wa.WW = WIN_TW + (s - 1) * dilation + 2 * m + 1;
        const int npx = wa.WH * wa.WW;
        lds = (size_t)npx * BK * 4 + (size_t)BM * r * s * 4 * 8 + sizeof(unsigned short) * 2 * (BM * LDKH + wbn * LDKH);
        fits = npx <= 14 * 32 && lds <= 160 * 1024 - 512;
    }
    if (!fits) return -1;
    const int blocks = n * wa.tiles_y * wa.tiles_x * rr_cdiv(k, wbn);
#define RR_WIN_LAUNCH(BNV, F32V)                                                                                          \
    do {                                                                                                                  \
        hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_fprop_win_kernel<BNV, F32V>),                              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                        \
        hipLaunchKernelGGL((dcn_fprop_win_kernel<BNV, F32V>), dim3(blocks), dim3(512), lds, stream, wa);                  \
    } while (0)
    wa.wpk = nullptr;
    if (wbn == 256 && bf16 && wpk != nullptr) {      // B tile by LDS-DMA from the weights packed to bf16 once per call
        hipLaunchKernelGGL(dcn_pack_weights_kernel, dim3(rr_cdiv((long)k * r * s * c / 4, 256)), dim3(256), 0, stream, a.w, wpk, k, c,
                           r * s);
        wa.wpk = wpk;
    }
    if (wbn == 256) { if (bf16) RR_WIN_LAUNCH(256, false); else RR_WIN_LAUNCH(256, true); }
    else { if (bf16) RR_WIN_LAUNCH(128, false); else RR_WIN_LAUNCH(128, true); }
#undef RR_WIN_LAUNCH
    RR_CHECK_LAUNCH(name);
    return RR_OK;
}

extern "C" int rr_dcn_fwd(const float *x, const float *offset, const float *mask, const float *w, const float *bias,
                          float *y, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                          int dilation, int deformable_groups, hipStream_t stream)
{
    DcnArgs a{};
    const int rc = fill_args(a, x, offset, mask, w, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
    if (rc != RR_OK) return rc;
    a.bias = bias; a.y = y;
    const int wrc = dcn_fwd_win(a, n, k, r, s, stride, dilation, c, h, wd, 0, stream, "rr_dcn_fwd");
    if (wrc != -1) return wrc;
    const int bn = k > 32 ? 128 : 32;
    const int blocks = rr_cdiv(a.M, BM) * rr_cdiv(k, bn);
    const size_t lds = sizeof(float) * 2 * (BM * LDK + bn * LDK);
    if (bn == 128) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_fprop_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(dcn_fprop_kernel<128>, dim3(blocks), dim3(256), lds, stream, a);
    } else {
        hipLaunchKernelGGL(dcn_fprop_kernel<32>, dim3(blocks), dim3(256), lds, stream, a);
    }
    RR_CHECK_LAUNCH("rr_dcn_fwd");
    return RR_OK;
}


extern "C" int rr_dcn_fwd_bf16(const float *x, const float *offset, const float *mask, const float *w, const float *bias,
                               float *y, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                               int dilation, int deformable_groups, hipStream_t stream)
{
    DcnArgs a{};
    const int rc = fill_args(a, x, offset, mask, w, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
    if (rc != RR_OK) return rc;
    a.bias = bias; a.y = y;
    const int bn = k > 32 ? 128 : 32;
    const int wrc = dcn_fwd_win(a, n, k, r, s, stride, dilation, c, h, wd, 1, stream, "rr_dcn_fwd_bf16");
    if (wrc != -1) return wrc;
    const int blocks = rr_cdiv(a.M, BM) * rr_cdiv(k, bn);
    const size_t lds = sizeof(unsigned short) * 2 * (BM * LDKH + bn * LDKH);
    if (bn == 128) hipLaunchKernelGGL(dcn_fprop_bf16_kernel<128>, dim3(blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(dcn_fprop_bf16_kernel<32>, dim3(blocks), dim3(256), lds, stream, a);
    RR_CHECK_LAUNCH("rr_dcn_fwd_bf16");
    return RR_OK;
}

extern "C" size_t rr_dcn_wpack_bytes(int c, int k, int r, int s)
{
    return (size_t)k * r * s * c * sizeof(unsigned short);
}

extern "C" int rr_dcn_fwd_bf16_packed(const float *x, const float *offset, const float *mask, const float *w, const float *bias,
                                      float *y, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h,
                                      int pad_w, int dilation, int deformable_groups, void *wpack, hipStream_t stream)
{
    RR_CHECK_ARG(wpack != nullptr, "rr_dcn_fwd_bf16_packed: wpack (rr_dcn_wpack_bytes) is required");
    DcnArgs a{};
    const int rc = fill_args(a, x, offset, mask, w, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
    if (rc != RR_OK) return rc;
    a.bias = bias; a.y = y;
    const int wrc = dcn_fwd_win(a, n, k, r, s, stride, dilation, c, h, wd, 1, stream, "rr_dcn_fwd_bf16_packed",
                                (c % 32 == 0 && (long)k * r * s * c * 2 < (1l << 31)) ? static_cast<unsigned short *>(wpack) : nullptr);
    if (wrc != -1) return wrc;
    return rr_dcn_fwd_bf16(x, offset, mask, w, bias, y, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups, stream);
}

extern "C" size_t rr_dcn_col_bytes(int n, int h, int wd, int c, int r, int s, int stride, int pad_h, int pad_w, int dilation)
{
    const long p = (h + 2 * pad_h - (dilation * (r - 1) + 1)) / stride + 1;
    const long q = (wd + 2 * pad_w - (dilation * (s - 1) + 1)) / stride + 1;
    return (size_t)n * p * q * r * s * c * sizeof(float);
}

// columns for the weight gradient: col [M][R*S*C]
extern "C" int rr_dcn_im2col(const float *x, const float *offset, const float *mask, float *col, int n, int h, int wd, int c,
                             int r, int s, int stride, int pad_h, int pad_w, int dilation, int deformable_groups,
                             hipStream_t stream)
{
    DcnArgs a{};
    const int rc = fill_args(a, x, offset, mask, nullptr, n, h, wd, c, 1, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
    if (rc != RR_OK) return rc;
    const long total = (long)a.M * r * s * (c / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dcn_im2col_kernel, dim3((int)blocks), dim3(256), 0, stream, a, col);
    RR_CHECK_LAUNCH("rr_dcn_im2col");
    return RR_OK;
}

// dcol [M][R*S*C] -> dx (zeroed here, then scattered with float atomics), doffset, dmask
extern "C" int rr_dcn_col2im(const float *x, const float *offset, const float *mask, const float *dcol, float *dx,
                             float *doffset, float *dmask, int n, int h, int wd, int c, int r, int s, int stride,
                             int pad_h, int pad_w, int dilation, int deformable_groups, hipStream_t stream)
{
    DcnArgs a{};
    const int rc = fill_args(a, x, offset, mask, nullptr, n, h, wd, c, 1, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
    if (rc != RR_OK) return rc;
    hipMemsetAsync(dx, 0, sizeof(float) * (size_t)n * h * wd * c, stream);
    const long items = (long)a.M * r * s * deformable_groups;
    long blocks = (items + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dcn_col2im_kernel, dim3((int)blocks), dim3(256), 0, stream, a, dcol, dx, doffset, dmask);
    RR_CHECK_LAUNCH("rr_dcn_col2im");
    return RR_OK;
}

// Fused backward, part 1: weight gradient.  dw [K][R][S][C] += dY^T x deformed columns (float atomics; pre-zeroed or
// holding the running gradient).  Requires K % 4 == 0 and one deformable group per 128-channel tile.
static int dcn_wgrad_plain(const float *x, const float *offset, const float *mask, const float *dy, float *dw, int n, int h,
                           int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                           int deformable_groups, hipStream_t stream)
{
    DcnBwdArgs b{};
    const int rc = fill_args(b.a, x, offset, mask, nullptr, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
    if (rc != RR_OK) return rc;
    RR_CHECK_ARG(k % 4 == 0, "rr_dcn_wgrad: K=%d must be a multiple of 4", k);
    RR_CHECK_ARG(deformable_groups == 1 || (c / deformable_groups) % 128 == 0,
                 "rr_dcn_wgrad: channels per deformable group (%d) must be a multiple of 128", c / deformable_groups);
    b.dy = dy; b.dw = dw;
    b.mt = rr_cdiv(k, 128); b.nt = rr_cdiv(c, 128);
    const int tiles = b.mt * b.nt * r * s;
    const int total_chunks = rr_cdiv(b.a.M, BK);
    int splits = tiles < 512 ? 512 / tiles : 1;
    if (splits > rr_cdiv(total_chunks, 8)) splits = rr_cdiv(total_chunks, 8);
    if (splits < 1) splits = 1;
    b.chunks_per_split = rr_cdiv(total_chunks, splits);
    splits = rr_cdiv(total_chunks, b.chunks_per_split);
    const size_t lds = sizeof(float) * 2 * (BK * 128 + BK * 128);
    hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dcn_wgrad_kernel, dim3(tiles * splits), dim3(256), lds, stream, b);
    RR_CHECK_LAUNCH("rr_dcn_wgrad");
    return RR_OK;
}

// Window kernel for either operand precision; returns -1 when the layer does not qualify (caller takes dcn_wgrad_kernel).
// LDS footprint of the window kernels for a margin of m pixels (0: the window does not fit with any margin >= 1)
static size_t dcn_wgrad_win_lds(int r, int s, int dilation, int m)
{
    const int npx = (WIN_TH + (r - 1) * dilation + 2 * m + 1) * (WIN_TW + (s - 1) * dilation + 2 * m + 1);
    return sizeof(float) * (size_t)(npx * 32 + BM * r * s * 4) + sizeof(unsigned short) * (size_t)((256 + r * s * 32) * 72);
}

static size_t dcn_dgrad_win_lds(int r, int s, int dilation, int m)
{
    const int npx = (WIN_TH + (r - 1) * dilation + 2 * m + 1) * (WIN_TW + (s - 1) * dilation + 2 * m + 1);
    // operand area: one image of the sweep (dY 128 x LDKH + weights taps x 32 x LDKH), and for the DMA sweep image 0 + the
    // third dY image (128 x 32 + taps x 32 x 32 + 128 x 32) — the larger of the two
    const size_t opa = std::max((size_t)(BM * LDKH + r * s * 32 * LDKH), (size_t)(2 * BM * 32 + r * s * 32 * 32));
    return sizeof(float) * (size_t)(((npx * 33 + 3) & ~3) + BM * r * s * 7 + npx * 32) + sizeof(unsigned short) * opa;
}

static bool dcn_win_fits(bool dgrad, int r, int s, int dilation)
{
    return (dgrad ? dcn_dgrad_win_lds(r, s, dilation, 1) : dcn_wgrad_win_lds(r, s, dilation, 1)) <= 160 * 1024 - 512;
}

static int dcn_wgrad_win(const float *x, const float *offset, const float *mask, const float *dy, float *dw, int n, int h,
                         int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                         int deformable_groups, int bf16, hipStream_t stream, const unsigned short *dyb = nullptr)
{
    const int rw = dcn_win_margin();
    if (rw > 0 && stride == 1 && r * s == 9 && c % 32 == 0 && k % 4 == 0 && h < 32768 && wd < 32768 &&
        (deformable_groups == 1 || (c / deformable_groups) % 32 == 0)) {
        DcnWinWgradArgs wb{};
        const int rc = fill_args(wb.w.a, x, offset, mask, nullptr, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
        if (rc != RR_OK) return rc;
        wb.w.tiles_y = rr_cdiv(wb.w.a.P, WIN_TH);
        wb.w.tiles_x = rr_cdiv(wb.w.a.Q, WIN_TW);
        wb.dy = dy; wb.dw = dw;
        size_t ldsw = 0;
        constexpr bool dydma_on = true;
        const bool dydma = dydma_on && bf16 && dyb != nullptr && k % 32 == 0 && (long)wb.w.a.M * k * 2 < (1l << 31) &&
                           (long)n * h * wd * c * 4 < (1l << 31);
        for (int m = rw; m >= 1; --m) {
            wb.w.RW = m;
            wb.w.WH = WIN_TH + (r - 1) * dilation + 2 * m + 1;
            wb.w.WW = WIN_TW + (s - 1) * dilation + 2 * m + 1;
            ldsw = dcn_wgrad_win_lds(r, s, dilation, m);
            // the dY ring: 8 waves x 6 KB instead of the 256 x 72 image; the samples unpadded
            if (dydma) ldsw = ldsw - sizeof(unsigned short) * (size_t)((256 + r * s * 32) * 72) + sizeof(unsigned short) * (size_t)(8 * 6 * 512 + r * s * 64 * 32);
            if (ldsw <= 160 * 1024 - 512) break;
        }
        if (ldsw <= 160 * 1024 - 512) {
            const int ntiles = n * wb.w.tiles_y * wb.w.tiles_x;
            wb.ktiles = rr_cdiv(k, 256);
            const int per_split = (c / 32) * wb.ktiles;
            int splits = rr_cdiv(256, per_split);          // one workgroup per CU (the window leaves room for one)
            if (splits > 8) splits = splits / 8 * 8;
            if (splits > ntiles) splits = ntiles;
            wb.splits = splits;
            wb.dma_window = (long)n * h * wd * c * 4 < (1l << 31);
            wb.dyb = dyb;
            if (dydma) {
                RR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_wgrad_win_kernel<9, false, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw), "rr_dcn_wgrad");
                hipLaunchKernelGGL((dcn_wgrad_win_kernel<9, false, true>), dim3(splits * per_split), dim3(512), ldsw, stream, wb);
            } else if (bf16) {
                hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_wgrad_win_kernel<9, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                hipLaunchKernelGGL((dcn_wgrad_win_kernel<9, false>), dim3(splits * per_split), dim3(512), ldsw, stream, wb);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_wgrad_win_kernel<9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                hipLaunchKernelGGL((dcn_wgrad_win_kernel<9, true>), dim3(splits * per_split), dim3(512), ldsw, stream, wb);
            }
            RR_CHECK_LAUNCH("rr_dcn_wgrad");
            return RR_OK;
        }
    }
    return -1;
}

static int dcn_wgrad_plain(const float *x, const float *offset, const float *mask, const float *dy, float *dw, int n, int h,
                           int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                           int deformable_groups, hipStream_t stream);

// 1 when rr_dcn_wgrad / rr_dcn_dgrad take this layer (the host layer runs the column path otherwise): K % 4 == 0 and
// the deformable groups either span whole 128-channel tiles (L2-gather kernels) or the window kernels apply (3x3,
// stride 1, C % 32 == 0, groups of a multiple of 32 channels).
extern "C" int rr_dcn_fused_bwd_supported_dil(int c, int k, int r, int s, int stride, int dilation, int deformable_groups)
{
    if (c <= 0 || k <= 0 || deformable_groups <= 0 || c % deformable_groups != 0 || k % 4 != 0 || c % 4 != 0 || dilation <= 0)
        return 0;
    const int cpg = c / deformable_groups;
    if (deformable_groups == 1 || cpg % 128 == 0) return 1;     // the L2-gather kernels take these whatever the window does
    // smaller groups exist on the window kernels only: BOTH of them must fit their LDS windows with a margin >= 1
    // (a 3x3 filter with dilation >= 3 does not: rr_dcn_dgrad would drop to the L2-gather kernel and refuse the groups)
    return dcn_win_margin() > 0 && stride == 1 && r * s == 9 && c % 32 == 0 && cpg % 32 == 0 &&
           dcn_win_fits(false, r, s, dilation) && dcn_win_fits(true, r, s, dilation);
}

extern "C" int rr_dcn_fused_bwd_supported(int c, int k, int r, int s, int stride, int deformable_groups)
{
    return rr_dcn_fused_bwd_supported_dil(c, k, r, s, stride, 1, deformable_groups);
}

extern "C" int rr_dcn_wgrad_bf16(const float *x, const float *offset, const float *mask, const float *dy, float *dw, int n, int h,
                                 int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                                 int deformable_groups, hipStream_t stream)
{
    const int rc = dcn_wgrad_win(x, offset, mask, dy, dw, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups, 1, stream);
    if (rc != -1) return rc;
    return dcn_wgrad_plain(x, offset, mask, dy, dw, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups, stream);
}

// rr_dcn_wgrad_bf16 with dY's bf16 image (its producer's, or one rr_to_bf16 pass shared with the data gradient): the dY
// operand goes global -> LDS by DMA
extern "C" int rr_dcn_wgrad_bf16_img(const float *x, const float *offset, const float *mask, const float *dy,
                                     const unsigned short *dy_bf16, float *dw, int n, int h, int wd, int c, int k, int r, int s,
                                     int stride, int pad_h, int pad_w, int dilation, int deformable_groups, hipStream_t stream)
{
    RR_CHECK_ARG(dy_bf16 != nullptr, "rr_dcn_wgrad_bf16_img: the bf16 image of dy is required");
    const int rc = dcn_wgrad_win(x, offset, mask, dy, dw, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups, 1, stream, dy_bf16);
    if (rc != -1) return rc;
    return dcn_wgrad_plain(x, offset, mask, dy, dw, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups, stream);
}

extern "C" int rr_dcn_wgrad(const float *x, const float *offset, const float *mask, const float *dy, float *dw, int n, int h,
                            int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                            int deformable_groups, hipStream_t stream)
{
    const int rc = dcn_wgrad_win(x, offset, mask, dy, dw, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups, 0, stream);
    if (rc != -1) return rc;
    return dcn_wgrad_plain(x, offset, mask, dy, dw, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups, stream);
}

// Fused backward, part 2: dx (zeroed here, then float atomics on the bilinear corners), doffset, dmask (plain stores).
// dy (fp32) -> bf16, round to nearest even: one pass per data-gradient call (see DcnWinBwdArgs::dyb)
__global__ __launch_bounds__(256) void dcn_to_bf16_kernel(const f32x4 *src, u16x4 *dst, long n4)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) dst[i] = f2bf4(src[i]);
}

static int dcn_dgrad_impl(const float *x, const float *offset, const float *mask, const float *w, const float *dy,
                          float *dx, float *doffset, float *dmask, int n, int h, int wd, int c, int k, int r, int s,
                          int stride, int pad_h, int pad_w, int dilation, int deformable_groups, int bf16, hipStream_t stream,
                          unsigned short *dyb = nullptr, bool dyb_ready = false, unsigned short *wpk = nullptr, bool accumulate = false)
{
    DcnBwdArgs b{};
    const int rc = fill_args(b.a, x, offset, mask, w, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
    if (rc != RR_OK) return rc;
    RR_CHECK_ARG(k % 4 == 0, "rr_dcn_dgrad: K=%d must be a multiple of 4", k);
    RR_CHECK_ARG((long)n * h * wd < (1l << 31), "rr_dcn_dgrad: input too large");
    b.dy = dy; b.dx = dx; b.doffset = doffset; b.dmask = dmask;
    if (!accumulate) hipMemsetAsync(dx, 0, sizeof(float) * (size_t)n * h * wd * c, stream);     // (every kernel below ADDS into dx with float atomics)
    const int rw = dcn_win_margin();
    if (rw > 0 && stride == 1 && c % 32 == 0 && h < 32768 && wd < 32768 &&
        (deformable_groups == 1 || (c / deformable_groups) % 32 == 0)) {
        DcnWinBwdArgs wb{};
        wb.w.a = b.a;
        wb.w.tiles_y = rr_cdiv(b.a.P, WIN_TH);
        wb.w.tiles_x = rr_cdiv(b.a.Q, WIN_TW);
        wb.dy = dy; wb.dx = dx; wb.doffset = doffset; wb.dmask = dmask;
        wb.dyb = nullptr;
        // two windows live in LDS here (d input in fixed point, input values): the margin is the largest <= the
        // requested one that fits (2 pixels for a 3x3 filter with dilation 1)
        size_t ldsw = 0;
        for (int m = rw; m >= 1; --m) {
            wb.w.RW = m;
            wb.w.WH = WIN_TH + (r - 1) * dilation + 2 * m + 1;
            wb.w.WW = WIN_TW + (s - 1) * dilation + 2 * m + 1;
            ldsw = dcn_dgrad_win_lds(r, s, dilation, m);
            if (ldsw <= 160 * 1024 - 512) break;
        }
        if (r * s == 9 && ldsw <= 160 * 1024 - 512) {
            const dim3 grid(n * wb.w.tiles_y * wb.w.tiles_x);
            if (bf16) {
                if (dyb != nullptr && dyb_ready) {
                    wb.dyb = dyb;                 // the caller's bf16 image of dY (written by dY's producer): nothing to convert
                } else if (dyb != nullptr && k % 4 == 0) {
                    const long n4 = (long)b.a.M * k / 4;
                    long cb = (n4 + 255) / 256;
                    if (cb > 256 * 32) cb = 256 * 32;
                    hipLaunchKernelGGL(dcn_to_bf16_kernel, dim3((int)cb), dim3(256), 0, stream, reinterpret_cast<const f32x4 *>(dy),
                                       reinterpret_cast<u16x4 *>(dyb), n4);
                    wb.dyb = dyb;
                }
                constexpr bool dma_on = true;
                if (dma_on && wb.dyb != nullptr && wpk != nullptr && k % 32 == 0 && (long)b.a.M * k * 2 < (1l << 31) &&
                    (long)n * h * wd * c * 4 < (1l << 31) && wb.w.WH * wb.w.WW <= 368 &&
                    ldsw + (size_t)wb.w.WH * wb.w.WW * 4 + 16 <= 160 * 1024 - 512 &&
                    (size_t)wb.w.WH * wb.w.WW * 33 * 4 >= (size_t)(BM * 32 + 2 * r * s * 32 * 32) * 2) {     // the ring's share of the d-input window
                    // both sweep operands by LDS-DMA: the weights packed to bf16 (the forward's layout) once per call
                    hipLaunchKernelGGL(dcn_pack_weights_kernel, dim3(rr_cdiv((long)k * r * s * c / 4, 256)), dim3(256), 0, stream, w, wpk, k, c,
                                       r * s);
                    wb.wpk = wpk;
                    wb.goff_at = (int)ldsw;
                    const size_t ldsd = ldsw + (((size_t)wb.w.WH * wb.w.WW * 4 + 15) & ~(size_t)15);
                    RR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_dgrad_win_kernel<9, false, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd), "rr_dcn_dgrad");
                    hipLaunchKernelGGL((dcn_dgrad_win_kernel<9, false, true>), grid, dim3(512), ldsd, stream, wb);
                    RR_CHECK_LAUNCH("rr_dcn_dgrad");
                    return RR_OK;
                }
                hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_dgrad_win_kernel<9, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                hipLaunchKernelGGL((dcn_dgrad_win_kernel<9, false>), grid, dim3(512), ldsw, stream, wb);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_dgrad_win_kernel<9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                hipLaunchKernelGGL((dcn_dgrad_win_kernel<9, true>), grid, dim3(512), ldsw, stream, wb);
            }
            RR_CHECK_LAUNCH("rr_dcn_dgrad");
            return RR_OK;
        }
    }
    RR_CHECK_ARG(deformable_groups == 1 || (c / deformable_groups) % 128 == 0,
                 "rr_dcn_dgrad: channels per deformable group (%d) must be a multiple of 128 on the L2-gather kernel", c / deformable_groups);
    const size_t lds = sizeof(float) * (2 * (BM * LDK + BK * 128) + BM * 8 + BM * 4 + BM * 3);
    hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dcn_dgrad_kernel, dim3(rr_cdiv(b.a.M, BM)), dim3(256), lds, stream, b);
    RR_CHECK_LAUNCH("rr_dcn_dgrad");
    return RR_OK;
}

extern "C" int rr_dcn_dgrad(const float *x, const float *offset, const float *mask, const float *w, const float *dy,
                            float *dx, float *doffset, float *dmask, int n, int h, int wd, int c, int k, int r, int s,
                            int stride, int pad_h, int pad_w, int dilation, int deformable_groups, hipStream_t stream)
{
    return dcn_dgrad_impl(x, offset, mask, w, dy, dx, doffset, dmask, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation,
                          deformable_groups, 0, stream);
}

// The same gradients with bf16 matrix operands (dY and W rounded to bf16 for the column-gradient GEMM, fp32 accumulation,
// fp32 scatter) and d input pre-summed on chip: the backward of rr_dcn_fwd_bf16 (BASELINE config 4).
extern "C" int rr_dcn_dgrad_bf16(const float *x, const float *offset, const float *mask, const float *w, const float *dy,
                                 float *dx, float *doffset, float *dmask, int n, int h, int wd, int c, int k, int r, int s,
                                 int stride, int pad_h, int pad_w, int dilation, int deformable_groups, hipStream_t stream)
{
    return dcn_dgrad_impl(x, offset, mask, w, dy, dx, doffset, dmask, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation,
                          deformable_groups, 1, stream);
}

extern "C" size_t rr_dcn_dyb_bytes(int n, int p, int q, int k)
{
    return (size_t)n * p * q * k * sizeof(unsigned short);
}

extern "C" int rr_dcn_dgrad_bf16_ws(const float *x, const float *offset, const float *mask, const float *w, const float *dy,
                                    float *dx, float *doffset, float *dmask, int n, int h, int wd, int c, int k, int r, int s,
                                    int stride, int pad_h, int pad_w, int dilation, int deformable_groups, void *dyb,
                                    hipStream_t stream)
{
    return dcn_dgrad_impl(x, offset, mask, w, dy, dx, doffset, dmask, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation,
                          deformable_groups, 1, stream, static_cast<unsigned short *>(dyb));
}

// dy_bf16: dY's bf16 image as its producer left it (same element order as dy; csrc/conv16.hip's contract) — no conversion pass
extern "C" int rr_dcn_dgrad_bf16_img(const float *x, const float *offset, const float *mask, const float *w, const float *dy,
                                     const unsigned short *dy_bf16, float *dx, float *doffset, float *dmask, int n, int h, int wd,
                                     int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                                     int deformable_groups, hipStream_t stream)
{
    RR_CHECK_ARG(dy_bf16 != nullptr, "rr_dcn_dgrad_bf16_img: the bf16 image of dy is required");
    return dcn_dgrad_impl(x, offset, mask, w, dy, dx, doffset, dmask, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation,
                          deformable_groups, 1, stream, const_cast<unsigned short *>(dy_bf16), true);
}

// Workspace of rr_dcn_dgrad_bf16_packed: the weights packed to bf16, then (when the caller has no bf16 image of dY) dY in bf16
extern "C" size_t rr_dcn_dgrad_ws_bytes(int n, int p, int q, int c, int k, int r, int s, int have_dy_bf16)
{
    const size_t wb = ((size_t)c * k * r * s * sizeof(unsigned short) + 255) & ~(size_t)255;
    return wb + (have_dy_bf16 ? 0 : (size_t)n * p * q * k * sizeof(unsigned short));
}

extern "C" int rr_dcn_dgrad_bf16_packed(const float *x, const float *offset, const float *mask, const float *w, const float *dy,
                                        const unsigned short *dy_bf16, float *dx, float *doffset, float *dmask, int n, int h,
                                        int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                                        int deformable_groups, int accumulate, void *ws, hipStream_t stream)
{
    RR_CHECK_ARG(ws != nullptr, "rr_dcn_dgrad_bf16_packed: workspace required (rr_dcn_dgrad_ws_bytes)");
    unsigned short *wpk = static_cast<unsigned short *>(ws);
    const size_t wbytes = ((size_t)c * k * r * s * sizeof(unsigned short) + 255) & ~(size_t)255;
    unsigned short *dyb = dy_bf16 ? const_cast<unsigned short *>(dy_bf16)
                                  : reinterpret_cast<unsigned short *>(static_cast<unsigned char *>(ws) + wbytes);
    return dcn_dgrad_impl(x, offset, mask, w, dy, dx, doffset, dmask, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation,
                          deformable_groups, 1, stream, dyb, dy_bf16 != nullptr, wpk, accumulate != 0);
}

extern "C" int rr_dcn_split_fwd(const float *om, long m, int third, float *offset, float *mask, hipStream_t stream)
{
    RR_CHECK_ARG(m >= 0 && third > 0, "rr_dcn_split_fwd: bad dims");
    if (m == 0) return RR_OK;
    long blocks = (m * 3 * third + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dcn_split_fwd_kernel, dim3((int)blocks), dim3(256), 0, stream, om, m, third, offset, mask);
    RR_CHECK_LAUNCH("rr_dcn_split_fwd");
    return RR_OK;
}

extern "C" int rr_dcn_split_bwd(const float *doffset, const float *dmask, const float *mask, long m, int third, float *dom,
                                hipStream_t stream)
{
    RR_CHECK_ARG(m >= 0 && third > 0, "rr_dcn_split_bwd: bad dims");
    if (m == 0) return RR_OK;
    long blocks = (m * 3 * third + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dcn_split_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, doffset, dmask, mask, m, third, dom);
    RR_CHECK_LAUNCH("rr_dcn_split_bwd");
    return RR_OK;
}
