// Modulated deformable convolution (DCNv2) for gfx950 — BASELINE.json config 4, SURVEY §8 row a22.
//
// Replaces ext/dcn of the reference: `dcn_v2_cuda_forward/backward` (src/cuda/dcn_v2_cuda.cu:42-172,
// 206-335) and kernels K2-K5 (src/cuda/dcn_v2_im2col_cuda.cu:125-327), bound by ext/dcn/dcn_v2.py:16-52.
//   out[n,o,p] = b_o + sum_{c,i,j} W[o,c,i,j] * m[n,g,ij,p] * bilinear(x[n,c], p*s - pad + ij*dil + d[n,g,ij,p])
// Forward = ONE gather-GEMM: the reference's im2col kernel + column buffer (9x the input) + batched
// cuBLAS GEMM become the A-operand load of an implicit GEMM on v_mfma_f32_32x32x2_f32 — per (pixel, tap)
// four NHWC corner rows are fetched as float4s along C and blended with the bilinear x mask weights on
// their way into LDS.  MFMA-bound like a 3x3 convolution (2*M*K*R*S*C FLOPs) with 4x its activation reads.
// Backward (this round): columns / column gradients are materialised like the reference does
// ([M, R*S*C] each), the two GEMMs run on the conv kernels (rr_conv_dgrad / rr_conv_wgrad as 1x1 layers),
// and one wave per (pixel, tap) turns the column gradient into d input (float atomics), d offset, d mask.
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
constexpr int LDKH = BK + 8;

__device__ __forceinline__ unsigned short f2bf(float f)
{
    unsigned int u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);          // round to nearest even (inputs are finite)
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ u16x4 f2bf4(f32x4 v)
{
    u16x4 r;
    r[0] = f2bf(v[0]); r[1] = f2bf(v[1]); r[2] = f2bf(v[2]); r[3] = f2bf(v[3]);
    return r;
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

extern "C" int rr_dcn_fwd(const float *x, const float *offset, const float *mask, const float *w, const float *bias,
                          float *y, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                          int dilation, int deformable_groups, hipStream_t stream)
{
    DcnArgs a{};
    const int rc = fill_args(a, x, offset, mask, w, n, h, wd, c, k, r, s, stride, pad_h, pad_w, dilation, deformable_groups);
    if (rc != RR_OK) return rc;
    a.bias = bias; a.y = y;
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
    const int blocks = rr_cdiv(a.M, BM) * rr_cdiv(k, bn);
    const size_t lds = sizeof(unsigned short) * 2 * (BM * LDKH + bn * LDKH);
    if (bn == 128) hipLaunchKernelGGL(dcn_fprop_bf16_kernel<128>, dim3(blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(dcn_fprop_bf16_kernel<32>, dim3(blocks), dim3(256), lds, stream, a);
    RR_CHECK_LAUNCH("rr_dcn_fwd_bf16");
    return RR_OK;
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
