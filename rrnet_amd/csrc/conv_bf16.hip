// bf16-operand NHWC convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation) — BASELINE
// config 4 ("RRNet ... bf16"), selected by cfg.Model.bf16; the headline configuration stays on csrc/conv.hip (fp32).
//
// The reference is fp32-only (/root/reference/backbones/hourglass.py:12-61,127-199 through nn.Conv2d / cuDNN), so this
// precision is builder-defined: activations, weights, gradients and every BatchNorm / loss / optimizer quantity stay
// fp32 in HBM; the two operands of a convolution are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on
// their way into LDS and multiplied with fp32 accumulation.  Parity contract (tests/test_conv_bf16_gpu.py): every kernel
// equals the fp32 kernel of csrc/conv.hip run on bf16-rounded operands within 1e-5 of the output scale (products of two
// bf16 values are exact in fp32: only the summation order differs).
//
// Same implicit GEMMs as csrc/conv.hip, no column matrix:
//   fprop : Y[m,ko]  = sum_{tap,c} X[pix(m,tap), c] * W[ko,tap,c]      M = N*P*Q, N = K, Kg = R*S*C
//   dgrad : stride 1 = fprop of dY with the flipped / transposed filter (rr_weight_flip_transpose), as in conv.hip
//   wgrad : dW[ko,tap,c] += sum_m dY[m,ko] * X[pix(m,tap), c]           M = K, N = C, Kg = N*P*Q (split)
// One K-step = 32 reduction indices = 2 MFMAs per 32x32 tile (16x fewer matrix instructions than fp32 for the same
// FLOPs): the kernels are bound by operand delivery (global -> registers -> LDS), not by the matrix pipe.
//   fprop: [m][k] / [n][k] LDS images of 40 bf16 per row (80-byte rows: one conflict-free ds_read_b128 = the 8 k values a
//          lane feeds to one MFMA), double buffered, one barrier per K-step, the next tile's global loads in flight
//          across it; 3 workgroups per CU cover each other's waits.
//   wgrad: both operands arrive k-major ([pixel][channel]); they are stored row-major in 32-column LDS blocks and read
//          through the hardware transpose ds_read_b64_tr_b16.
// Roofline: MFMA bf16 (2.5 PFLOP/s dense) for the FLOPs, HBM for the fp32 operands: at 256 -> 256 3x3 on 8 x 256 x 256
// the layer moves 1.07 GB for 618.5 GFLOP (ridge at 8 TB/s: 0.13 ms; MFMA at peak: 0.25 ms).
#include "common.h"
#include "rrnet_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

int rr_conv_pick_ksplit(int blocks, int nk);      // csrc/conv.hip: the occupancy model shared with the fp32 kernels
int rr_conv_small_tiles();
int rr_conv_mid_tiles();

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = BK + 8;          // [row][k] image row stride in bf16 (80 bytes): conflict-free ds_read_b128

__device__ __forceinline__ u16x4 f2bf4(f32x4 v) { return __builtin_bit_cast(u16x4, __builtin_convertvector(v, bf16x4)); }

__device__ __forceinline__ int xcd_remap(int bid, int nb)
{
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *p, long bytes)
{   // descriptor inputs through readfirstlane: provably wave-uniform, no waterfall loop (cdna_hip_programming.md T20)
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

struct ConvArgs {
    const float *src;  // X [N,H,W,C]  (or dY for the stride-1 data gradient)
    const float *w;    // [K][R][S][C] fp32
    const unsigned short *w16;   // the same filter already rounded to bf16 (B16 instantiation: read 8 channels per load, no converts)
    float *dst;        // Y [N,P,Q,K]
    const float *bias;
    double *stat_slab; // [mtiles][2][K] per-block column sums / sums of squares, or null
    // BatchNorm-backward sums / masked store of the producer (see csrc/conv.hip ConvArgs): same semantics
    const float *bs_y, *bs_z, *bs_mean, *bs_invstd, *bs_msc, *bs_msh;
    int bs_relu_bias;
    int N, SH, SW, SC, DH, DW, DC;
    int R, S, stride, pad_h, pad_w;
    int relu, accumulate;
    int M, wK, wC;
    int ksplit;
    // strided destination (stride-2 data gradient, one output parity class per launch): logical output pixel (n, h, w) of
    // the DH x DW grid lands at physical pixel (n, h * osh + oh0, w * osw + ow0) of an OH x OW map; osh == 0: dense
    int OH, OW, osh, osw, oh0, ow0;
};

// 128 x BN output tile, 256 threads = 4 waves (BN 128: 2x2 waves of 64x64; BN 64: 4x1 waves of 32x64; BN 32: 4x1 of 32x32)
// B16: the filter comes as bf16 (a.w16; C % 8 == 0): half the B loads, no converts, 16-byte LDS stores for the B image.
// SO: strided destination (ConvArgs::osh; the stride-2 data gradient's parity-class launches) — its own instantiation: folded
// into the plain kernel the per-element pixel arithmetic cost 20 registers and the third workgroup per CU (585 -> 524 TFLOP/s).
template <int BN, bool BNS, bool B16 = false, bool SO = false>
__global__ __launch_bounds__(256) void conv_igemm_bf16_kernel(const ConvArgs a)
{
    constexpr int WN = BN / 64 ? BN / 64 : 1;
    constexpr int WM = 4 / WN;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int A_ELEMS = BM * LDK, B_ELEMS = BN * LDK;     // bf16 elements
    constexpr int CPR = BK / 4;                 // float4 columns per row (8)
    constexpr int RPP = 256 / CPR;              // rows per pass of the 256 threads (32)
    constexpr int AJ = BM / RPP;                // 4
    constexpr int BJ = B16 ? (BN / 64 > 0 ? BN / 64 : 1) : (BN / RPP > 0 ? BN / RPP : 1);   // B16: 4 chunks of 8 channels per row, 64 rows per pass

    extern __shared__ __align__(16) unsigned short lds16[];
    unsigned short *As = lds16;                 // [2][A_ELEMS]
    unsigned short *Bs = lds16 + 2 * A_ELEMS;   // [2][B_ELEMS]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = (a.DC + BN - 1) / BN;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tile = logical % ntiles, m_tile = logical / ntiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int RS = a.R * a.S;
    const int cpt = (a.SC + BK - 1) / BK;
    const int nk_all = cpt * RS;
    int kc_lo = 0, kc_hi = nk_all;
    if (a.ksplit > 1) {
        const int per = (nk_all + a.ksplit - 1) / a.ksplit;
        kc_lo = blockIdx.z * per;
        kc_hi = kc_lo + per < nk_all ? kc_lo + per : nk_all;
        if (kc_lo >= kc_hi) return;
    }

    // ---- per-thread rows: element offset of the source pixel under tap (0,0) and a bit mask of the taps inside the image
    const int a_col = (t % CPR) * 4, a_row = t / CPR;
    int a_boff[AJ];
    unsigned long long a_mask[AJ];
    const int hw = a.DH * a.DW;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int m = m0 + a_row + RPP * j;
        int n = 0, h = 0, w = 0;
        const bool live = m < a.M;
        if (live) {
            n = m / hw;
            const int rem = m - n * hw;
            h = rem / a.DW;
            w = rem - h * a.DW;
        }
        const int ih0 = h * a.stride - a.pad_h, iw0 = w * a.stride - a.pad_w;
        unsigned long long mk = 0ull;
        if (live) {
            for (int ri = 0; ri < a.R; ++ri) {
                const int ih = ih0 + ri;
                if (ih < 0 || ih >= a.SH) continue;
                for (int si = 0; si < a.S; ++si) {
                    const int iw = iw0 + si;
                    if (iw >= 0 && iw < a.SW) mk |= 1ull << (ri * a.S + si);
                }
            }
        }
        a_mask[j] = mk;
        a_boff[j] = (int)(((((long)n * a.SH + ih0) * a.SW + iw0) * a.SC + a_col) * 4);
    }
    int b_boff[BJ];
    bool b_ok[BJ];
    const int b_row = B16 ? t / 4 : a_row, b_col = B16 ? (t % 4) * 8 : a_col, b_rpp = B16 ? 64 : RPP;
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int ko = n0 + b_row + b_rpp * j;
        b_ok[j] = ko < a.wK && (b_row + b_rpp * j) < BN;
        b_boff[j] = (int)(((long)ko * RS * a.wC + b_col) * (B16 ? 2 : 4));
    }
    const __amdgpu_buffer_rsrc_t rs_src = make_srd(a.src, (long)a.N * a.SH * a.SW * a.SC * 4);
    const __amdgpu_buffer_rsrc_t rs_w = B16 ? make_srd(a.w16, (long)a.wK * RS * a.wC * 2) : make_srd(a.w, (long)a.wK * RS * a.wC * 4);
    constexpr unsigned OOB = 0xFFFFFFF0u;       // beyond any (< 2 GiB) tensor: the hardware returns 0, no select on the data

    f32x4 ra[AJ], rb[BJ];
    // wave-uniform state of the K-step being fetched (tap inner, channel chunk outer: the taps re-read the same lines from L2)
    int p_cch = kc_lo / RS, p_tl = kc_lo - (kc_lo / RS) * RS;
    int p_ri = p_tl / a.S, p_si = p_tl - (p_tl / a.S) * a.S;
    int p_adelta = 0, p_wdelta = 0, p_tlc = 0;
    bool p_cok = false, p_wcok = false, p_live = true;
    auto prep = [&]() {
        const int c0 = p_cch * BK;
        p_tlc = p_tl;
        p_adelta = ((p_ri * a.SW + p_si) * a.SC + c0) * 4;
        p_cok = c0 + a_col < a.SC;                          // (SC == wC: one test serves both operands)
        p_wcok = c0 + b_col < a.wC;
        p_wdelta = (p_tl * a.wC + c0) * (B16 ? 2 : 4);
        ++p_tl;
        if (++p_si == a.S) { p_si = 0; ++p_ri; }
        if (p_tl == RS) { p_tl = 0; p_ri = 0; p_si = 0; ++p_cch; }
    };
    auto load_all = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            // (timing experiment, round 4: without this tap-mask arithmetic — wrong at the borders — the kernel runs 6 % faster)
            const unsigned ok = (unsigned)p_cok & (unsigned)((a_mask[j] >> p_tlc) & 1ull) & (unsigned)p_live;
            const unsigned off = ok ? (unsigned)(a_boff[j] + p_adelta) : OOB;
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const unsigned ok = (unsigned)b_ok[j] & (unsigned)p_wcok & (unsigned)p_live;
            const unsigned off = ok ? (unsigned)(b_boff[j] + p_wdelta) : OOB;
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));   // B16: 8 bf16
        }
    };
    auto store_all = [&](int buf) {
        unsigned short *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            *reinterpret_cast<u16x4 *>(A + (a_row + RPP * j) * LDK + a_col) = f2bf4(ra[j]);
        if constexpr (B16) {
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                if (BN >= 64 || b_row < BN)
                    *reinterpret_cast<f32x4 *>(B + (b_row + 64 * j) * LDK + b_col) = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j)      // (a_row + RPP * j < BN always: BJ = BN / RPP, a_row < RPP — no guard, no exec-mask branch)
                *reinterpret_cast<u16x4 *>(B + (a_row + RPP * j) * LDK + a_col) = f2bf4(rb[j]);
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

    if (kc_lo < kc_hi) {
        prep();
        load_all();
        store_all(0);
        p_live = kc_lo + 1 < kc_hi;
        prep();
        load_all();                         // tile kc_lo + 1 stays in registers until the first iteration stores it
    }
    __syncthreads();
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        const int buf = (kc - kc_lo) & 1;
        const unsigned short *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
        bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[kk][i] = *reinterpret_cast<const bf16x8 *>(A + ((wm * TM + i) * 32 + lr) * LDK + kk * 16 + lh * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[kk][j] = *reinterpret_cast<const bf16x8 *>(B + ((wn * TN + j) * 32 + lr) * LDK + kk * 16 + lh * 8);
        }
        // tile kc + 1 (loaded one iteration ago) -> the other LDS buffer, whose last readers passed the previous barrier;
        // then tile kc + 2 goes out.  (Staging BEHIND the matrix instructions instead — a whole iteration for the loads to
        // land — measured the same: the loop is bound by instruction issue, ~96 non-MFMA instructions per 8 MFMAs.)
        store_all(buf ^ 1);
        p_live = kc + 2 < kc_hi;
        prep();
        load_all();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
        __syncthreads();
    }

    // ---- epilogue (as csrc/conv.hip).  D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    double *sred = reinterpret_cast<double *>(lds16);   // [WM][BN][2], reuses the staging LDS
    const bool do_stats = a.stat_slab != nullptr && a.ksplit <= 1;
    const __amdgpu_buffer_rsrc_t bs_rs_y = make_srd(BNS ? a.bs_y : a.src, (long)a.M * a.DC * 4);
    const __amdgpu_buffer_rsrc_t bs_rs_z = make_srd(BNS && a.bs_z != nullptr ? a.bs_z : a.src, (long)a.M * a.DC * 4);
    const int mode_e = a.ksplit > 1 ? 2 : (a.accumulate ? 1 : 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ncol = n0 + (wn * TN + j) * 32 + lr;
        const bool n_ok = ncol < a.DC;
        const float bv = (a.bias != nullptr && n_ok) ? a.bias[ncol] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        float bs_m = 0.f, bs_i = 0.f, bs_sc = 0.f, bs_sh = 0.f;
        if constexpr (BNS) {
            if (n_ok && !a.bs_relu_bias) {
                bs_m = a.bs_mean[ncol]; bs_i = a.bs_invstd[ncol];
                if (a.bs_z == nullptr) {
                    if (a.bs_msc != nullptr) { bs_sc = a.bs_msc[ncol]; bs_sh = a.bs_msh[ncol]; }
                    else bs_sh = 1.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (BNS) {
                const bool use_z = a.bs_z != nullptr;
                const unsigned voff = n_ok ? (unsigned)(((m0 + (wm * TM + i) * 32 + 4 * lh) * a.DC + ncol) * 4) : 0xFFFFFFF0u;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float yv[8], zv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = hh * 8 + q;
                        const int soff = __builtin_amdgcn_readfirstlane(((e & 3) + 8 * (e >> 2)) * a.DC * 4);
                        yv[q] = a.bs_relu_bias ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bs_rs_y, voff, soff, 0));
                        zv[q] = use_z ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bs_rs_z, voff, soff, 0)) : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = hh * 8 + q;
                        const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        if (m < a.M && n_ok) {
                            float *p = a.dst + (long)m * a.DC + ncol;
                            float v = acc[i][j][e];
                            if (mode_e == 1) v += *p;
                            const bool on = use_z ? zv[q] > 0.f : rr_bn_affine(yv[q], bs_sc, bs_sh) > 0.f;
                            const float d = on ? v : 0.f;
                            *p = a.bs_relu_bias ? d : v;
                            s1 += d;
                            s2 += d * ((yv[q] - bs_m) * bs_i);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float v = acc[i][j][e] + bv;
                if (a.relu) v = v > 0.f ? v : 0.f;
                if (m < a.M && n_ok) {
                    long pix = m;
                    if constexpr (SO) {
                        const int n = m / hw, rem = m - n * hw;
                        const int h = rem / a.DW;
                        pix = ((long)n * a.OH + (h * a.osh + a.oh0)) * a.OW + ((rem - h * a.DW) * a.osw + a.ow0);
                    }
                    float *p = a.dst + pix * a.DC + ncol;
                    if (mode_e == 2) {
                        unsafeAtomicAdd(p, v);
                    } else {
                        if (mode_e == 1) v += *p;
                        *p = v;
                        s1 += v;
                        s2 += v * v;
                    }
                }
            }
        }
        if (do_stats) {
            double d1 = (double)s1, d2 = (double)s2;
            d1 += __shfl_xor(d1, 32, 64);
            d2 += __shfl_xor(d2, 32, 64);
            if (lh == 0) {
                const int cl = (wn * TN + j) * 32 + lr;
                sred[(wm * BN + cl) * 2 + 0] = d1;
                sred[(wm * BN + cl) * 2 + 1] = d2;
            }
        }
    }
    if (do_stats) {
        __syncthreads();
        if (t < BN && n0 + t < a.DC) {
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                d1 += sred[(w * BN + t) * 2 + 0];
                d2 += sred[(w * BN + t) * 2 + 1];
            }
            double *slab = a.stat_slab + (long)m_tile * 2 * a.DC;
            slab[n0 + t] = d1;
            slab[a.DC + n0 + t] = d2;
        }
    }
}

// column sums / sums of squares of y -> slab row 0 (split-K keeps the statistics out of the conv epilogue)
__global__ __launch_bounds__(256) void colstats_bf16_kernel(const float *y, long M, int C, double *slab)
{
    __shared__ double red[2][256 * 4];
    const int C4 = C / 4, lanes = 256 / C4;
    const int t = threadIdx.x, cq = t % C4, pl = t / C4;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pl < lanes)
        for (long p = (long)blockIdx.x * lanes + pl; p < M; p += (long)gridDim.x * lanes) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(y + p * C + cq * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += (double)v[e]; s2[e] += (double)v[e] * (double)v[e]; }
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][t * 4 + e] = s1[e]; red[1][t * 4 + e] = s2[e]; }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        double a1 = 0.0, a2 = 0.0;
        for (int l = 0; l < lanes; ++l) {
            a1 += red[0][(l * C4 + c / 4) * 4 + (c & 3)];
            a2 += red[1][(l * C4 + c / 4) * 4 + (c & 3)];
        }
        unsafeAtomicAdd(slab + c, a1);
        unsafeAtomicAdd(slab + C + c, a2);
    }
}

struct BnSumArgs {
    const float *y, *z, *mean, *invstd, *msc, *msh;
    double *slab, *sums;
    int relu_bias;
};

struct OutMap { int DH, DW, OH, OW, osh, osw, oh0, ow0; };   // explicit logical output grid + strided destination

size_t igemm_lds(int bn) { return sizeof(unsigned short) * 2 * (size_t)(BM + bn) * LDK; }

template <typename K>
int launch(K kern, int blocks, int gz, size_t lds, hipStream_t stream, const ConvArgs &args, const char *name)
{
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks, 1, gz), dim3(256), lds, stream, args);
    RR_CHECK_LAUNCH(name);
    return RR_OK;
}

int fprop_impl(const float *x, const float *w, const float *bias, float *y, double *stat_slab, int n, int h, int wd, int c,
               int k, int r, int s, int stride, int pad_h, int pad_w, int relu, int accumulate, hipStream_t stream,
               const BnSumArgs *bs = nullptr, const OutMap *om = nullptr, const unsigned short *w16 = nullptr)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && k > 0 && r > 0 && s > 0 && stride > 0, "rr_conv_fprop_bf16: bad dims");
    RR_CHECK_ARG(c % 4 == 0 && r * s <= 64, "rr_conv_fprop_bf16: C=%d must be a multiple of 4 and R*S <= 64 (fp32 path for the rest)", c);
    ConvArgs a{};
    a.src = x; a.w = w; a.dst = y; a.bias = bias; a.stat_slab = stat_slab;
    a.w16 = (w16 != nullptr && c % 8 == 0) ? w16 : nullptr;       // 16-byte loads of 8 channels
    RR_CHECK_ARG(w != nullptr || a.w16 != nullptr, "rr_conv_fprop_bf16: no filter");
    a.N = n; a.SH = h; a.SW = wd; a.SC = c;
    a.DH = (h + 2 * pad_h - r) / stride + 1; a.DW = (wd + 2 * pad_w - s) / stride + 1; a.DC = k;
    if (om != nullptr) {        // pads are LEADING pads; taps past the far edge are masked by the gather
        a.DH = om->DH; a.DW = om->DW; a.OH = om->OH; a.OW = om->OW; a.osh = om->osh; a.osw = om->osw; a.oh0 = om->oh0; a.ow0 = om->ow0;
    }
    RR_CHECK_ARG(a.DH > 0 && a.DW > 0, "rr_conv_fprop_bf16: empty output");
    a.R = r; a.S = s; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
    a.relu = relu; a.accumulate = accumulate; a.ksplit = 1;
    const long M = (long)n * a.DH * a.DW;
    RR_CHECK_ARG(M < (1l << 31) && (long)n * h * wd * c * 4 < (1l << 31) && (long)k * r * s * c * 4 < (1l << 31) && M * k * 4 < (1l << 31),
                 "rr_conv_fprop_bf16: tensors must stay below 2 GiB (32-bit buffer offsets)");
    RR_CHECK_ARG(om == nullptr || (bs == nullptr && stat_slab == nullptr), "rr_conv_fprop_bf16: strided destination without fused sums");
    a.M = (int)M; a.wK = k; a.wC = c;
    int bn = k > 64 ? 128 : (k > 32 ? 64 : 32);
    {
        static int force = -1;          // experiment switch: RR_BF16_BN=64 runs the wide layers on 128 x 64 tiles
        if (force < 0) { const char *e = getenv("RR_BF16_BN"); force = e ? atoi(e) : 0; }
        if (force == 64 && bn == 128) bn = 64;
    }
    if (bn == 128 && rr_cdiv(M, BM) * rr_cdiv(k, 128) <= rr_conv_small_tiles()) bn = 32;
    else if (bn == 128 && rr_cdiv(M, BM) * rr_cdiv(k, 128) <= rr_conv_mid_tiles()) bn = 64;
    const int blocks = rr_cdiv(M, BM) * rr_cdiv(k, bn);
    const int nk = rr_cdiv(c, BK) * r * s;
    int ks = (bias == nullptr && !relu && k % 4 == 0 && k <= 1024) ? rr_conv_pick_ksplit(blocks, nk) : 1;
    if (bs != nullptr && bs->relu_bias) ks = 1;
    if (om != nullptr) ks = 1;       // (the zero fill of a split-K destination would wipe the other parity classes)
    if (ks > 1) {
        a.ksplit = ks;
        if (!accumulate) hipMemsetAsync(y, 0, sizeof(float) * (size_t)M * k, stream);
    }
    const bool tiles_full = M % BM == 0;
    RR_CHECK_ARG(bs == nullptr || !bs->relu_bias || tiles_full, "rr_conv_dgrad_s1_relubias_bf16: N*H*W = %ld must be a multiple of 128", M);
    const bool fused = bs != nullptr && ks == 1 && tiles_full;
    if (fused) {
        a.stat_slab = bs->slab;
        a.bs_y = bs->y; a.bs_z = bs->z; a.bs_mean = bs->mean; a.bs_invstd = bs->invstd; a.bs_msc = bs->msc; a.bs_msh = bs->msh;
        a.bs_relu_bias = bs->relu_bias;
    }
    int rc;
    const char *name = "rr_conv_fprop_bf16";
#define RR_IG(BNv, BNSv, SOv)                                                                                      \
    (a.w16 ? launch(conv_igemm_bf16_kernel<BNv, BNSv, true, SOv>, blocks, ks, igemm_lds(BNv), stream, a, name)           \
           : launch(conv_igemm_bf16_kernel<BNv, BNSv, false, SOv>, blocks, ks, igemm_lds(BNv), stream, a, name))
    if (fused) rc = bn == 128 ? RR_IG(128, true, false) : bn == 64 ? RR_IG(64, true, false) : RR_IG(32, true, false);
    else if (a.osh) rc = bn == 128 ? RR_IG(128, false, true) : bn == 64 ? RR_IG(64, false, true) : RR_IG(32, false, true);
    else rc = bn == 128 ? RR_IG(128, false, false) : bn == 64 ? RR_IG(64, false, false) : RR_IG(32, false, false);
#undef RR_IG
    if (rc == RR_OK && bs != nullptr) {
        if (fused) return rr_bn_reduce_slab(bs->slab, (int)rr_cdiv(M, BM), k, bs->sums, stream);
        return rr_bn_bwd_reduce(y, bs->z, bs->y, bs->mean, bs->invstd, bs->msc, bs->msh, bs->sums, M, k, 1, stream);
    }
    if (rc == RR_OK && ks > 1 && stat_slab != nullptr) {
        hipMemsetAsync(stat_slab, 0, rr_conv_stat_slab_bytes(n, a.DH, a.DW, k), stream);
        const int lanes = 256 / (k / 4);
        int sblocks = rr_cdiv(M, (long)lanes * 8);
        if (sblocks > 256) sblocks = 256;
        hipLaunchKernelGGL(colstats_bf16_kernel, dim3(sblocks), dim3(256), 0, stream, y, M, k, stat_slab);
        RR_CHECK_LAUNCH("rr_conv_fprop_bf16(stats)");
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------
// wgrad: dW[ko][tap][c] += sum over a slice of the N*P*Q pixels of dY[m][ko] * X[pix(m,tap)][c].  GEMM M = K (ko), N = C,
// reduction over pixels.  Both operands are k-major in memory ([pixel][channel]): each is stored row-major into LDS as
// 32-column blocks [block][32 pixels][32 channels] (64-byte rows) and read through ds_read_b64_tr_b16, which hands every
// lane the 8 consecutive pixels of its channel (checked lane by lane on the device: tools/tr_probe.hip).
struct WgradArgs {
    const float *x, *dy;
    float *dw;
    int N, H, W, C, K, R, S, P, Q, stride, pad_h, pad_w;
    int M, chunks_per_split, mt, nt;
};

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

// 128 (ko) x 128 (c) tile per workgroup and tap; 2x2 waves of 64x64.  K-step = 32 pixels.
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const WgradArgs a)
{
    constexpr int BLK = 32 * 32 + 32;            // one 32-column block of a K-step: [32 pixels][32 channels] (+64 B: the 8-byte
                                                 // stores of a 16-lane group go to two blocks, on disjoint banks)
    constexpr int IMG = 4 * BLK;                 // 128 channels
    extern __shared__ __align__(16) unsigned short lds16[];
    unsigned short *As = lds16;                  // [2][IMG]  dY  (ko)
    unsigned short *Bs = lds16 + 2 * IMG;        // [2][IMG]  X   (c)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int RS = a.R * a.S;
    int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tap = logical % RS; logical /= RS;
    const int n_tile = logical % a.nt; logical /= a.nt;
    const int m_tile = logical % a.mt;
    const int split = logical / a.mt;
    const int r = tap / a.S, s = tap - r * a.S;
    const int ko0 = m_tile * 128, c0 = n_tile * 128;
    const int total_chunks = (a.M + BK - 1) / BK;
    const int kc_begin = split * a.chunks_per_split;
    int kc_end = kc_begin + a.chunks_per_split;
    if (kc_end > total_chunks) kc_end = total_chunks;
    if (kc_begin >= kc_end) return;

    // staging: thread = (pixel row t / 32 + 8 j, channel quad t % 32)
    const int s_col = (t & 31) * 4, s_row = t >> 5;
    const bool a_ok = ko0 + s_col < a.K, b_ok = c0 + s_col < a.C;
    const __amdgpu_buffer_rsrc_t rs_dy = make_srd(a.dy, (long)a.M * a.K * 4);
    const __amdgpu_buffer_rsrc_t rs_x = make_srd(a.x, (long)a.N * a.H * a.W * a.C * 4);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    // running (n, p, q) of this thread's four pixel rows
    int bn_[4], bp_[4], bq_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long m = (long)kc_begin * BK + s_row + 8 * j;
        const int pq = a.P * a.Q;
        bn_[j] = (int)(m / pq);
        const int rem = (int)(m - (long)bn_[j] * pq);
        bp_[j] = rem / a.Q;
        bq_[j] = rem - bp_[j] * a.Q;
    }
    f32x4 ra[4], rb[4];
    auto load_all = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long m = (long)kc * BK + s_row + 8 * j;
            const unsigned ok = (unsigned)a_ok & (unsigned)(m < a.M) & (unsigned)(kc < kc_end);
            const unsigned off = ok ? (unsigned)((m * a.K + ko0 + s_col) * 4) : OOB;
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ih = bp_[j] * a.stride - a.pad_h + r, iw = bq_[j] * a.stride - a.pad_w + s;
            const unsigned ok = (unsigned)b_ok & (unsigned)(bn_[j] < a.N) & (unsigned)((unsigned)ih < (unsigned)a.H) &
                                (unsigned)((unsigned)iw < (unsigned)a.W) & (unsigned)(kc < kc_end);
            const unsigned off = ok ? (unsigned)((((bn_[j] * a.H + ih) * a.W + iw) * a.C + c0 + s_col) * 4) : OOB;
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
            bq_[j] += BK;
            while (bq_[j] >= a.Q) {
                bq_[j] -= a.Q;
                if (++bp_[j] == a.P) { bp_[j] = 0; ++bn_[j]; }
            }
        }
    };
    auto store_all = [&](int buf) {
        unsigned short *A = As + buf * IMG, *B = Bs + buf * IMG;
        const int blk = (s_col >> 5) * BLK, cc = s_col & 31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<u16x4 *>(A + blk + (s_row + 8 * j) * 32 + cc) = f2bf4(ra[j]);
            *reinterpret_cast<u16x4 *>(B + blk + (s_row + 8 * j) * 32 + cc) = f2bf4(rb[j]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_all(kc_begin);
    store_all(0);
    load_all(kc_begin + 1);
    __syncthreads();
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int buf = (kc - kc_begin) & 1;
        const unsigned short *A = As + buf * IMG, *B = Bs + buf * IMG;
        bf16x8 fa[2][2], fb[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[kk][i] = lds_tr_frag(A + (wm * 2 + i) * BLK, kk * 16, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[kk][j] = lds_tr_frag(B + (wn * 2 + j) * BLK, kk * 16, lane);
        }
        store_all(buf ^ 1);
        load_all(kc + 2);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
        __syncthreads();
    }
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = c0 + (wn * 2 + j) * 32 + lr;
        if (c >= a.C) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ko = ko0 + (wm * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (ko < a.K) unsafeAtomicAdd(a.dw + ((long)ko * RS + tap) * a.C + c, acc[i][j][e]);
            }
    }
}

}  // namespace

extern "C" int rr_conv_fprop_bf16(const float *x, const float *w, const float *bias, float *y, double *stat_slab,
                                  int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h,
                                  int pad_w, int relu, const unsigned short *w_bf16, hipStream_t stream)
{
    return fprop_impl(x, w, bias, y, stat_slab, n, h, wd, c, k, r, s, stride, pad_h, pad_w, relu, 0, stream, nullptr, nullptr, w_bf16);
}

extern "C" int rr_conv_dgrad_s1_bf16(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                     int r, int s, int pad_h, int pad_w, int accumulate, const unsigned short *wt_bf16,
                                     hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1_bf16: pad must be in [0, kernel)");
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1_bf16: empty dy");
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate, stream,
                      nullptr, nullptr, wt_bf16);
}

extern "C" int rr_conv_dgrad_s1_bnsum_bf16(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                           int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_y,
                                           const float *prod_z, const float *prod_mean, const float *prod_invstd,
                                           const float *prod_mask_scale, const float *prod_mask_shift, double *slab,
                                           double *sums, const unsigned short *wt_bf16, hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1_bnsum_bf16: pad must be in [0, kernel)");
    RR_CHECK_ARG(prod_y && prod_mean && prod_invstd && slab && sums && (!prod_mask_scale == !prod_mask_shift),
                 "rr_conv_dgrad_s1_bnsum_bf16: the producer's y / mean / invstd and the two buffers are required");
    RR_CHECK_ARG(c % 4 == 0 && c <= 1024, "rr_conv_dgrad_s1_bnsum_bf16: C=%d must be a multiple of 4 and <= 1024", c);
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1_bnsum_bf16: empty dy");
    const BnSumArgs bs{prod_y, prod_z, prod_mean, prod_invstd, prod_mask_scale, prod_mask_shift, slab, sums, 0};
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate, stream, &bs,
                      nullptr, wt_bf16);
}

extern "C" int rr_conv_dgrad_s1_relubias_bf16(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                              int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_z,
                                              double *slab, double *sums, hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1_relubias_bf16: pad must be in [0, kernel)");
    RR_CHECK_ARG(prod_z && slab && sums, "rr_conv_dgrad_s1_relubias_bf16: the producer's output and the two buffers are required");
    RR_CHECK_ARG(c % 4 == 0 && k % 4 == 0 && c <= 1024, "rr_conv_dgrad_s1_relubias_bf16: C=%d, K=%d must be multiples of 4", c, k);
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1_relubias_bf16: empty dy");
    const BnSumArgs bs{prod_z, prod_z, nullptr, nullptr, nullptr, nullptr, slab, sums, 1};
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate, stream, &bs);
}

// Stride-2 data gradient on the forward kernel: dx[n, 2a+ph, 2b+pw, c] = sum over the taps r = r0 + 2i, s = s0 + 2j that
// reach output parity class (ph, pw) (r0 = (ph + pad_h) & 1) of dY[n, a + (ph+pad_h-r0)/2 - i, ..., k] * w[k][r][s][c] — per
// class a stride-1 correlation of dY with a sub-filter of ceil / floor (R/2) x (S/2) taps, written to every second pixel.
// The four sub-filters are packed, flipped and transposed ([c][i'][j'][k], i' = Rc-1-i), into caller scratch of
// k*r*s*c floats by one kernel; a 3x3 visits 1 / 2 / 2 / 4 taps instead of masking three quarters of a dilated filter.
__global__ __launch_bounds__(256) void weight_parity_pack_kernel(const float *w, float *wsub, int K, int C, int R, int S,
                                                                 int pad_h, int pad_w, long total)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % K);
        long rest = idx / K;
        const int tap = (int)(rest % (R * S));
        const int c = (int)(rest / (R * S));
        const int r = tap / S, s = tap - r * S;
        const int ph = (r - pad_h) & 1, pw = (s - pad_w) & 1;
        const int r0 = (ph + pad_h) & 1, s0 = (pw + pad_w) & 1;
        const int Rc = (R - r0 + 1) / 2, Sc = (S - s0 + 1) / 2;
        // class blocks in the order (0,0), (0,1), (1,0), (1,1); block size C * Rc * Sc * K
        long base = 0;
        for (int cl = 0; cl < ph * 2 + pw; ++cl) {
            const int q0 = ((cl >> 1) + pad_h) & 1, t0 = ((cl & 1) + pad_w) & 1;
            base += (long)C * (q0 < R ? (R - q0 + 1) / 2 : 0) * (t0 < S ? (S - t0 + 1) / 2 : 0) * K;
        }
        const int ii = Rc - 1 - (r - r0) / 2, jj = Sc - 1 - (s - s0) / 2;
        wsub[base + (((long)c * Rc + ii) * Sc + jj) * K + k] = w[((long)k * R * S + tap) * C + c];
    }
}

extern "C" int rr_conv_dgrad_s2_bf16(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k,
                                     int r, int s, int pad_h, int pad_w, int accumulate, float *wsub, hipStream_t stream)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && k > 0 && r > 0 && s > 0, "rr_conv_dgrad_s2_bf16: bad dims");
    RR_CHECK_ARG(c % 4 == 0 && k % 4 == 0 && r * s <= 64 && pad_h >= 0 && pad_w >= 0 && wsub != nullptr,
                 "rr_conv_dgrad_s2_bf16: C, K multiples of 4, R*S <= 64, scratch of k*r*s*c floats required");
    const int p = (h + 2 * pad_h - r) / 2 + 1, q = (wd + 2 * pad_w - s) / 2 + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s2_bf16: empty dy");
    const long total = (long)k * c * r * s;
    hipLaunchKernelGGL(weight_parity_pack_kernel, dim3((int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                       stream, w, wsub, k, c, r, s, pad_h, pad_w, total);
    RR_CHECK_LAUNCH("rr_conv_dgrad_s2_bf16(pack)");
    int Rc[4], Sc[4], lead_h[4], lead_w[4];
    bool any_empty = false;
    for (int cl = 0; cl < 4; ++cl) {
        const int ph = cl >> 1, pw = cl & 1;
        const int r0 = (ph + pad_h) & 1, s0 = (pw + pad_w) & 1;
        Rc[cl] = r0 < r ? (r - r0 + 1) / 2 : 0;
        Sc[cl] = s0 < s ? (s - s0 + 1) / 2 : 0;
        lead_h[cl] = (Rc[cl] - 1) - (ph + pad_h - r0) / 2;
        lead_w[cl] = (Sc[cl] - 1) - (pw + pad_w - s0) / 2;
        const int Hc = (h - ph + 1) / 2, Wc = (wd - pw + 1) / 2;
        if (Hc > 0 && Wc > 0 && Rc[cl] * Sc[cl] == 0) any_empty = true;
        RR_CHECK_ARG(Rc[cl] * Sc[cl] == 0 || (lead_h[cl] >= 0 && lead_w[cl] >= 0), "rr_conv_dgrad_s2_bf16: unsupported padding %d,%d", pad_h, pad_w);
    }
    // parity classes no tap reaches (a 1x1 stride 2: three of four) are zero
    if (any_empty && !accumulate) hipMemsetAsync(dx, 0, sizeof(float) * (size_t)n * h * wd * c, stream);
    long base = 0;
    for (int cl = 0; cl < 4; ++cl) {
        const int ph = cl >> 1, pw = cl & 1;
        const int Hc = (h - ph + 1) / 2, Wc = (wd - pw + 1) / 2;
        const long blk = (long)c * Rc[cl] * Sc[cl] * k;
        if (blk > 0 && Hc > 0 && Wc > 0) {
            const OutMap om{Hc, Wc, h, wd, 2, 2, ph, pw};
            const int rc = fprop_impl(dy, wsub + base, nullptr, dx, nullptr, n, p, q, k, c, Rc[cl], Sc[cl], 1, lead_h[cl], lead_w[cl], 0,
                                      accumulate, stream, nullptr, &om);
            if (rc != RR_OK) return rc;
        }
        base += blk;
    }
    return RR_OK;
}

extern "C" int rr_conv_wgrad_bf16(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k,
                                  int r, int s, int stride, int pad_h, int pad_w, int out_h, int out_w, hipStream_t stream)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && k > 0 && r > 0 && s > 0 && stride > 0, "rr_conv_wgrad_bf16: bad dims");
    RR_CHECK_ARG(c % 4 == 0 && k % 4 == 0, "rr_conv_wgrad_bf16: C=%d, K=%d must be multiples of 4 (fp32 path for the rest)", c, k);
    WgradArgs a{};
    a.x = x; a.dy = dy; a.dw = dw;
    a.N = n; a.H = h; a.W = wd; a.C = c; a.K = k; a.R = r; a.S = s;
    a.P = out_h > 0 ? out_h : (h + 2 * pad_h - r) / stride + 1;
    a.Q = out_w > 0 ? out_w : (wd + 2 * pad_w - s) / stride + 1;
    a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
    const long M = (long)n * a.P * a.Q;
    RR_CHECK_ARG(M > 0 && M < (1l << 31) && M * k * 4 < (1l << 31) && (long)n * h * wd * c * 4 < (1l << 31),
                 "rr_conv_wgrad_bf16: tensors must stay below 2 GiB (32-bit buffer offsets)");
    a.M = (int)M;
    a.mt = rr_cdiv(k, 128); a.nt = rr_cdiv(c, 128);
    const int tiles = a.mt * a.nt * r * s;
    const int total_chunks = rr_cdiv(M, BK);
    // pixel splits: fill the resident-workgroup slots (256 CUs x 4) once, never fewer than 16 K-steps per split
    const int slots = 1024;
    int splits = tiles < slots ? slots / tiles : 1;
    if (splits > rr_cdiv(total_chunks, 16)) splits = rr_cdiv(total_chunks, 16);
    if (splits < 1) splits = 1;
    a.chunks_per_split = rr_cdiv(total_chunks, splits);
    splits = rr_cdiv(total_chunks, a.chunks_per_split);
    const size_t lds = sizeof(unsigned short) * 4 * 4 * (32 * 32 + 32);  // 2 operands x 2 buffers x 4 blocks of 32 x 32 (+ pad)
    hipLaunchKernelGGL(conv_wgrad_bf16_kernel, dim3(tiles * splits), dim3(256), lds, stream, a);
    RR_CHECK_LAUNCH("rr_conv_wgrad_bf16");
    return RR_OK;
}
