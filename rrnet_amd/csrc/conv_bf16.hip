// 16-bit-operand NHWC convolutions on the gfx950 matrix cores, fp32 accumulation.  Two arithmetics share the kernels:
//   bf16   (rr_conv_*_bf16, cfg.Model.bf16 — BASELINE config 4): each operand rounded to ONE bf16 value;
//   f16x3  (rr_conv_*_f16x3, cfg.Model.conv_math — an fp32-class arithmetic): each operand the sum of TWO fp16 values after a
//          power-of-two scaling by the tensor's maximum, three products per tile (template parameters SP = 2, F16; the
//          entry points and their contract: the "Split-operand entry points" section at the end of this file).
// The headline configuration stays on csrc/conv.hip (v_mfma_f32_32x32x2_f32).  What follows describes the bf16 form.
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
#include <type_traits>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

int rr_conv_pick_ksplit(int blocks, int nk);      // csrc/conv.hip: the occupancy model shared with the fp32 kernels
int rr_conv_small_tiles();
int rr_conv_mid_tiles();

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = BK + 8;          // [row][k] image row stride in bf16 (80 bytes): conflict-free ds_read_b128

__device__ __forceinline__ u16x4 f2bf4(f32x4 v) { return __builtin_bit_cast(u16x4, __builtin_convertvector(v, bf16x4)); }

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// v = parts[0] + parts[1] (+ parts[2]) + O(2^-16 / 2^-24 |v|): each part the round-to-nearest-even bf16 of what is left
// (the subtractions are exact in fp32).  Not for |v| >= 2^127.5 (the hi part rounds to infinity) or non-finite v.
// F16: the parts are fp16 (11 significant bits each: two parts carry 22 bits), v pre-multiplied by the operand's power-of-two
// scale (ConvArgs::amax_*) so that the tensor's largest magnitude sits in [2^14, 2^15) of fp16's range.
template <int SP, bool F16 = false>
__device__ __forceinline__ void split_bf4(f32x4 v, u16x4 (&parts)[SP], float scale = 1.f)
{
    if constexpr (F16) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        f32x2 lo2 = {v[0] * scale, v[1] * scale}, hi2 = {v[2] * scale, v[3] * scale};
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const f16x2 h0 = __builtin_convertvector(lo2, f16x2), h1 = __builtin_convertvector(hi2, f16x2);
            const u32x2 pk = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
            parts[i] = __builtin_bit_cast(u16x4, pk);
            if (i + 1 < SP) {
                lo2 -= __builtin_convertvector(h0, f32x2);
                hi2 -= __builtin_convertvector(h1, f32x2);
            }
        }
        return;
    }
    // pairs: one v_cvt_pk_bf16_f32 rounds two values, shift / mask widen them again (5.5 vector instructions per element for
    // three parts; element-wise conversion, which is what __builtin_convertvector of the 4-vector compiles to, took 7.9)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    f32x2 lo2 = {v[0], v[1]}, hi2 = {v[2], v[3]};
#pragma unroll
    for (int i = 0; i < SP; ++i) {
        const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf16x2));
        const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf16x2));
        const u32x2 pk = {p0, p1};
        parts[i] = __builtin_bit_cast(u16x4, pk);
        if (i + 1 < SP) {
            lo2[0] -= __builtin_bit_cast(float, p0 << 16);
            lo2[1] -= __builtin_bit_cast(float, p0 & 0xffff0000u);
            hi2[0] -= __builtin_bit_cast(float, p1 << 16);
            hi2[1] -= __builtin_bit_cast(float, p1 & 0xffff0000u);
        }
    }
}

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
    // F16 instantiations: bit patterns of max|src| and max|w| (rr_absmax_bits), from which the kernel derives the power-of-two
    // operand scales; null: scale 1
    const unsigned *amax_src, *amax_w;
    // workgroup -> tile map: 1 = every XCD works on ONE column tile (n_tile = xcd % ntiles): the filter rows an XCD re-reads for
    // every row tile (2.4 MB of fp16 hi + lo for 128 of 256 output channels) then stay in its 4 MiB L2; with the column tiles
    // interleaved on every XCD (4.7 MB) they do not, and every re-read goes to the memory side
    int xcd_n;
};

// Operand scale of the split-operand (f16x3) arithmetic from the word holding the bit pattern of max|tensor|: 2^(14 - floor(log2 max)),
// an exact power of two that places the maximum in [2^14, 2^15).  Exponents below 15 (max < 2^-112, zero and denormals included)
// are clamped: (268 - e) << 23 would otherwise reach the Inf / NaN encodings (e <= 13) — such a tensor is scaled by 2^126 at most
// and simply keeps fewer than 22 bits.  The epilogue undoes the two scales one after the other (each factor an exact power of two:
// their PRODUCT overflows / its inverse underflows for tiny operands).
__device__ __forceinline__ float split_scale_from_bits(unsigned bits)
{
    unsigned e = (bits >> 23) & 0xffu;
    e = e < 15u ? 15u : e;
    return __builtin_bit_cast(float, (268u - e) << 23);
}
__device__ __forceinline__ float split_scale_inv(float sc) { return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, sc)); }

// 128 x BN output tile, 256 threads = 4 waves (BN 128: 2x2 waves of 64x64; BN 64: 4x1 waves of 32x64; BN 32: 4x1 of 32x32)
// B16: the filter comes as bf16 (a.w16; C % 8 == 0): half the B loads, no converts, 16-byte LDS stores for the B image.
// SO: strided destination (ConvArgs::osh; the stride-2 data gradient's parity-class launches) — its own instantiation: folded
// into the plain kernel the per-element pixel arithmetic cost 20 registers and the third workgroup per CU (585 -> 524 TFLOP/s).
// SP: operand split (csrc header of the split section below).  1: each operand rounded to ONE bf16 value.  2 / 3: each fp32
// operand is written as a sum of 2 / 3 bf16 values (hi + mid [+ lo], each the round-to-nearest of what the previous ones left)
// and the products hi*hi + (hi*mid + mid*hi) [+ hi*lo + mid*mid + lo*hi] are accumulated — 3 / 6 matrix instructions per
// product tile instead of 1, 2^-16 / 2^-24 relative per product.  The small products go to their own accumulator (added to
// the hi*hi one in the epilogue), so their rounding never touches the main sum.
// WS (wave specialisation, 512 threads): waves 0-3 only read LDS and issue matrix instructions, waves 4-7 only fetch, split
// and stage the next tile.  Each SIMD then holds one wave of each kind and overlaps them in hardware — with 3 / 6 matrix
// instructions per product tile the loop is matrix-bound only if nothing else sits in the matrix waves' instruction stream.
template <int BN, bool BNS, bool B16 = false, bool SO = false, int SP = 1, bool WS = false, bool F16 = false>
__global__ __launch_bounds__(WS ? 512 : 256, (!WS && SP == 2) ? 2 : 1) void conv_igemm_bf16_kernel(const ConvArgs a)
{
    typedef typename std::conditional<F16, f16x8, bf16x8>::type frag_t;
    // operand scales (F16): 2^(14 - floor(log2(max|tensor|))), exact powers of two; the product is undone in the epilogue
    float sc_a = 1.f, sc_b = 1.f, sc_ia = 1.f, sc_ib = 1.f;
    if constexpr (F16) {
        auto scale_of = [](const unsigned *p) {
            if (p == nullptr) return 1.f;
            return split_scale_from_bits(__builtin_amdgcn_readfirstlane(*p));                // 2^(14 - (e - 127))
        };
        sc_a = scale_of(a.amax_src);
        sc_b = scale_of(a.amax_w);
        sc_ia = split_scale_inv(sc_a);
        sc_ib = split_scale_inv(sc_b);
    }
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
    unsigned short *As = lds16;                      // [2][SP][A_ELEMS]
    unsigned short *Bs = lds16 + 2 * SP * A_ELEMS;   // [2][SP][B_ELEMS]

    const int t = WS ? (threadIdx.x & 255) : threadIdx.x;     // index inside the thread's role group
    const bool producer = WS && threadIdx.x >= 256;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = (a.DC + BN - 1) / BN;
    int n_tile, m_tile;
    if (a.xcd_n) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, groups = 8 / ntiles;
        n_tile = xcd % ntiles;
        m_tile = (xcd / ntiles) * ((int)gridDim.x / ntiles / groups) + idx;
    } else {
        const int logical = xcd_remap(blockIdx.x, gridDim.x);
        n_tile = logical % ntiles;
        m_tile = logical / ntiles;
    }
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
    const __amdgpu_buffer_rsrc_t rs_w = B16 ? make_srd(a.w16, (long)a.wK * RS * a.wC * 2 * SP) : make_srd(a.w, (long)a.wK * RS * a.wC * 4);
    const int w16_image = a.wK * RS * a.wC * 2;      // B16, SP > 1: the filter's hi / mid / lo images follow each other
    constexpr unsigned OOB = 0xFFFFFFF0u;       // beyond any (< 2 GiB) tensor: the hardware returns 0, no select on the data

    constexpr int NBI = B16 ? SP : 1;            // filter images fetched per K-step
    f32x4 ra[AJ], rb[NBI][BJ];
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
    auto load_into = [&](f32x4 (&ra)[AJ], f32x4 (&rb)[NBI][BJ]) {
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
#pragma unroll
            for (int sp = 0; sp < (B16 ? SP : 1); ++sp) {
                const unsigned off = ok ? (unsigned)(b_boff[j] + p_wdelta + sp * w16_image) : OOB;
                rb[sp][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));   // B16: 8 values of 16 bits
            }
        }
    };
    auto load_all = [&]() { load_into(ra, rb); };
    auto store_from = [&](f32x4 (&ra)[AJ], f32x4 (&rb)[NBI][BJ], int buf) {
        unsigned short *A = As + buf * SP * A_ELEMS, *B = Bs + buf * SP * B_ELEMS;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            u16x4 parts[SP];
            split_bf4<SP, F16>(ra[j], parts, sc_a);
#pragma unroll
            for (int sp = 0; sp < SP; ++sp)
                *reinterpret_cast<u16x4 *>(A + sp * A_ELEMS + (a_row + RPP * j) * LDK + a_col) = parts[sp];
        }
        if constexpr (B16) {
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                if (BN >= 64 || b_row < BN)
#pragma unroll
                    for (int sp = 0; sp < SP; ++sp)
                        *reinterpret_cast<f32x4 *>(B + sp * B_ELEMS + (b_row + 64 * j) * LDK + b_col) = rb[sp][j];
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {    // (a_row + RPP * j < BN always: BJ = BN / RPP, a_row < RPP — no guard, no exec-mask branch)
                u16x4 parts[SP];
                split_bf4<SP, F16>(rb[0][j], parts, sc_b);
#pragma unroll
                for (int sp = 0; sp < SP; ++sp)
                    *reinterpret_cast<u16x4 *>(B + sp * B_ELEMS + (a_row + RPP * j) * LDK + a_col) = parts[sp];
            }
        }
    };

    auto store_all = [&](int buf) { store_from(ra, rb, buf); };

    f32x16 acc[TM][TN], acl[SP > 1 ? TM : 1][SP > 1 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if constexpr (SP > 1) acl[i][j][e] = 0.f;
            }
    const int lr = lane & 31, lh = lane >> 5;

    if constexpr (WS) {
        // LDS stores / reads of this wave retired, then the barrier: no vmcnt wait (the producers' loads of tile kc + 2 stay in
        // flight across it, __syncthreads() would drain them).  "memory": the compiler moves no LDS access across it.
#define RR_WS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
        if (producer) {
            // Two tiles in flight in registers (ra / rb and ra2 / rb2 alternate): a tile's loads are issued a whole K-step before
            // its split + LDS store needs them.  The loads are inline assembly and the waits are counted BY HAND: hipcc's own wait
            // insertion is conservative at a loop header (registers loaded before the back edge get vmcnt(0) at their first use,
            // whatever was issued after them), which exposed a full memory latency in every second K-step of this loop.
            // Every load's destination is an output of its asm statement and an in/out operand of the wait that retires it, so
            // nothing the compiler schedules can read it in between.
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            auto srd4 = [](const void *p, long bytes) {
                const unsigned long long u = reinterpret_cast<unsigned long long>(p);
                i32x4 r;
                r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
                r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
                r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
                r[3] = 0x00020000;
                return r;
            };
            const i32x4 q_src = srd4(a.src, (long)a.N * a.SH * a.SW * a.SC * 4);
            const i32x4 q_w = B16 ? srd4(a.w16, (long)a.wK * RS * a.wC * 2 * SP) : srd4(a.w, (long)a.wK * RS * a.wC * 4);
            constexpr int NLOADS = AJ + NBI * BJ;       // loads per tile and thread
            static_assert(NLOADS == 8 || NLOADS == 6 || NLOADS == 10, "the wait count below is a literal");
            auto fetch = [&](f32x4 (&ra)[AJ], f32x4 (&rb)[NBI][BJ]) {
#pragma unroll
                for (int j = 0; j < AJ; ++j) {
                    const unsigned ok = (unsigned)p_cok & (unsigned)((a_mask[j] >> p_tlc) & 1ull) & (unsigned)p_live;
                    const unsigned off = ok ? (unsigned)(a_boff[j] + p_adelta) : OOB;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ra[j]) : "v"(off), "s"(q_src) : "memory");
                }
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const unsigned ok = (unsigned)b_ok[j] & (unsigned)p_wcok & (unsigned)p_live;
#pragma unroll
                    for (int sp = 0; sp < NBI; ++sp) {
                        const unsigned off = ok ? (unsigned)(b_boff[j] + p_wdelta + sp * w16_image) : OOB;
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[sp][j]) : "v"(off), "s"(q_w) : "memory");
                    }
                }
            };
            // all but the newest NLOADS loads have landed: the tile in (ra, rb) is complete, the next one may still be in flight
            auto landed = [&](f32x4 (&ra)[AJ], f32x4 (&rb)[NBI][BJ]) {
                static_assert(AJ == 4, "operand list below");
                if constexpr (NLOADS == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]) :: "memory");
                else if constexpr (NLOADS == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]) :: "memory");
                else asm volatile("s_waitcnt vmcnt(10)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]) :: "memory");
#pragma unroll
                for (int sp = 0; sp < NBI; ++sp)
#pragma unroll
                    for (int j = 0; j < BJ; ++j) asm volatile("" : "+v"(rb[sp][j]) :: "memory");   // (ordered behind the wait: both volatile)
            };
            f32x4 ra2[AJ], rb2[NBI][BJ];
            prep();
            fetch(ra, rb);                          // tile 0
            p_live = kc_lo + 1 < kc_hi;
            prep();
            fetch(ra2, rb2);                        // tile 1
            landed(ra, rb);
            store_from(ra, rb, 0);
            p_live = kc_lo + 2 < kc_hi;
            prep();
            fetch(ra, rb);                          // tile 2
            RR_WS_BARRIER();
            for (int kc = kc_lo; kc < kc_hi; kc += 2) {
                landed(ra2, rb2);
                store_from(ra2, rb2, 1);            // tile kc + 1: the readers of tile kc - 1 passed the previous barrier
                p_live = kc + 3 < kc_hi;
                prep();
                fetch(ra2, rb2);
                RR_WS_BARRIER();
                if (kc + 1 >= kc_hi) break;
                landed(ra, rb);
                store_from(ra, rb, 0);              // tile kc + 2
                p_live = kc + 4 < kc_hi;
                prep();
                fetch(ra, rb);
                RR_WS_BARRIER();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the two tiles past the end (all-zero loads) before the registers are reused
        } else {
            // Matrix waves, rotated by half a K-step against the barrier: the fragments of a tile's second half and of the next
            // tile's first half are fetched while the other half's matrix instructions run, so no LDS latency is exposed.
            //   barrier #kc+1 sits between the two halves of tile kc: by then this wave holds ALL of tile kc in registers
            //   (its buffer is free for tile kc + 2) and the producers have finished tile kc + 1.
            frag_t f0a[SP][TM], f0b[SP][TN], f1a[SP][TM], f1b[SP][TN];
            auto frags = [&](frag_t (&fa)[SP][TM], frag_t (&fb)[SP][TN], int buf, int kk) {
                const unsigned short *A = As + buf * SP * A_ELEMS, *B = Bs + buf * SP * B_ELEMS;
#pragma unroll
                for (int sp = 0; sp < SP; ++sp) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[sp][i] = *reinterpret_cast<const frag_t *>(A + sp * A_ELEMS + ((wm * TM + i) * 32 + lr) * LDK + kk * 16 + lh * 8);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[sp][j] = *reinterpret_cast<const frag_t *>(B + sp * B_ELEMS + ((wn * TN + j) * 32 + lr) * LDK + kk * 16 + lh * 8);
                }
            };
            auto mma = [&](frag_t (&fa)[SP][TM], frag_t (&fb)[SP][TN]) {
#define RR_MM(ACC, PA, PB)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                      \
        ACC[i][j] = mfma16(fa[PA][i], fb[PB][j], ACC[i][j]);
                RR_MM(acc, 0, 0)
                if constexpr (SP >= 2) { RR_MM(acl, 0, 1) RR_MM(acl, 1, 0) }
                if constexpr (SP == 3) { RR_MM(acl, 1, 1) RR_MM(acl, 0, 2) RR_MM(acl, 2, 0) }
#undef RR_MM
            };
            __builtin_amdgcn_s_setprio(1);
            RR_WS_BARRIER();
            frags(f0a, f0b, 0, 0);
            for (int kc = kc_lo; kc < kc_hi; ++kc) {
                const int buf = (kc - kc_lo) & 1;
                frags(f1a, f1b, buf, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(f0a, f0b);
                __builtin_amdgcn_sched_barrier(0);
                RR_WS_BARRIER();
                frags(f0a, f0b, buf ^ 1, 0);        // (after the last tile: a buffer of zeros, never used)
                __builtin_amdgcn_sched_barrier(0);
                mma(f1a, f1b);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
#undef RR_WS_BARRIER
    } else {
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
        const unsigned short *A = As + buf * SP * A_ELEMS, *B = Bs + buf * SP * B_ELEMS;
        // SP == 1: the fragments of both 16-wide halves up front (as tuned).  SP > 1: one half at a time — half the fragment
        // registers, which keeps the split kernel within 256 registers (two workgroups per CU)
        constexpr int NKK = SP > 1 ? 1 : 2;
        frag_t fa[NKK][SP][TM], fb[NKK][SP][TN];
        auto frags = [&](int slot, int kk) {
#pragma unroll
            for (int sp = 0; sp < SP; ++sp) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[slot][sp][i] = *reinterpret_cast<const frag_t *>(A + sp * A_ELEMS + ((wm * TM + i) * 32 + lr) * LDK + kk * 16 + lh * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[slot][sp][j] = *reinterpret_cast<const frag_t *>(B + sp * B_ELEMS + ((wn * TN + j) * 32 + lr) * LDK + kk * 16 + lh * 8);
            }
        };
        auto mma = [&](int slot) {
            // (product outermost, tiles inner: consecutive matrix instructions never share an accumulator)
#define RR_MM(ACC, PA, PB)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                      \
        ACC[i][j] = mfma16(fa[slot][PA][i], fb[slot][PB][j], ACC[i][j]);
            if constexpr (SP == 3) { RR_MM(acl, 0, 2) RR_MM(acl, 2, 0) RR_MM(acl, 1, 1) }
            if constexpr (SP >= 2) { RR_MM(acl, 0, 1) RR_MM(acl, 1, 0) }
            RR_MM(acc, 0, 0)
#undef RR_MM
        };
        frags(0, 0);
        if constexpr (SP == 1) frags(1, 1);
        // tile kc + 1 (loaded one iteration ago) -> the other LDS buffer, whose last readers passed the previous barrier;
        // then tile kc + 2 goes out.  (Staging BEHIND the matrix instructions instead — a whole iteration for the loads to
        // land — measured the same: the loop is bound by instruction issue, ~96 non-MFMA instructions per 8 MFMAs.)
        store_all(buf ^ 1);
        p_live = kc + 2 < kc_hi;
        prep();
        load_all();
        mma(0);
        if constexpr (SP == 1) {
            mma(1);
        } else {
            frags(0, 1);
            mma(0);
        }
        __syncthreads();
    }
    }
    if constexpr (SP > 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] += acl[i][j];
                if constexpr (F16) acc[i][j] = acc[i][j] * sc_ia * sc_ib;
            }
    }

    // ---- epilogue (as csrc/conv.hip).  D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    double *sred = reinterpret_cast<double *>(lds16);   // [WM][BN][2], reuses the staging LDS
    const bool do_stats = a.stat_slab != nullptr && a.ksplit <= 1;
    const __amdgpu_buffer_rsrc_t bs_rs_y = make_srd(BNS ? a.bs_y : a.src, (long)a.M * a.DC * 4);
    const __amdgpu_buffer_rsrc_t bs_rs_z = make_srd(BNS && a.bs_z != nullptr ? a.bs_z : a.src, (long)a.M * a.DC * 4);
    const int mode_e = a.ksplit > 1 ? 2 : (a.accumulate ? 1 : 0);
    if (!producer) {
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
    }
    if (do_stats) {
        __syncthreads();
        if (t < BN && n0 + t < a.DC && !producer) {
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

// filter [K][R][S][C] fp32 -> its two fp16 parts, scaled by the power of two the convolution kernel derives from the same word:
// image 0 = hi, image 1 = lo (each K*R*S*C halves).  A 2.4 MB filter: ~5 us, against 48 of the 149 vector instructions per
// K-step and thread that splitting the filter tile inside the convolution costs.
__global__ __launch_bounds__(256) void weight_split_f16_kernel(const f32x4 *w, const unsigned *amax, unsigned short *out, long n4)
{
    const float scale = split_scale_from_bits(*amax);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        u16x4 parts[2];
        split_bf4<2, true>(w[i], parts, scale);
        *reinterpret_cast<u16x4 *>(out + 4 * i) = parts[0];
        *reinterpret_cast<u16x4 *>(out + 4 * (n4 + i)) = parts[1];
    }
}

size_t igemm_lds(int bn, int sp = 1) { return sizeof(unsigned short) * 2 * (size_t)sp * (BM + bn) * LDK; }

template <typename K>
int launch(K kern, int blocks, int gz, size_t lds, hipStream_t stream, const ConvArgs &args, const char *name, int threads = 256)
{
    if (lds > 48 * 1024)
        RR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), name);
    hipLaunchKernelGGL(kern, dim3(blocks, 1, gz), dim3(threads), lds, stream, args);
    RR_CHECK_LAUNCH(name);
    return RR_OK;
}

int fprop_impl(const float *x, const float *w, const float *bias, float *y, double *stat_slab, int n, int h, int wd, int c,
               int k, int r, int s, int stride, int pad_h, int pad_w, int relu, int accumulate, hipStream_t stream,
               const BnSumArgs *bs = nullptr, const OutMap *om = nullptr, const unsigned short *w16 = nullptr,
               int split = 0, const unsigned *amax_src = nullptr, const unsigned *amax_w = nullptr,
               const unsigned short *w_split = nullptr)
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
    if (bn == 128 && rr_cdiv(M, BM) * rr_cdiv(k, 128) <= rr_conv_small_tiles()) bn = 32;
    else if (bn == 128 && rr_cdiv(M, BM) * rr_cdiv(k, 128) <= rr_conv_mid_tiles()) bn = 64;
    const int blocks = rr_cdiv(M, BM) * rr_cdiv(k, bn);
    {
        const int nt = rr_cdiv(k, bn), mt = rr_cdiv(M, BM);
        a.xcd_n = (nt > 1 && 8 % nt == 0 && mt % (8 / nt) == 0) ? 1 : 0;
    }
    const int nk = rr_cdiv(c, BK) * r * s;
    int ks = (bias == nullptr && !relu && k % 4 == 0 && k <= 1024) ? rr_conv_pick_ksplit(blocks, nk) : 1;
    if (bs != nullptr && bs->relu_bias) ks = 1;
    if (om != nullptr) ks = 1;       // (the zero fill of a split-K destination would wipe the other parity classes)
    if (ks > 1) {
        a.ksplit = ks;
        // the split-K destination and (filled by colstats_bf16_kernel after the launch) the statistics slab: one zero-fill launch
        const bool zslab = stat_slab != nullptr && bs == nullptr;
        RR_CHECK_HIP(rr_zero2(accumulate ? nullptr : y, accumulate ? 0 : sizeof(float) * (size_t)M * k, zslab ? stat_slab : nullptr,
                              zslab ? rr_conv_stat_slab_bytes(n, a.DH, a.DW, k) : 0, stream), "rr_conv_fprop_bf16");
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
    // split operands (rr_conv_*_f16x3): two fp16 parts per operand, three matrix instructions per product tile
#define RR_SX(BNv, BNSv, SOv) launch(conv_igemm_bf16_kernel<BNv, BNSv, false, SOv, 2, false, true>, blocks, ks, igemm_lds(BNv, 2), stream, a, name)
    if (split) {
        // (A wave-specialised 512-thread form of this kernel — round 4 — measured 2.84 ms against 2.27 ms for two 256-thread
        // workgroups per CU at 256 -> 256 3x3 on 8 x 256 x 256: a SIMD does not overlap one wave's matrix instructions with another
        // wave's vector instructions (tools/coissue_probe.hip).  Its launch arm was removed in round 6.)
        a.w16 = nullptr;
        a.amax_src = amax_src; a.amax_w = amax_w;
        name = "rr_conv_fprop_f16x3";
        if (w_split != nullptr && c % 8 == 0 && bn == 128) {
            // the filter's two fp16 images, made by rr_weight_split_f16 (once per optimizer step, outside the backward pass): the B16 instantiation
            a.w16 = w_split;
#define RR_SB(BNSv, SOv) launch(conv_igemm_bf16_kernel<128, BNSv, true, SOv, 2, false, true>, blocks, ks, igemm_lds(128, 2), stream, a, name)
            rc = fused ? RR_SB(true, false) : (a.osh ? RR_SB(false, true) : RR_SB(false, false));
#undef RR_SB
        } else
        if (fused) rc = bn == 128 ? RR_SX(128, true, false) : bn == 64 ? RR_SX(64, true, false) : RR_SX(32, true, false);
        else if (a.osh) rc = bn == 128 ? RR_SX(128, false, true) : bn == 64 ? RR_SX(64, false, true) : RR_SX(32, false, true);
        else rc = bn == 128 ? RR_SX(128, false, false) : bn == 64 ? RR_SX(64, false, false) : RR_SX(32, false, false);
    } else
    if (fused) rc = bn == 128 ? RR_IG(128, true, false) : bn == 64 ? RR_IG(64, true, false) : RR_IG(32, true, false);
    else if (a.osh) rc = bn == 128 ? RR_IG(128, false, true) : bn == 64 ? RR_IG(64, false, true) : RR_IG(32, false, true);
    else rc = bn == 128 ? RR_IG(128, false, false) : bn == 64 ? RR_IG(64, false, false) : RR_IG(32, false, false);
#undef RR_IG
#undef RR_SX
    if (rc == RR_OK && bs != nullptr) {
        if (fused) return rr_bn_reduce_slab(bs->slab, (int)rr_cdiv(M, BM), k, bs->sums, stream);
        return rr_bn_bwd_reduce(y, bs->z, bs->y, bs->mean, bs->invstd, bs->msc, bs->msh, bs->sums, M, k, 1, stream);
    }
    if (rc == RR_OK && ks > 1 && stat_slab != nullptr) {
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
    const unsigned *amax_x, *amax_dy;       // split instantiation: bit patterns of max|x|, max|dy| (rr_absmax_bits)
};

template <typename F = bf16x8>
__device__ __forceinline__ F lds_tr_frag(const unsigned short *img, int k0, int lane)
{
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned short *a = img + (k0 + 8 * (g >> 1) + q) * 32 + 16 * (g & 1) + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a + 4 * 32));
    union { s16x4 h[2]; F v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}

// 128 (ko) x 128 (c) tile per workgroup and tap; 2x2 waves of 64x64.  K-step = 32 pixels.
// SP = 2, F16: both operands split into two fp16 parts (see conv_igemm_bf16_kernel), three matrix instructions per tile.
template <int SP = 1, bool F16 = false>
__global__ __launch_bounds__(256, SP > 1 ? 2 : 1) void conv_wgrad_bf16_kernel(const WgradArgs a)
{
    typedef typename std::conditional<F16, f16x8, bf16x8>::type frag_t;
    float sc_a = 1.f, sc_b = 1.f, sc_ia = 1.f, sc_ib = 1.f;
    if constexpr (F16) {
        auto scale_of = [](const unsigned *p) {
            if (p == nullptr) return 1.f;
            return split_scale_from_bits(__builtin_amdgcn_readfirstlane(*p));
        };
        sc_a = scale_of(a.amax_dy);
        sc_b = scale_of(a.amax_x);
        sc_ia = split_scale_inv(sc_a);
        sc_ib = split_scale_inv(sc_b);
    }
    constexpr int BLK = 32 * 32 + 32;            // one 32-column block of a K-step: [32 pixels][32 channels] (+64 B: the 8-byte
                                                 // stores of a 16-lane group go to two blocks, on disjoint banks)
    constexpr int IMG = 4 * BLK;                 // 128 channels
    extern __shared__ __align__(16) unsigned short lds16[];
    unsigned short *As = lds16;                  // [2][SP][IMG]  dY  (ko)
    unsigned short *Bs = lds16 + 2 * SP * IMG;   // [2][SP][IMG]  X   (c)

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
        unsigned short *A = As + buf * SP * IMG, *B = Bs + buf * SP * IMG;
        const int blk = (s_col >> 5) * BLK, cc = s_col & 31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u16x4 pa[SP], pb[SP];
            split_bf4<SP, F16>(ra[j], pa, sc_a);
            split_bf4<SP, F16>(rb[j], pb, sc_b);
#pragma unroll
            for (int sp = 0; sp < SP; ++sp) {
                *reinterpret_cast<u16x4 *>(A + sp * IMG + blk + (s_row + 8 * j) * 32 + cc) = pa[sp];
                *reinterpret_cast<u16x4 *>(B + sp * IMG + blk + (s_row + 8 * j) * 32 + cc) = pb[sp];
            }
        }
    };

    f32x16 acc[2][2], acl[SP > 1 ? 2 : 1][SP > 1 ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if constexpr (SP > 1) acl[i][j][e] = 0.f;
            }

    load_all(kc_begin);
    store_all(0);
    load_all(kc_begin + 1);
    __syncthreads();
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int buf = (kc - kc_begin) & 1;
        const unsigned short *A = As + buf * SP * IMG, *B = Bs + buf * SP * IMG;
        // fragments of one 16-pixel half at a time (SP > 1: 32 registers instead of 64 — the split kernel must stay within 256
        // registers for its second workgroup per CU)
        frag_t fa[SP][2], fb[SP][2];
        auto frags = [&](int kk) {
#pragma unroll
            for (int sp = 0; sp < SP; ++sp) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[sp][i] = lds_tr_frag<frag_t>(A + sp * IMG + (wm * 2 + i) * BLK, kk * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[sp][j] = lds_tr_frag<frag_t>(B + sp * IMG + (wn * 2 + j) * BLK, kk * 16, lane);
            }
        };
        auto mma = [&]() {
#define RR_MM(ACC, PA, PB)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                        \
        ACC[i][j] = mfma16(fa[PA][i], fb[PB][j], ACC[i][j]);
            RR_MM(acc, 0, 0)
            if constexpr (SP >= 2) { RR_MM(acl, 0, 1) RR_MM(acl, 1, 0) }
            if constexpr (SP == 3) { RR_MM(acl, 1, 1) RR_MM(acl, 0, 2) RR_MM(acl, 2, 0) }
#undef RR_MM
        };
        if constexpr (SP == 1) {
            frag_t ga[2], gb[2];       // (one part: both halves up front, as the kernel was tuned)
#pragma unroll
            for (int i = 0; i < 2; ++i) ga[i] = lds_tr_frag<frag_t>(A + (wm * 2 + i) * BLK, 16, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) gb[j] = lds_tr_frag<frag_t>(B + (wn * 2 + j) * BLK, 16, lane);
            frags(0);
            store_all(buf ^ 1);
            load_all(kc + 2);
            mma();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(ga[i], gb[j], acc[i][j]);
        } else {
            frags(0);
            store_all(buf ^ 1);
            load_all(kc + 2);
            mma();
            frags(1);
            mma();
        }
        __syncthreads();
    }
    if constexpr (SP > 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] += acl[i][j];
                if constexpr (F16) acc[i][j] = acc[i][j] * sc_ia * sc_ib;
            }
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

__global__ __launch_bounds__(256) void absmax_bits_kernel(const f32x4 *x, long n4, unsigned *out)
{
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = x[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // non-negative floats order as their bit patterns; look before the same-address atomic (see bn_apply_kernel)
    if ((threadIdx.x & 63) == 0 && __builtin_bit_cast(unsigned, m) > *(volatile unsigned *)out) atomicMax(out, __builtin_bit_cast(unsigned, m));
}

}  // namespace

extern "C" int rr_absmax_bits(const float *x, long n, unsigned *out, hipStream_t stream)
{
    RR_CHECK_ARG(n % 4 == 0, "rr_absmax_bits: n must be a multiple of 4");
    if (n == 0) return RR_OK;
    long blocks = (n / 4 + 256 * 8 - 1) / (256 * 8);       // >= 8 float4 per thread; a filter takes a few workgroups, not 1024
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3((int)blocks), dim3(256), 0, stream, reinterpret_cast<const f32x4 *>(x), n / 4, out);
    RR_CHECK_LAUNCH("rr_absmax_bits");
    return RR_OK;
}

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

static int dgrad_s2_impl(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k, int r, int s, int pad_h,
                         int pad_w, int accumulate, float *wsub, hipStream_t stream, int split, const unsigned *amax_dy,
                         const unsigned *amax_w)
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
    if (any_empty && !accumulate) RR_CHECK_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)n * h * wd * c, stream), "rr_conv_dgrad_s2_bf16");
    long base = 0;
    for (int cl = 0; cl < 4; ++cl) {
        const int ph = cl >> 1, pw = cl & 1;
        const int Hc = (h - ph + 1) / 2, Wc = (wd - pw + 1) / 2;
        const long blk = (long)c * Rc[cl] * Sc[cl] * k;
        if (blk > 0 && Hc > 0 && Wc > 0) {
            const OutMap om{Hc, Wc, h, wd, 2, 2, ph, pw};
            const int rc = fprop_impl(dy, wsub + base, nullptr, dx, nullptr, n, p, q, k, c, Rc[cl], Sc[cl], 1, lead_h[cl], lead_w[cl], 0,
                                      accumulate, stream, nullptr, &om, nullptr, split, amax_dy, amax_w);
            if (rc != RR_OK) return rc;
        }
        base += blk;
    }
    return RR_OK;
}

extern "C" int rr_conv_dgrad_s2_bf16(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k,
                                     int r, int s, int pad_h, int pad_w, int accumulate, float *wsub, hipStream_t stream)
{
    return dgrad_s2_impl(dy, w, dx, n, h, wd, c, k, r, s, pad_h, pad_w, accumulate, wsub, stream, 0, nullptr, nullptr);
}

static int wgrad_impl(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k, int r, int s, int stride,
                      int pad_h, int pad_w, int out_h, int out_w, hipStream_t stream, int split, const unsigned *amax_x,
                      const unsigned *amax_dy)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && k > 0 && r > 0 && s > 0 && stride > 0, "rr_conv_wgrad_bf16: bad dims");
    RR_CHECK_ARG(c % 4 == 0 && k % 4 == 0, "rr_conv_wgrad_bf16: C=%d, K=%d must be multiples of 4 (fp32 path for the rest)", c, k);
    WgradArgs a{};
    a.x = x; a.dy = dy; a.dw = dw;
    a.amax_x = amax_x; a.amax_dy = amax_dy;
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
    // pixel splits: fill the resident-workgroup slots (256 CUs x 4; x 2 for the split kernel's 68 KB of LDS) once, never fewer
    // than 16 K-steps per split
    const int slots = split ? 512 : 1024;
    int splits = tiles < slots ? slots / tiles : 1;
    if (splits > rr_cdiv(total_chunks, 16)) splits = rr_cdiv(total_chunks, 16);
    if (splits < 1) splits = 1;
    a.chunks_per_split = rr_cdiv(total_chunks, splits);
    splits = rr_cdiv(total_chunks, a.chunks_per_split);
    const size_t lds = sizeof(unsigned short) * 4 * 4 * (32 * 32 + 32) * (split ? 2 : 1);  // 2 operands x 2 buffers x (parts) x 4 blocks of 32 x 32 (+ pad)
    if (split) {
        auto kern = conv_wgrad_bf16_kernel<2, true>;
        RR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "rr_conv_wgrad_f16x3");
        hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(256), lds, stream, a);
    } else {
        hipLaunchKernelGGL((conv_wgrad_bf16_kernel<1, false>), dim3(tiles * splits), dim3(256), lds, stream, a);
    }
    RR_CHECK_LAUNCH("rr_conv_wgrad_bf16");
    return RR_OK;
}

extern "C" int rr_conv_wgrad_bf16(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k,
                                  int r, int s, int stride, int pad_h, int pad_w, int out_h, int out_w, hipStream_t stream)
{
    return wgrad_impl(x, dy, dw, n, h, wd, c, k, r, s, stride, pad_h, pad_w, out_h, out_w, stream, 0, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------
// Split-operand entry points ("f16x3"): the same convolutions with each fp32 operand written as hi + lo, two fp16 values
// (22 significant bits), after a power-of-two scaling that puts the tensor's largest magnitude into [2^14, 2^15), and
// hi*hi + hi*lo + lo*hi accumulated in fp32 on v_mfma_f32_32x32x16_f16 — three matrix instructions at the 16-bit rate
// instead of eight v_mfma_f32_32x32x2_f32 for the same 32 x 32 x 16 block (the fp32 matrix rate of gfx950 is 1/16 of the
// 16-bit one).  amax_*: device words holding the bit pattern of max|tensor| (rr_absmax_bits), read by the kernels, no host
// synchronisation.  Error against an fp64 convolution: below the fp32-MFMA kernels' own (tests/test_conv_split_gpu.py).
extern "C" int rr_weight_split_f16(const float *w, long n, const unsigned *amax_w, unsigned short *out, hipStream_t stream)
{
    RR_CHECK_ARG(w && amax_w && out && n > 0 && n % 4 == 0, "rr_weight_split_f16: n must be a positive multiple of 4, pointers non-null");
    const long n4 = n / 4;
    long sb = (n4 + 255) / 256;
    if (sb > 1024) sb = 1024;
    hipLaunchKernelGGL(weight_split_f16_kernel, dim3((int)sb), dim3(256), 0, stream, reinterpret_cast<const f32x4 *>(w), amax_w, out, n4);
    RR_CHECK_LAUNCH("rr_weight_split_f16");
    return RR_OK;
}

extern "C" int rr_conv_fprop_f16x3(const float *x, const float *w, const float *bias, float *y, double *stat_slab,
                                   int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h,
                                   int pad_w, int relu, const unsigned *amax_x, const unsigned *amax_w,
                                   const unsigned short *w_split, hipStream_t stream)
{
    RR_CHECK_ARG(amax_x && amax_w, "rr_conv_fprop_f16x3: the operands' maxima are required");
    return fprop_impl(x, w, bias, y, stat_slab, n, h, wd, c, k, r, s, stride, pad_h, pad_w, relu, 0, stream, nullptr, nullptr, nullptr,
                      1, amax_x, amax_w, w_split);
}

extern "C" int rr_conv_dgrad_s1_f16x3(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                      int r, int s, int pad_h, int pad_w, int accumulate, const unsigned *amax_dy,
                                      const unsigned *amax_w, const unsigned short *wt_split, hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1_f16x3: pad must be in [0, kernel)");
    RR_CHECK_ARG(amax_dy && amax_w, "rr_conv_dgrad_s1_f16x3: the operands' maxima are required");
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1_f16x3: empty dy");
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate, stream,
                      nullptr, nullptr, nullptr, 1, amax_dy, amax_w, wt_split);
}

extern "C" int rr_conv_dgrad_s1_bnsum_f16x3(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                            int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_y,
                                            const float *prod_z, const float *prod_mean, const float *prod_invstd,
                                            const float *prod_mask_scale, const float *prod_mask_shift, double *slab,
                                            double *sums, const unsigned *amax_dy, const unsigned *amax_w,
                                            const unsigned short *wt_split, hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1_bnsum_f16x3: pad must be in [0, kernel)");
    RR_CHECK_ARG(prod_y && prod_mean && prod_invstd && slab && sums && (!prod_mask_scale == !prod_mask_shift) && amax_dy && amax_w,
                 "rr_conv_dgrad_s1_bnsum_f16x3: the producer's y / mean / invstd, the two buffers and the maxima are required");
    RR_CHECK_ARG(c % 4 == 0 && c <= 1024, "rr_conv_dgrad_s1_bnsum_f16x3: C=%d must be a multiple of 4 and <= 1024", c);
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1_bnsum_f16x3: empty dy");
    const BnSumArgs bs{prod_y, prod_z, prod_mean, prod_invstd, prod_mask_scale, prod_mask_shift, slab, sums, 0};
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate, stream, &bs,
                      nullptr, nullptr, 1, amax_dy, amax_w, wt_split);
}

extern "C" int rr_conv_dgrad_s1_relubias_f16x3(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                               int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_z,
                                               double *slab, double *sums, const unsigned *amax_dy, const unsigned *amax_w,
                                               hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1_relubias_f16x3: pad must be in [0, kernel)");
    RR_CHECK_ARG(prod_z && slab && sums && amax_dy && amax_w, "rr_conv_dgrad_s1_relubias_f16x3: the producer's output, the two buffers and the maxima are required");
    RR_CHECK_ARG(c % 4 == 0 && k % 4 == 0 && c <= 1024, "rr_conv_dgrad_s1_relubias_f16x3: C=%d, K=%d must be multiples of 4", c, k);
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1_relubias_f16x3: empty dy");
    const BnSumArgs bs{prod_z, prod_z, nullptr, nullptr, nullptr, nullptr, slab, sums, 1};
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate, stream, &bs,
                      nullptr, nullptr, 1, amax_dy, amax_w);
}

extern "C" int rr_conv_dgrad_s2_f16x3(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k,
                                      int r, int s, int pad_h, int pad_w, int accumulate, float *wsub, const unsigned *amax_dy,
                                      const unsigned *amax_w, hipStream_t stream)
{
    RR_CHECK_ARG(amax_dy && amax_w, "rr_conv_dgrad_s2_f16x3: the operands' maxima are required");
    return dgrad_s2_impl(dy, w, dx, n, h, wd, c, k, r, s, pad_h, pad_w, accumulate, wsub, stream, 1, amax_dy, amax_w);
}

extern "C" int rr_conv_wgrad_f16x3(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k,
                                   int r, int s, int stride, int pad_h, int pad_w, int out_h, int out_w, const unsigned *amax_x,
                                   const unsigned *amax_dy, hipStream_t stream)
{
    RR_CHECK_ARG(amax_x && amax_dy, "rr_conv_wgrad_f16x3: the operands' maxima are required");
    return wgrad_impl(x, dy, dw, n, h, wd, c, k, r, s, stride, pad_h, pad_w, out_h, out_w, stream, 1, amax_x, amax_dy);
}
