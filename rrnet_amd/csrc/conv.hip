// im2col-free NHWC fp32 convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the hourglass / head / stage-2 stack of the reference, what it delegates to
// cuDNN through nn.Conv2d: /root/reference/backbones/hourglass.py:17,21,25,48,143,167,173,
// /root/reference/detectors/centernet_detector.py:62,73,85,
// /root/reference/detectors/fasterrcnn_detector.py:11 + backbones/resnet.py:22-28.
//
// One implicit-GEMM kernel family, no column matrix is ever materialised:
//   fprop : Y[m,ko]  = sum_{tap,c}  X[pix(m,tap), c]   * W[ko,tap,c]      M = N*P*Q, N = K,  Kg = R*S*C
//   dgrad : dX[m,c]  = sum_{tap,ko} dY[pix'(m,tap),ko] * W[ko,tap,c]      M = N*H*W, N = C,  Kg = R*S*K
//   wgrad : dW[ko,tap,c] += sum_m   dY[m,ko] * X[pix(m,tap), c]           M = K, N = C, Kg = N*P*Q (split)
// Activations NHWC, weights OHWI ([K][R][S][C], torch channels_last of the reference's
// parameter).  Workgroup = 256 threads = 4 waves; block tile 128 x BN (BN = 128: 2x2 waves of
// 64x64; BN = 32: 4x1 waves of 32x32) x 32 deep; operands are staged global -> registers ->
// LDS (double buffered, one barrier per K-step) with the pixel gather / zero padding done on
// the way in; every wave runs TM*TN accumulators of 32x32 (16 VGPRs each).  The [m][k] LDS
// images are padded to 36 floats per row so that one ds_read_b128 per lane feeds four
// MFMA K-steps conflict-free; [k][n] images are read with ds_read_b32.
// K-chunks run channel-chunk outer / filter-tap inner so the 9 taps of a 3x3 re-read the same
// 128-B lines from L2, and block ids are remapped so that an XCD owns a contiguous tile range.
//
// Roofline: MFMA-bound (fp32 matrix peak 157.3 TFLOP/s); algorithmic FLOPs = 2*M*N*Kg.
#include "common.h"
#include "rrnet_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16 zero bytes in global memory: masked lanes of the gathers load from here, so padding needs no
// select on the loaded value (a select right behind the load would pin an s_waitcnt vmcnt in the stream)
__device__ float rr_zero16[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDA = 36;  // [m][k] image row stride (floats): conflict-free ds_read_b128

struct ConvArgs {
    const float *src;  // fprop: X [N,H,W,C]      dgrad: dY [N,P,Q,K]
    const float *w;    // [K][R][S][C]
    float *dst;        // fprop: Y [N,P,Q,K]      dgrad: dX [N,H,W,C]
    const float *bias; // fprop only, [K] or null
    double *stat_slab; // fprop only: [mtiles][2][K] per-block column sums / sums of squares, or null
    // BatchNorm-backward sums of the layer that PRODUCED the tensor whose gradient this launch writes (stride-1 data
    // gradient through the forward kernel): with bs_y set the slab receives, per block and column c, sum(d) and
    // sum(d * xhat) with d = dst * relu-mask (mask from bs_z > 0, or recomputed as bs_y*bs_msc+bs_msh > 0) and
    // xhat = (bs_y - bs_mean) * bs_invstd — what rr_bn_bwd_reduce would compute in a separate pass over dst and y.
    const float *bs_y, *bs_z, *bs_mean, *bs_invstd, *bs_msc, *bs_msh;
    // bs_relu_bias != 0: the producer is conv + bias + ReLU (a head's 3x3 layer, centernet_detector.py:62): the epilogue
    // stores the MASKED gradient dst*(bs_z > 0) and the slab's first row holds its column sums = the bias gradient
    // (what rr_bias_relu_bwd computes in a pass of its own); bs_mean / bs_invstd are not read.
    int bs_relu_bias;
    int N;
    int SH, SW, SC;    // source spatial / channels
    int DH, DW, DC;    // destination spatial / channels (DC = GEMM N)
    int R, S, stride, pad_h, pad_w;
    int relu, accumulate;
    int M;             // N*DH*DW
    int Kg;            // R*S*SC
    int wK, wC;        // weight dims K, C
    const float *zero; // 16 zero bytes (rr_zero16) for masked gather lanes
    int ksplit;        // > 1: grid.z slices the K loop and the epilogue adds with float atomics (dst pre-zeroed
                       // or holding the running sum); no bias / ReLU / statistics in that mode
    int parity_order;  // dgrad, stride 2: 4 x 2 bits, parity class handled by blockIdx.y = 0..3 (most taps first)
    int parity;        // dgrad, stride 2: blockIdx.y = output parity class (h%2, w%2); only the taps that
                       // can reach that class are visited (1/2/2/4 of a 3x3) instead of masking 3/4 of the MFMAs
    int pos_major;     // fprop on tiny maps (the stage-2 head's 3x3 convolution on 3x3 RoI maps, fasterrcnn_detector.py:18
                       // -> Bottleneck.conv2): an M tile = ONE output pixel of 128 consecutive images, so the taps that
                       // fall into the padding are the same for every row of the tile and are skipped, not masked
                       // (a padded 3x3 on a 3x3 map: 49 of 81 (pixel, tap) pairs are real)
};

__device__ __forceinline__ int xcd_remap(int bid, int nb)
{
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// MODE 0 = fprop, 1 = dgrad.  SCALAR: source/weight channel counts not multiples of 4.
// BKT = K-step depth (32: 2 workgroups per CU by LDS; 16: 4 per CU).
// PIPE: software-pipelined main loop (global loads two K-steps ahead, LDS fragments one 8-deep group
// ahead, barrier placed between the third and fourth MFMA group) so that a wave's MFMA stream never
// waits on a barrier or on LDS latency.
// BNS: the epilogue also emits the producer's BatchNorm-backward sums (ConvArgs::bs_y); a separate instantiation so
// that the plain kernel keeps its register budget (three workgroups per CU).
// POSM: position-major M tiles (ConvArgs::pos_major); like BNS a separate instantiation — folded into the plain kernel
// the extra state cost 20 registers and the third workgroup per CU (457 -> 466 ms per train step).
template <int BN, int MODE, bool SCALAR, int BKT, int PIPE, bool BNS = false, bool POSM = false>
__global__ __launch_bounds__(256, PIPE == 3 ? 3 : 1) void conv_igemm_kernel(const ConvArgs a)
{
    constexpr int WN = BN / 64 ? BN / 64 : 1;   // waves along N
    constexpr int WM = 4 / WN;                  // waves along M
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr bool B_KN = (MODE == 1);          // dgrad reads W as [k][n]
    // [m][k] image row stride: padded by 4 floats for conflict-free ds_read_b128; PIPE 3 (LDS-DMA staging) cannot pad
    // (a wave's 64 x 16 B land contiguously) and XOR-swizzles the 16-byte chunks of a row with (row & 7) instead
    constexpr int LDK = PIPE == 3 ? BKT : BKT + 4;
    constexpr int LDB = B_KN ? BN : LDK;
    constexpr int A_ELEMS = BM * LDK;
    constexpr int B_ELEMS = B_KN ? BKT * BN : BN * LDK;
    constexpr int CPR = BKT / 4;                // float4 columns per [m][k] row
    constexpr int RPP = 256 / CPR;              // rows per pass of the 256 threads
    constexpr int AJ = BM / RPP;                // float4 per thread for the A image
    constexpr int BJ_NK = BN / RPP > 0 ? BN / RPP : 1;
    constexpr int TPR = BN / 4;                 // [k][n] image: threads per k-row
    constexpr int KRPP = 256 / TPR;             // k-rows per pass
    constexpr int BJ_KN = BKT / KRPP > 0 ? BKT / KRPP : 1;
    constexpr int BJ = B_KN ? BJ_KN : BJ_NK;

    extern __shared__ __align__(16) float lds[];
    constexpr int NBUF = PIPE >= 2 ? 1 : 2;   // PIPE 2 / 3: one LDS image per operand, three / four workgroups per CU
    float *As = lds;                    // [NBUF][A_ELEMS]
    float *Bs = lds + NBUF * A_ELEMS;   // [NBUF][B_ELEMS]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = (a.DC + BN - 1) / BN;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tile = logical % ntiles, m_tile = logical / ntiles;
    int m0 = m_tile * BM;
    const int n0 = n_tile * BN;

    const int RS = a.R * a.S;
    // tap sub-lattice visited by this block: all taps, or (parity mode) r = r0, r0+2, ..  s = s0, s0+2, ..
    int r0 = 0, s0 = 0, tstep = 1, Rc = a.R, Sc = a.S;
    int ph = 0, pw = 0, Hc = a.DH, Wc = a.DW, Mloc = a.M;
    if (MODE == 1 && a.parity) {
        // heaviest parity class first (grid.y is the slow dispatch dimension): no long tail of 4-tap blocks
        const int pcls = (a.parity_order >> (2 * blockIdx.y)) & 3;
        ph = pcls >> 1; pw = pcls & 1;
        Hc = (a.DH - ph + 1) / 2; Wc = (a.DW - pw + 1) / 2;
        Mloc = a.N * Hc * Wc;
        if (m0 >= Mloc) return;
        r0 = (ph + a.pad_h) & 1; s0 = (pw + a.pad_w) & 1; tstep = 2;
        Rc = r0 < a.R ? (a.R - r0 + 1) / 2 : 0;
        Sc = s0 < a.S ? (a.S - s0 + 1) / 2 : 0;
    }
    int pm_pix = 0;
    if constexpr (POSM) {
        // the nine tiles of one block of images are neighbours in the launch order (same XCD: the input rows they
        // share come from its L2)
        const int hw = a.DH * a.DW;
        pm_pix = m_tile % hw;
        m0 = (m_tile / hw) * BM;
        Mloc = a.N;
        ph = pm_pix / a.DW; pw = pm_pix - ph * a.DW;
        const int h_lo = ph * a.stride - a.pad_h, w_lo = pw * a.stride - a.pad_w;   // source pixel under tap (0,0)
        r0 = h_lo < 0 ? -h_lo : 0; s0 = w_lo < 0 ? -w_lo : 0;
        const int r1 = a.SH - h_lo < a.R ? a.SH - h_lo : a.R, s1 = a.SW - w_lo < a.S ? a.SW - w_lo : a.S;
        Rc = r1 > r0 ? r1 - r0 : 0; Sc = s1 > s0 ? s1 - s0 : 0;
    }
    const int RSc = Rc * Sc;
    const int cpt = (a.SC + BKT - 1) / BKT;                // channel chunks per tap (vector mode)
    const int nk_all = SCALAR ? (a.Kg + BKT - 1) / BKT : cpt * RSc;
    // split-K (grid.z): small-spatial layers have too few output tiles for 256 CUs
    int kc_lo = 0, kc_hi = nk_all;
    if (a.ksplit > 1) {
        const int per = (nk_all + a.ksplit - 1) / a.ksplit;
        kc_lo = blockIdx.z * per;
        kc_hi = kc_lo + per < nk_all ? kc_lo + per : nk_all;
        if (kc_lo >= kc_hi) return;
    }

    // ---- per-thread A rows: destination pixel coordinates
    // PIPE 3: lane (row, slot) fetches chunk slot ^ (row & 7), so that its DMA lands at the swizzled position
    const int a_col = PIPE == 3 ? (((t % CPR) ^ ((t / CPR) & 7)) * 4) : (t % CPR) * 4;
    const int a_row = t / CPR;
    int a_n[AJ], a_h[AJ], a_w[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int m = m0 + a_row + RPP * j;
        if (POSM && m < Mloc) {
            a_n[j] = m; a_h[j] = ph; a_w[j] = pw;
        } else if (m < Mloc) {
            const int hw = Hc * Wc;
            const int n = m / hw, rem = m - n * hw;
            const int h = rem / Wc;
            a_n[j] = n; a_h[j] = h * tstep + ph; a_w[j] = (rem - h * Wc) * tstep + pw;
        } else {
            a_n[j] = -1; a_h[j] = 0; a_w[j] = 0;
        }
    }

    f32x4 ra[AJ], rb[BJ];

    // ---- vector mode: branch-free gather.  Per row: element offset of the source pixel seen through
    // local tap (0,0) and a bit mask of the taps that stay inside the source image; per K-step only a
    // wave-uniform tap delta is added.  (fprop: ih = h*stride - pad + r; dgrad stride 1: ih = h + pad - r;
    // dgrad parity class: ih = hh + (ph + pad - r0)/2 - ri.)
    const int sgn = MODE == 0 ? 1 : -1;
    long a_base[AJ];
    unsigned long long a_mask[AJ];
    long b_base[BJ];
    bool b_ok[BJ];
    if (!SCALAR) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            int ih0, iw0;
            if (MODE == 0) {
                ih0 = a_h[j] * a.stride - a.pad_h + r0; iw0 = a_w[j] * a.stride - a.pad_w + s0;   // r0 = s0 = 0 unless pos_major
            } else if (a.parity) {
                ih0 = (a_h[j] - ph) / 2 + (ph + a.pad_h - r0) / 2; iw0 = (a_w[j] - pw) / 2 + (pw + a.pad_w - s0) / 2;
            } else {
                ih0 = a_h[j] + a.pad_h; iw0 = a_w[j] + a.pad_w;
            }
            unsigned long long mk = 0ull;
            if (a_n[j] >= 0) {
                for (int ri = 0; ri < Rc; ++ri) {
                    const int ih = ih0 + sgn * ri;
                    if (ih < 0 || ih >= a.SH) continue;
                    for (int si = 0; si < Sc; ++si) {
                        const int iw = iw0 + sgn * si;
                        if (iw >= 0 && iw < a.SW) mk |= 1ull << (ri * Sc + si);
                    }
                }
            }
            a_mask[j] = mk;
            a_base[j] = (((long)(a_n[j] < 0 ? 0 : a_n[j]) * a.SH + ih0) * a.SW + iw0) * a.SC + a_col;
        }
        if (!B_KN) {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int ko = n0 + a_row + RPP * j;
                b_ok[j] = ko < a.wK && (a_row + RPP * j) < BN;
                b_base[j] = (long)ko * RS * a.wC + a_col;
            }
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int c = n0 + (t % TPR) * 4;
                b_ok[j] = c < a.wC && (t / TPR + KRPP * j) < BKT;
                b_base[j] = (long)(t / TPR + KRPP * j) * RS * a.wC + c;
            }
        }
    }

    auto src_offset = [&](int j, int r, int s, long &off) -> bool {
        if (a_n[j] < 0) return false;
        int ih, iw;
        if (MODE == 0) {
            ih = a_h[j] * a.stride - a.pad_h + r;
            iw = a_w[j] * a.stride - a.pad_w + s;
            if (ih < 0 || iw < 0 || ih >= a.SH || iw >= a.SW) return false;
        } else {
            const int th = a_h[j] + a.pad_h - r, tw = a_w[j] + a.pad_w - s;
            if (th < 0 || tw < 0) return false;
            if (a.stride == 1) { ih = th; iw = tw; }
            else {
                ih = th / a.stride; iw = tw / a.stride;
                if (ih * a.stride != th || iw * a.stride != tw) return false;
            }
            if (ih >= a.SH || iw >= a.SW) return false;
        }
        off = (((long)a_n[j] * a.SH + ih) * a.SW + iw) * a.SC;
        return true;
    };

    auto load_tiles = [&](int kc) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (!SCALAR) {
            const int cch = kc / RSc, tl = kc - cch * RSc;
            const int ri = tl / Sc, si = tl - ri * Sc;
            const int tap = (r0 + tstep * ri) * a.S + (s0 + tstep * si);
            const int c0 = cch * BKT;
            const long a_delta = (long)sgn * ((long)ri * a.SW + si) * a.SC + c0;
            const bool c_ok = c0 + a_col < a.SC;
            // A: BM rows x BKT channels of one tap; out-of-image rows read a safe address and are zeroed
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const bool ok = c_ok && ((a_mask[j] >> tl) & 1ull);
                const float *p = ok ? a.src + a_base[j] + a_delta : a.zero;
                ra[j] = *reinterpret_cast<const f32x4 *>(p);
            }
            if (!B_KN) {  // fprop: B[n = ko][k = (tap, c)] = w[ko][tap][c]
                const long w_delta = (long)tap * a.wC + c0;
                const bool wc_ok = c0 + a_col < a.wC;
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const bool ok = b_ok[j] && wc_ok;
                    const float *p = ok ? a.w + b_base[j] + w_delta : a.zero;
                    rb[j] = *reinterpret_cast<const f32x4 *>(p);
                }
            } else {      // dgrad: B[k = (tap, ko)][n = c] = w[ko][tap][c]; c0 indexes the source channels = K
                const long w_delta = ((long)c0 * RS + tap) * a.wC;
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const bool ok = b_ok[j] && (c0 + t / TPR + KRPP * j < a.wK);
                    const float *p = ok ? a.w + b_base[j] + w_delta : a.zero;
                    rb[j] = *reinterpret_cast<const f32x4 *>(p);
                }
            }
        } else {
            // linear k = tap * SC + c over R*S*SC; the 4 columns of this thread decode once per chunk
            int e_tap[4], e_c[4];
            bool e_ok[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kc * BKT + a_col + e;
                e_ok[e] = k < a.Kg;
                e_tap[e] = k / a.SC;
                e_c[e] = k - e_tap[e] * a.SC;
            }
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                f32x4 v = zero;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    long off;
                    const int r = e_tap[e] / a.S, s = e_tap[e] - r * a.S;
                    if (e_ok[e] && src_offset(j, r, s, off)) v[e] = a.src[off + e_c[e]];
                }
                ra[j] = v;
            }
            if (!B_KN) {  // w[ko][k] with k linear — OHWI is already [K][R*S*C]
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const int ko = n0 + a_row + RPP * j;
                    f32x4 v = zero;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = kc * BKT + a_col + e;
                        if (ko < a.wK && (a_row + RPP * j) < BN && k < a.Kg) v[e] = a.w[(long)ko * a.Kg + k];
                    }
                    rb[j] = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const int kl = t / TPR + KRPP * j;
                    const int k = kc * BKT + kl;                 // k = tap * K + ko
                    const int tap = k / a.SC, ko = k - tap * a.SC;
                    const int c = n0 + (t % TPR) * 4;
                    f32x4 v = zero;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kl < BKT && k < a.Kg && c + e < a.wC) v[e] = a.w[((long)ko * RS + tap) * a.wC + c + e];
                    rb[j] = v;
                }
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float *A = As + buf * A_ELEMS;
        float *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            *reinterpret_cast<f32x4 *>(A + (a_row + RPP * j) * LDK + a_col) = ra[j];
        if (!B_KN) {
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                if ((a_row + RPP * j) < BN)
                    *reinterpret_cast<f32x4 *>(B + (a_row + RPP * j) * LDK + a_col) = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                if ((t / TPR + KRPP * j) < BKT)
                    *reinterpret_cast<f32x4 *>(B + (t / TPR + KRPP * j) * LDB + (t % TPR) * 4) = rb[j];
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

    if constexpr (PIPE == 0) {
    if (kc_lo < kc_hi) {
        load_tiles(kc_lo);
        store_tiles(0);
    }
    __syncthreads();
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        const int buf = (kc - kc_lo) & 1;
        if (kc + 1 < kc_hi) load_tiles(kc + 1);
        const float *A = As + buf * A_ELEMS;
        const float *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BKT / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const f32x4 *>(A + ((wm * TM + i) * 32 + lr) * LDK + kk * 8 + lh * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!B_KN) {
                    fb[j] = *reinterpret_cast<const f32x4 *>(B + ((wn * TN + j) * 32 + lr) * LDK + kk * 8 + lh * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fb[j][e] = B[(kk * 8 + lh * 4 + e) * LDB + (wn * TN + j) * 32 + lr];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < kc_hi) store_tiles(buf ^ 1);
        __syncthreads();
    }
    } else {
    static_assert(PIPE == 0 || (BKT == 32 && !SCALAR && (BN == 128 || BN == 64)),
                  "pipelined loop: 4 groups of 8 per K-step, vector gather, 128x128 or 128x64 tile");
    // Each K-step = 4 groups x 4 sub-groups of 4 MFMAs.  Memory instructions are dealt out between the
    // sub-groups (never clustered): a VMEM / DS issue that would stall this wave's in-order stream then
    // overlaps the 64-cycle MFMAs already in the pipe (probe: clustered ds_write+barrier costs 5 %,
    // clustered global loads another 7 % of MFMA throughput).
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto read_frags = [&](int buf, int kk, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
        const float *A = As + buf * A_ELEMS;
        const float *B = Bs + buf * B_ELEMS;
        if constexpr (PIPE == 3) {
            const int slot = ((kk * 2 + lh) ^ (lr & 7)) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const f32x4 *>(A + ((wm * TM + i) * 32 + lr) * LDK + slot);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const f32x4 *>(B + ((wn * TN + j) * 32 + lr) * LDK + slot);
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
            fa[i] = *reinterpret_cast<const f32x4 *>(A + ((wm * TM + i) * 32 + lr) * LDK + kk * 8 + lh * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!B_KN) {
                fb[j] = *reinterpret_cast<const f32x4 *>(B + ((wn * TN + j) * 32 + lr) * LDK + kk * 8 + lh * 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) fb[j][e] = B[(kk * 8 + lh * 4 + e) * LDB + (wn * TN + j) * 32 + lr];
            }
        }
    };
    auto sub = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN], int e) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // Buffer-addressed gathers: 32-bit byte offsets, out-of-range => hardware returns 0 (no zero-page
    // select, no 64-bit pointer arithmetic per load).  The host only takes this path when both tensors are
    // smaller than 2 GiB.
    // descriptor inputs pass through readfirstlane so that hipcc can PROVE the SRD wave-uniform; otherwise it
    // wraps every buffer_load in a waterfall loop (cdna_hip_programming.md T20)
    auto make_srd = [](const float *p, long bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_src = make_srd(a.src, (long)a.N * a.SH * a.SW * a.SC * 4);
    const __amdgpu_buffer_rsrc_t rs_w = make_srd(a.w, (long)a.wK * RS * a.wC * 4);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    int a_boff[AJ], b_boff[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) a_boff[j] = (int)(a_base[j] * 4);
#pragma unroll
    for (int j = 0; j < BJ; ++j) b_boff[j] = (int)(b_base[j] * 4);
    // wave-uniform state of the K-step being fetched, advanced incrementally (tap inner, channel chunk outer)
    int p_cch = kc_lo / RSc, p_tl = kc_lo - (kc_lo / RSc) * RSc;
    int p_ri = p_tl / Sc, p_si = p_tl - (p_tl / Sc) * Sc;
    int p_adelta = 0, p_wdelta = 0, p_c0 = 0, p_tlc = 0;
    bool p_cok = false, p_wcok = false, p_live = true;
    auto prep = [&]() {            // describes step (p_cch, p_ri, p_si), then advances to the next one
        const int tap = (r0 + tstep * p_ri) * a.S + (s0 + tstep * p_si);
        p_c0 = p_cch * BKT;
        p_tlc = p_tl;
        p_adelta = (sgn * (p_ri * a.SW + p_si) * a.SC + p_c0) * 4;
        p_cok = p_c0 + a_col < a.SC;
        p_wdelta = (B_KN ? (p_c0 * RS + tap) * a.wC : tap * a.wC + p_c0) * 4;
        p_wcok = p_c0 + a_col < a.wC;
        ++p_tl;
        if (++p_si == Sc) { p_si = 0; ++p_ri; }
        if (p_tl == RSc) { p_tl = 0; p_ri = 0; p_si = 0; ++p_cch; }
    };
    auto load_a = [&](int j) {
        // bitwise, not &&: a short-circuit on the per-lane p_cok becomes a divergent branch whose arms share
        // destination registers, and hipcc then parks an s_waitcnt vmcnt(0) between consecutive loads
        const unsigned ok = (unsigned)p_cok & (unsigned)((a_mask[j] >> p_tlc) & 1ull) & (unsigned)p_live;
        const unsigned off = ok ? (unsigned)(a_boff[j] + p_adelta) : OOB;
        ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0));
    };
    auto load_b = [&](int j) {
        if (j >= BJ) return;           // 128x64 tile: two weight float4 per thread (j is a literal at every call site)
        const unsigned ok = (B_KN ? ((unsigned)b_ok[j] & (unsigned)(p_c0 + t / TPR + KRPP * j < a.wK))
                                  : ((unsigned)b_ok[j] & (unsigned)p_wcok)) & (unsigned)p_live;
        const unsigned off = ok ? (unsigned)(b_boff[j] + p_wdelta) : OOB;
        rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
    };
    auto store_a = [&](int j, int buf) {
        *reinterpret_cast<f32x4 *>(As + buf * A_ELEMS + (a_row + RPP * j) * LDK + a_col) = ra[j];
    };
    auto store_b = [&](int j, int buf) {
        if (j >= BJ) return;
        if (!B_KN) *reinterpret_cast<f32x4 *>(Bs + buf * B_ELEMS + (a_row + RPP * j) * LDK + a_col) = rb[j];
        else *reinterpret_cast<f32x4 *>(Bs + buf * B_ELEMS + (t / TPR + KRPP * j) * LDB + (t % TPR) * 4) = rb[j];
    };
    static_assert(PIPE == 0 || (AJ == 4 && (BJ == 4 || BJ == 2)), "piece schedule below: 4 + (4 | 2) float4 per thread");
    if constexpr (PIPE == 3) {
    // LDS-DMA staging (buffer_load_dwordx4 ... lds): no staging registers, no ds_write; one unpadded, XOR-swizzled LDS
    // image.  Per K-step: after the last fragment reads of tile kc everybody waits (1), each wave fires its 8 DMA
    // loads of tile kc+1 under group 3's MFMAs, waits for its own (vmcnt 0), everybody waits (2); the exposed part of
    // the load latency is covered by the other workgroups of the CU.
    // Opt-in (RR_CONV_PIPE=3), measured on MI355X: at three workgroups per CU it ties with PIPE 2 (137-139 TFLOP/s on
    // the 256x256 layer); a fourth workgroup needs <= 128 registers per lane and spills 17 dwords whose
    // scratch reloads put s_waitcnt vmcnt(0) between the DMA loads, so PIPE 2 stays the default.
    static_assert(PIPE != 3 || !B_KN, "DMA staging: [n][k] weight image only (fprop / flipped-weight dgrad)");
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma_a = [&](int j) {
        const unsigned ok = (unsigned)p_cok & (unsigned)((a_mask[j] >> p_tlc) & 1ull) & (unsigned)p_live;
        const unsigned off = ok ? (unsigned)(a_boff[j] + p_adelta) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void *)(As + (wave * 8 + RPP * j) * LDK), 16, off, 0, 0, 0);
    };
    auto dma_b = [&](int j) {
        const unsigned ok = ((unsigned)b_ok[j] & (unsigned)p_wcok) & (unsigned)p_live;
        const unsigned off = ok ? (unsigned)(b_boff[j] + p_wdelta) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(Bs + (wave * 8 + RPP * j) * LDK), 16, off, 0, 0, 0);
    };
    if (kc_lo < kc_hi) {
        prep();
#pragma unroll
        for (int j = 0; j < 4; ++j) { dma_a(j); dma_b(j); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kc_lo < kc_hi) read_frags(0, 0, fa0, fb0);
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        p_live = kc + 1 < kc_hi;
        read_frags(0, 1, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        read_frags(0, 2, fa0, fb0);
        sub(fa1, fb1, 0); sub(fa1, fb1, 1); sub(fa1, fb1, 2); sub(fa1, fb1, 3);
        prep();
        read_frags(0, 3, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        dma_a(0); dma_a(1); dma_a(2); dma_a(3);
        sub(fa1, fb1, 0);
        dma_b(0); dma_b(1); dma_b(2); dma_b(3);
        sub(fa1, fb1, 1); sub(fa1, fb1, 2); sub(fa1, fb1, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, 0, fa0, fb0);
    }
    } else {
    if (kc_lo < kc_hi) {
        prep();
#pragma unroll
        for (int j = 0; j < 4; ++j) { load_a(j); load_b(j); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { store_a(j, 0); store_b(j, 0); }
        if (kc_lo + 1 < kc_hi) {      // stays in registers until the first phase's group 1
            prep();
#pragma unroll
            for (int j = 0; j < 4; ++j) { load_a(j); load_b(j); }
        }
    }
    __syncthreads();
    if (kc_lo < kc_hi) read_frags(0, 0, fa0, fb0);
    if constexpr (PIPE == 2) {
    // Single LDS image, two barriers per K-step, both inside group 3: after the last fragment reads of tile kc
    // everybody waits (1), the registers holding tile kc+1 are written over it, everybody waits (2), then the
    // first fragments of kc+1 and the global loads of kc+2 go out — all under group 3's 16 MFMAs.  Half the LDS
    // of the double-buffered loop: a third workgroup fits on the CU and covers the barrier waits.
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        p_live = kc + 2 < kc_hi;
        read_frags(0, 1, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        read_frags(0, 2, fa0, fb0);
        sub(fa1, fb1, 0); sub(fa1, fb1, 1); sub(fa1, fb1, 2); sub(fa1, fb1, 3);
        read_frags(0, 3, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        store_a(0, 0); store_a(1, 0); store_a(2, 0); store_a(3, 0);
        sub(fa1, fb1, 0);
        store_b(0, 0); store_b(1, 0); store_b(2, 0); store_b(3, 0);
        sub(fa1, fb1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, 0, fa0, fb0);
        prep();
        load_a(0); load_a(1); load_a(2); load_a(3);
        sub(fa1, fb1, 2);
        load_b(0); load_b(1); load_b(2); load_b(3);
        sub(fa1, fb1, 3);
    }
    } else {
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        const int buf = (kc - kc_lo) & 1;
        // The body is branch-free on purpose: past the end of the K range the loads turn into out-of-range
        // buffer reads (zeros) and the stores / fragment reads touch an LDS buffer nobody consumes.  With
        // `if (has_next)` arms hipcc's waitcnt pass loses track of which loads the ds_writes already retired
        // and parks s_waitcnt vmcnt(1) between consecutive load pairs, serialising them.
        p_live = kc + 2 < kc_hi;
        // group 0 (fragment set 0); prefetch set 1 <- group 1's fragments
        read_frags(buf, 1, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        // group 1 (set 1); prefetch set 0 <- group 2; stage the next K-step into the other LDS buffer
        read_frags(buf, 2, fa0, fb0);
        store_a(0, buf ^ 1); store_a(1, buf ^ 1);
        sub(fa1, fb1, 0);
        store_a(2, buf ^ 1); store_a(3, buf ^ 1);
        sub(fa1, fb1, 1);
        store_b(0, buf ^ 1); store_b(1, buf ^ 1);
        sub(fa1, fb1, 2);
        store_b(2, buf ^ 1); store_b(3, buf ^ 1);
        sub(fa1, fb1, 3);
        // group 2 (set 0); prefetch set 1 <- group 3; refill the registers two K-steps ahead
        prep();
        read_frags(buf, 3, fa1, fb1);
        load_a(0); load_a(1);
        sub(fa0, fb0, 0);
        load_a(2); load_a(3);
        sub(fa0, fb0, 1);
        load_b(0); load_b(1);
        sub(fa0, fb0, 2);
        load_b(2); load_b(3);
        sub(fa0, fb0, 3);
        // the only barrier of the step: this wave's fragment reads of `buf` have landed and its share of
        // the next tile is written (lgkmcnt(0)); the global loads just issued stay in flight across it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // group 3 (set 1) overlaps the first fragment reads of the next K-step
        read_frags(buf ^ 1, 0, fa0, fb0);
        sub(fa1, fb1, 0); sub(fa1, fb1, 1); sub(fa1, fb1, 2); sub(fa1, fb1, 3);
    }
    }
    }
    }

    // ---- epilogue.  D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    double *sred = reinterpret_cast<double *>(lds);   // [WM][BN][2], reuses the staging LDS
    const bool do_stats = (MODE == 0) && a.stat_slab != nullptr && a.ksplit <= 1;
    constexpr bool bnsum = BNS;                                     // statistics = the producer's BN-backward sums
    // (the host takes the BNS variant only for destinations below 2 GiB; unused and dropped otherwise)
    auto bs_srd = [](const float *p, long bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t bs_rs_y = bs_srd(BNS ? a.bs_y : a.src, (long)a.M * a.DC * 4);
    const __amdgpu_buffer_rsrc_t bs_rs_z = bs_srd(BNS && a.bs_z != nullptr ? a.bs_z : a.src, (long)a.M * a.DC * 4);
    const int mode_e = a.ksplit > 1 ? 2 : (a.accumulate ? 1 : 0);   // wave-uniform: hoisted out of the store loops
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ncol = n0 + (wn * TN + j) * 32 + lr;
        const bool n_ok = ncol < a.DC;
        const float bv = (MODE == 0 && a.bias != nullptr && n_ok) ? a.bias[ncol] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        float bs_m = 0.f, bs_i = 0.f, bs_sc = 0.f, bs_sh = 0.f;
        if constexpr (bnsum) {
            if (n_ok && !a.bs_relu_bias) {
                bs_m = a.bs_mean[ncol]; bs_i = a.bs_invstd[ncol];
                if (a.bs_z == nullptr) {
                    if (a.bs_msc != nullptr) { bs_sc = a.bs_msc[ncol]; bs_sh = a.bs_msh[ncol]; }
                    else bs_sh = 1.f;          // a layer without ReLU: every element counts
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (bnsum) {
                // Two half-tiles of 8 rows: the producer's pre-BN output (and, for layers with a residual, its
                // post-activation output) is fetched for a half-tile before the first dependent use.  One per-lane
                // byte offset (row of this lane half, its column), the row inside the tile goes through the buffer
                // instruction's scalar offset: no 64-bit address per load, and rows past the end of the tensor read 0.
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
                        if (m < Mloc && n_ok) {
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
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int m = m0 + (wm * TM + i) * 32 + row;
                float v = acc[i][j][e] + bv;
                if (MODE == 0 && a.relu) v = v > 0.f ? v : 0.f;
                if (m < Mloc && n_ok) {
                    long pix = m;
                    if constexpr (POSM) pix = (long)m * (a.DH * a.DW) + pm_pix;
                    if (MODE == 1 && a.parity) {
                        const int hw = Hc * Wc;
                        const int n = m / hw, rem = m - n * hw;
                        const int h = rem / Wc;
                        pix = ((long)n * a.DH + (h * 2 + ph)) * a.DW + ((rem - h * Wc) * 2 + pw);
                    }
                    float *p = a.dst + pix * a.DC + ncol;
                    if (mode_e == 2) {
                        unsafeAtomicAdd(p, v);        // partial sums of the K slices meet in a zeroed / running dst
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

// ---------------------------------------------------------------------------------------------
// wgrad: dW[ko][tap][c] += sum over a slice of the N*P*Q pixels.  GEMM M = K (ko), N = C.
struct WgradArgs {
    const float *x;   // [N,H,W,C]
    const float *dy;  // [N,P,Q,K]
    float *dw;        // [K][R][S][C], accumulated with float atomics
    const float *zero; // 16 zero bytes for masked gather lanes
    int N, H, W, C, K, R, S, P, Q, stride, pad_h, pad_w;
    int M;            // N*P*Q
    int chunks_per_split;
    int mt, nt;       // tiles along K and C
};

// BMW x BN output tile (ko x c).  128x128: 2x2 waves of 64x64; 128x32 / 32x128: 4 waves of one 32x32;
// 32x32: the 4 waves split the 32-deep K-step between them (their partial sums meet in the atomics).
// PIPE: 0 plain double-buffered loop; 1 software-pipelined; 2 software-pipelined for Q % BK == 0, where the BK pixels
// of a K-step lie in one output row: the row walk (n, p, q0) is wave-uniform and lives in SGPRs, the tensor offsets go
// through the buffer instruction's scalar offset, and the per-lane work per load is one add, one compare, one select.
// PIPE 3 = PIPE 2 with ONE LDS image per operand and two barriers inside the last MFMA group (three workgroups per CU,
// like conv_igemm_kernel<PIPE 2>).
template <int BMW, int BN, bool A_SCALAR, bool B_SCALAR, int PIPE>
__global__ __launch_bounds__(256, PIPE == 3 ? 3 : 1) void conv_wgrad_kernel(const WgradArgs a)
{
    constexpr int TILES = (BMW / 32) * (BN / 32);
    constexpr int KS = TILES == 1 ? 4 : 1;                       // waves splitting K
    constexpr int WN = TILES == 16 ? 2 : (BN / 32 >= 4 ? 4 : 1);
    constexpr int WM = 4 / (WN * KS);
    constexpr int TM = BMW / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int A_ELEMS = BK * BMW, B_ELEMS = BK * BN;
    constexpr int TPR_A = BMW / 4, RPP_A = 256 / TPR_A, AJ = BMW / 32;
    constexpr int TPR_B = BN / 4, RPP_B = 256 / TPR_B, BJ = BN / 32;

    extern __shared__ __align__(16) float lds[];
    constexpr int NBUF = PIPE == 3 ? 1 : 2;
    float *As = lds, *Bs = lds + NBUF * A_ELEMS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = KS > 1 ? wave : 0;
    const int wm = KS > 1 ? 0 : wave / WN, wn = KS > 1 ? 0 : wave % WN;
    const int RS = a.R * a.S;
    int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tap = logical % RS; logical /= RS;
    const int n_tile = logical % a.nt; logical /= a.nt;
    const int m_tile = logical % a.mt;
    const int split = logical / a.mt;
    const int r = tap / a.S, s = tap - r * a.S;
    const int ko0 = m_tile * BMW, c0 = n_tile * BN;
    const int total_chunks = (a.M + BK - 1) / BK;
    const int kc_begin = split * a.chunks_per_split;
    int kc_end = kc_begin + a.chunks_per_split;
    if (kc_end > total_chunks) kc_end = total_chunks;
    if (kc_begin >= kc_end) return;

    f32x4 ra[AJ], rb[BJ];
    const int a_row = t / TPR_A, a_col = (t % TPR_A) * 4;
    const int b_row = t / TPR_B, b_col = (t % TPR_B) * 4;

    // running (n, p, q) of every B row: advanced by BK pixels per K-step instead of two integer
    // divisions per row per step
    int bn_[BJ], bp_[BJ], bq_[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const long m = (long)kc_begin * BK + b_row + RPP_B * j;
        const int pq = a.P * a.Q;
        bn_[j] = (int)(m / pq);
        const int rem = (int)(m - (long)bn_[j] * pq);
        bp_[j] = rem / a.Q;
        bq_[j] = rem - bp_[j] * a.Q;
    }
    const bool a_ko_ok = ko0 + a_col < a.K;
    const bool b_c_ok = c0 + b_col < a.C;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tiles = [&](int kc) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = kc * BK + a_row + RPP_A * j;
            const int ko = ko0 + a_col;
            if (!A_SCALAR) {
                const bool ok = a_ko_ok && m < a.M;
                const float *p = ok ? a.dy + (long)m * a.K + ko : a.zero;
                ra[j] = *reinterpret_cast<const f32x4 *>(p);
            } else {
                f32x4 v = zero4;
                if (m < a.M) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ko + e < a.K) v[e] = a.dy[(long)m * a.K + ko + e];
                }
                ra[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int c = c0 + b_col;
            const int ih = bp_[j] * a.stride - a.pad_h + r, iw = bq_[j] * a.stride - a.pad_w + s;
            const bool in = bn_[j] < a.N && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
            const long off = (((long)bn_[j] * a.H + ih) * a.W + iw) * a.C + c;
            if (!B_SCALAR) {
                const bool ok = in && b_c_ok;
                const float *p = ok ? a.x + off : a.zero;
                rb[j] = *reinterpret_cast<const f32x4 *>(p);
            } else {
                f32x4 v = zero4;
                if (in) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < a.C) v[e] = a.x[off + e];
                }
                rb[j] = v;
            }
            // advance this row by BK pixels
            bq_[j] += BK;
            while (bq_[j] >= a.Q) {
                bq_[j] -= a.Q;
                if (++bp_[j] == a.P) { bp_[j] = 0; ++bn_[j]; }
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4 *>(A + (a_row + RPP_A * j) * BMW + a_col) = ra[j];
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4 *>(B + (b_row + RPP_B * j) * BN + b_col) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    constexpr bool WPIPE = PIPE != 0 && (BMW == 128 && BN == 128 && !A_SCALAR && !B_SCALAR);
    constexpr bool ALIGNED = PIPE >= 2;
    if constexpr (!WPIPE) {
    load_tiles(kc_begin);
    store_tiles(0);
    __syncthreads();
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int buf = (kc - kc_begin) & 1;
        if (kc + 1 < kc_end) load_tiles(kc + 1);
        const float *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int k2 = 0; k2 < BK / 2 / KS; ++k2) {
            const int kr = 2 * (k2 + wk * (BK / 2 / KS)) + lh;
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
        if (kc + 1 < kc_end) store_tiles(buf ^ 1);
        __syncthreads();
    }
    } else {
    // Software-pipelined form (same scheme as conv_igemm_kernel<PIPE>): 16 sub-steps of 4 MFMAs per
    // K-step; fragments one sub-step ahead in registers, the next tile's ds_writes in sub-steps 1-4, the
    // global loads two K-steps ahead in sub-steps 6-9, the only barrier after sub-step 14.
    // fragments of one GROUP (4 sub-steps = 8 pixels of K) per register set, two sets: the reads of the next
    // group are issued a whole group (16 MFMAs) ahead, as in conv_igemm_kernel<PIPE> — with one sub-step of
    // lead the ds_read_b32 latency under load was still exposed every 4 MFMAs
    float fa0[4][TM], fb0[4][TN], fa1[4][TM], fb1[4][TN];
    auto rdg = [&](int buf, int g, float (&fa)[4][TM], float (&fb)[4][TN]) {
        const float *A = As + buf * A_ELEMS, *B = Bs + buf * B_ELEMS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = 2 * (4 * g + q) + lh;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[q][i] = A[kr * BMW + (wm * TM + i) * 32 + lr];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[q][j] = B[kr * BN + (wn * TN + j) * 32 + lr];
        }
    };
    auto sub = [&](const float (&fa)[4][TM], const float (&fb)[4][TN], int q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][i], fb[q][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto make_srd = [](const float *p, long bytes) {   // provably wave-uniform descriptor (T20)
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_dy = make_srd(a.dy, (long)a.M * a.K * 4);
    // Uniform row walk (ALIGNED): the descriptor of x starts pad_w pixels BEFORE the tensor, so that the per-lane
    // offset (lq*stride + s) * C is never negative.  A negative (wrapped) lane offset is out of range for the
    // hardware's bounds check even when the scalar offset brings the address back inside the tensor — the left-most
    // tap column then silently lost the pixel in front of every K-step that does not start a row (Q > 32).  Lanes
    // whose pixel really lies outside the row are masked by the iw test below and never dereference the shifted base.
    const long x_shift = ALIGNED ? (long)a.pad_w * a.C * 4 : 0;
    const __amdgpu_buffer_rsrc_t rs_x = make_srd(reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.x) - x_shift),
                                                 (long)a.N * a.H * a.W * a.C * 4 + x_shift);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    // A rows: byte offset of row j at K-step 0 of this split is a_off[j]; every K-step adds BK*K*4 (scalar)
    int a_off[AJ], a_m[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        a_m[j] = a_row + RPP_A * j;
        a_off[j] = (a_m[j] * a.K + ko0 + a_col) * 4;
    }
    const int a_step = BK * a.K * 4;
    const int rr_off = r - a.pad_h, ss_off = s - a.pad_w;
    constexpr unsigned FAR = 0x80000000u;   // beyond any (< 2 GiB) tensor, no 32-bit wrap when a scalar offset is added
    // ---- aligned mode state: uniform pixel-row walk + per-lane constants
    int s_n = 0, s_p = 0, s_q0 = 0;
    unsigned a_voff[AJ], b_voff[BJ];
    int b_iw0[BJ];
    if constexpr (ALIGNED) {
        const int m0 = kc_begin * BK, pq = a.P * a.Q;
        s_n = m0 / pq;
        const int rem = m0 - s_n * pq;
        s_p = rem / a.Q;
        s_q0 = rem - s_p * a.Q;
#pragma unroll
        for (int j = 0; j < AJ; ++j) a_voff[j] = a_ko_ok ? (unsigned)a_off[j] : FAR;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int lq = b_row + RPP_B * j;                     // pixel of this row inside the K-step
            b_iw0[j] = lq * a.stride + ss_off;                    // iw = q0*stride + b_iw0
            // relative to the shifted descriptor: (lq*stride + s) pixels, always >= 0
            b_voff[j] = b_c_ok ? (unsigned)(((lq * a.stride + s) * a.C + c0 + b_col) * 4) : FAR;
        }
    }
    auto load_a = [&](int j, int kc) {
        if constexpr (ALIGNED) {
            const int kcl = kc < kc_end ? kc : kc_end - 1;        // past the end: re-read the last chunk (never consumed)
            const int soff = __builtin_amdgcn_readfirstlane(kcl * a_step);
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, a_voff[j], soff, 0));
        } else {
            const unsigned ok = (unsigned)a_ko_ok & (unsigned)(kc * BK + a_m[j] < a.M) & (unsigned)(kc < kc_end);
            const unsigned off = ok ? (unsigned)(a_off[j] + kc * a_step) : OOB;
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off, 0, 0));
        }
    };
    auto load_b = [&](int j) {
        if constexpr (ALIGNED) {
            const int ih = s_p * a.stride + rr_off;
            const bool s_ok = (s_n < a.N) & ((unsigned)ih < (unsigned)a.H);
            // scalar part of the address: pixel (n, ih, q0*stride), never negative
            const int sbase = s_ok ? (((s_n * a.H + ih) * a.W + s_q0 * a.stride) * a.C) * 4 : 0;
            const int iw = s_q0 * a.stride + b_iw0[j];
            const bool ok = s_ok & ((unsigned)iw < (unsigned)a.W);
            const unsigned voff = ok ? b_voff[j] : FAR;
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rs_x, voff, __builtin_amdgcn_readfirstlane(sbase), 0));
        } else {
            const int ih = bp_[j] * a.stride + rr_off, iw = bq_[j] * a.stride + ss_off;
            const unsigned ok = (unsigned)b_c_ok & (unsigned)(bn_[j] < a.N) & (unsigned)((unsigned)ih < (unsigned)a.H) &
                                (unsigned)((unsigned)iw < (unsigned)a.W);
            const unsigned off = ok ? (unsigned)((((bn_[j] * a.H + ih) * a.W + iw) * a.C + c0 + b_col) * 4) : OOB;
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        }
    };
    auto adv_b = [&](int j) {   // advance row j by BK pixels (kept apart from the load to keep VALU clusters short)
        if constexpr (ALIGNED) {
            if (j == BJ - 1) {                                    // one uniform step per K-step
                s_q0 += BK;
                if (s_q0 >= a.Q) { s_q0 = 0; if (++s_p == a.P) { s_p = 0; ++s_n; } }
            }
        } else {
        bq_[j] += BK;
        const bool w1 = bq_[j] >= a.Q;
        bq_[j] -= w1 ? a.Q : 0;
        bp_[j] += w1 ? 1 : 0;
        const bool w2 = bp_[j] >= a.P;
        bp_[j] = w2 ? 0 : bp_[j];
        bn_[j] += w2 ? 1 : 0;
        if (a.Q < BK) {
            while (bq_[j] >= a.Q) {
                bq_[j] -= a.Q;
                if (++bp_[j] == a.P) { bp_[j] = 0; ++bn_[j]; }
            }
        }
        }
    };
    auto st_a = [&](int j, int buf) {
        *reinterpret_cast<f32x4 *>(As + buf * A_ELEMS + (a_row + RPP_A * j) * BMW + a_col) = ra[j];
    };
    auto st_b = [&](int j, int buf) {
        *reinterpret_cast<f32x4 *>(Bs + buf * B_ELEMS + (b_row + RPP_B * j) * BN + b_col) = rb[j];
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) { load_a(j, kc_begin); load_b(j); adv_b(j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { st_a(j, 0); st_b(j, 0); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { load_a(j, kc_begin + 1); load_b(j); adv_b(j); }
    __syncthreads();
    rdg(0, 0, fa0, fb0);
    if constexpr (PIPE == 3) {
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        rdg(0, 1, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        rdg(0, 2, fa0, fb0);
        sub(fa1, fb1, 0); sub(fa1, fb1, 1); sub(fa1, fb1, 2); sub(fa1, fb1, 3);
        rdg(0, 3, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        st_a(0, 0); st_a(1, 0); st_a(2, 0); st_a(3, 0);
        sub(fa1, fb1, 0);
        st_b(0, 0); st_b(1, 0); st_b(2, 0); st_b(3, 0);
        sub(fa1, fb1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rdg(0, 0, fa0, fb0);
        load_a(0, kc + 2); load_a(1, kc + 2); load_a(2, kc + 2); load_a(3, kc + 2);
        sub(fa1, fb1, 2);
        load_b(0); load_b(1); load_b(2); load_b(3);
        adv_b(0); adv_b(1); adv_b(2); adv_b(3);
        sub(fa1, fb1, 3);
    }
    } else {
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int buf = (kc - kc_begin) & 1;
        // branch-free body (see conv_igemm_kernel<PIPE>): past the end of this split's K range the loads are
        // out-of-range buffer reads and the stores / fragment reads touch an LDS buffer nobody consumes
        // group 0
        rdg(buf, 1, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        // group 1: stage the next K-step
        rdg(buf, 2, fa0, fb0);
        st_a(0, buf ^ 1); st_a(1, buf ^ 1);
        sub(fa1, fb1, 0);
        st_a(2, buf ^ 1); st_a(3, buf ^ 1);
        sub(fa1, fb1, 1);
        st_b(0, buf ^ 1); st_b(1, buf ^ 1);
        sub(fa1, fb1, 2);
        st_b(2, buf ^ 1); st_b(3, buf ^ 1);
        sub(fa1, fb1, 3);
        // group 2: refill the registers two K-steps ahead
        rdg(buf, 3, fa1, fb1);
        load_a(0, kc + 2); load_a(1, kc + 2);
        sub(fa0, fb0, 0);
        load_a(2, kc + 2); load_a(3, kc + 2);
        sub(fa0, fb0, 1);
        load_b(0); load_b(1);
        sub(fa0, fb0, 2);
        load_b(2); load_b(3);
        sub(fa0, fb0, 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // group 3 overlaps the first fragment reads of the next K-step and the row bookkeeping
        rdg(buf ^ 1, 0, fa0, fb0);
        adv_b(0); adv_b(1);
        sub(fa1, fb1, 0);
        adv_b(2); adv_b(3);
        sub(fa1, fb1, 1);
        sub(fa1, fb1, 2);
        sub(fa1, fb1, 3);
    }
    }
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
                if (ko < a.K) unsafeAtomicAdd(a.dw + ((long)ko * RS + tap) * a.C + c, acc[i][j][e]);
            }
    }
}

const float *zero_page()
{
    // a __device__ symbol has one instance per device: cache per device id
    static float *p[64] = {};
    int dev = 0;
    hipGetDevice(&dev);
    dev &= 63;
    if (!p[dev]) hipGetSymbolAddress(reinterpret_cast<void **>(&p[dev]), HIP_SYMBOL(rr_zero16));
    return p[dev];
}

template <typename K, typename A>
int launch(K kern, int blocks, size_t lds, hipStream_t stream, const A &args, const char *name, int grid_y = 1,
           int grid_z = 1)
{
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks, grid_y, grid_z), dim3(256), lds, stream, args);
    RR_CHECK_LAUNCH(name);
    return RR_OK;
}

size_t igemm_lds(int bn, bool b_kn, int bk, int nbuf = 2)
{
    return sizeof(float) * nbuf * (BM * (bk + 4) + (b_kn ? bk * bn : bn * (bk + 4)));
}

int conv_rows_kernel() { return 1; }

int conv_pos_major() { return 1; }

int wgrad_aligned() { return 1; }       // 0 generic row walk, 1 uniform row walk + one LDS image (default), 2 uniform + two images

int mid_tiles() { return 48; }       // <= 48 tiles of 128x128 (the 16x16 level): 128x64 tiles; measured worse at 192 (32x32)

int small_tiles() { return 16; }       // <= 16 tiles of 128x128 (the 8x8 level): 128x32 tiles give 4x the workgroups

int conv_bk() { return 32; }

// split-K factor for layers whose output has too few tiles to fill the chip
int pick_ksplit(int blocks, int nk)
{
    static int enabled = -1;
    if (enabled < 0) {
        const char *e = getenv("RR_CONV_SPLITK");
        enabled = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (!enabled || blocks >= 256 || nk < 16) return 1;   // a full first wave of tiles: splitting costs more than it balances
    // Occupancy model: 256 CUs, two workgroups resident per CU.  A pair shares the MFMA pipes at ~0.87 of peak, a
    // lone workgroup reaches ~0.75 (nobody fills its issue gaps); every workgroup pays ~3 K-steps of prologue +
    // epilogue.  The busiest CU holds ceil(total / 256) workgroups; pick the split with the shortest makespan.
    int best = 1;
    float best_t = 0.f;
    for (int ks = 1; ks <= 8 && ks <= nk / 8; ++ks) {
        const int total = blocks * ks;
        const int n = (total + 255) / 256;
        const float per = (float)((nk + ks - 1) / ks) + 3.0f + (ks > 1 ? 1.0f : 0.0f);   // + atomic epilogue
        const float t = (float)(n / 2) * (2.0f * per / 0.87f) + (float)(n % 2) * (per / 0.75f);
        if (best_t == 0.f || t < best_t * 0.95f) { best = ks; best_t = t; }
    }
    return best;
}

// column sums / sums of squares of y -> slab row 0 (the other rows are zeroed by the caller); used when
// split-K keeps the statistics out of the conv epilogue.  Thread = (pixel lane, channel quad), LDS reduce
// over the pixel lanes, one double atomic per channel per workgroup.
__global__ __launch_bounds__(256) void colstats_kernel(const float *y, long M, int C, double *slab)
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

int conv_pipe() { return 2; }       // 0 plain loop, 1 pipelined (two LDS images), 2 pipelined with one LDS image (default)

template <int MODE>
int launch_igemm(ConvArgs &a, int bn, bool scalar, int blocks, int gy, int gz, hipStream_t stream, const char *name)
{
    const int bk = conv_bk();
    const size_t lds = igemm_lds(bn, MODE == 1, bk);
#define IG(BNv, SCv, BKv, PIPEv)                                                                                     \
    launch(conv_igemm_kernel<BNv, MODE, SCv, BKv, PIPEv>, blocks,                                                    \
           PIPEv == 3 ? sizeof(float) * (BM + bn) * bk : (PIPEv == 2 ? igemm_lds(bn, MODE == 1, bk, 1) : lds), stream, a, name, gy, gz)
    if constexpr (MODE == 0) {
        if (a.pos_major) {           // host rule (fprop_impl): vector gather, both tensors below 2 GiB
            if (bn == 128) return launch(conv_igemm_kernel<128, 0, false, 32, 2, false, true>, blocks, igemm_lds(bn, false, 32, 1), stream, a, name, gy, gz);
            return launch(conv_igemm_kernel<64, 0, false, 32, 2, false, true>, blocks, igemm_lds(64, false, 32, 1), stream, a, name, gy, gz);
        }
        if (a.bs_y != nullptr) {     // producer's BatchNorm-backward sums in the epilogue: vector kernels, K-step 32
            const bool small = (long)a.N * a.SH * a.SW * a.SC * 4 < (1l << 31) && (long)a.wK * a.R * a.S * a.wC * 4 < (1l << 31);
            RR_CHECK_ARG(!scalar && bk == 32, "conv: BatchNorm-backward sums need the vector kernels (C %% 4 == 0, RR_CONV_BK=32)");
            if (bn == 128)
                return conv_pipe() >= 1 && small
                           ? launch(conv_igemm_kernel<128, 0, false, 32, 2, true>, blocks, igemm_lds(bn, false, bk, 1), stream, a, name, gy, gz)
                           : launch(conv_igemm_kernel<128, 0, false, 32, 0, true>, blocks, lds, stream, a, name, gy, gz);
            if (bn == 64)
                return conv_pipe() >= 1 && small
                           ? launch(conv_igemm_kernel<64, 0, false, 32, 2, true>, blocks, igemm_lds(bn, false, bk, 1), stream, a, name, gy, gz)
                           : launch(conv_igemm_kernel<64, 0, false, 32, 0, true>, blocks, lds, stream, a, name, gy, gz);
            return launch(conv_igemm_kernel<32, 0, false, 32, 0, true>, blocks, lds, stream, a, name, gy, gz);
        }
    }
    if (bk == 32) {
        // the pipelined kernel addresses both tensors through 32-bit buffer offsets
        const bool small = (long)a.N * a.SH * a.SW * a.SC * 4 < (1l << 31) && (long)a.wK * a.R * a.S * a.wC * 4 < (1l << 31);
        if (bn == 128) {
            if (scalar) return IG(128, true, 32, 0);
            if constexpr (MODE == 0) {
                if (conv_pipe() == 3 && small) return IG(128, false, 32, 3);
            }
            if (conv_pipe() >= 2 && small) return IG(128, false, 32, 2);
            return conv_pipe() && small ? IG(128, false, 32, 1) : IG(128, false, 32, 0);
        }
        if (bn == 64)     // 4 waves along M, 32x64 each: 33..64-column layers and under-filled 384-column levels
            return conv_pipe() >= 2 && small ? IG(64, false, 32, 2) : IG(64, false, 32, 0);
        return scalar ? IG(32, true, 32, 0) : IG(32, false, 32, 0);
    }
    if (bn == 128) return scalar ? IG(128, true, 16, 0) : IG(128, false, 16, 0);
    return scalar ? IG(32, true, 16, 0) : IG(32, false, 16, 0);
#undef IG
}

}  // namespace

extern "C" size_t rr_conv_stat_slab_bytes(int n, int p, int q, int k)
{
    const long M = (long)n * p * q;
    return (size_t)((M + BM - 1) / BM) * 2 * k * sizeof(double);
}

// BatchNorm-backward sums of the producer of the tensor a stride-1 data gradient writes (see ConvArgs::bs_y)
struct BnSumArgs {
    const float *y, *z, *mean, *invstd, *msc, *msh;
    double *slab;      // [ceil(M/128)][2][C] scratch
    double *sums;      // [2][C], zeroed by the caller
    int relu_bias;     // producer = conv + bias + ReLU: masked store, sums[0..C) = bias gradient
};

static int fprop_impl(const float *x, const float *w, const float *bias, float *y, double *stat_slab,
                      int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h,
                      int pad_w, int relu, int accumulate, hipStream_t stream, const BnSumArgs *bs = nullptr)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && k > 0 && r > 0 && s > 0 && stride > 0, "rr_conv_fprop: bad dims");
    ConvArgs a{};
    a.src = x; a.w = w; a.dst = y; a.bias = bias; a.stat_slab = stat_slab;
    a.N = n; a.SH = h; a.SW = wd; a.SC = c;
    a.DH = (h + 2 * pad_h - r) / stride + 1; a.DW = (wd + 2 * pad_w - s) / stride + 1; a.DC = k;
    RR_CHECK_ARG(a.DH > 0 && a.DW > 0, "rr_conv_fprop: empty output");
    a.R = r; a.S = s; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
    a.relu = relu; a.accumulate = accumulate; a.ksplit = 1; a.zero = zero_page();
    const long M = (long)n * a.DH * a.DW;
    RR_CHECK_ARG(M < (1l << 31) && (long)n * h * wd * c < (1l << 40), "rr_conv_fprop: tensor too large");
    a.M = (int)M; a.Kg = r * s * c; a.wK = k; a.wC = c;
    // 1x1 to 49..64 channels (both 32-column tiles of the rows kernel in use) on very many rows without statistics — the
    // stage-2 head's conv1 at inference: the row-streaming kernel of csrc/headtail.hip (weights in registers, no per-tile
    // prologue)
    if (conv_rows_kernel() && r == 1 && s == 1 && stride == 1 && pad_h == 0 && pad_w == 0 && (c == 128 || c == 256) && k > 48 && k <= 64
        && stat_slab == nullptr && bs == nullptr && !accumulate && M >= 64 * 1024 && M * c * 4 < (1l << 31))
        return rr_conv1x1_rows(x, w, bias, y, M, c, k, relu, stream);
    const bool scalar = (c % 4) != 0 || r * s > 64;   // the vector path keeps a 64-bit tap mask per row
    const int bk = conv_bk();
    int bn = k > 64 ? 128 : (k > 32 ? (!scalar && bk == 32 ? 64 : 128) : 32);
    if (bn == 128 && !scalar && rr_cdiv(M, BM) * rr_cdiv(k, 128) <= small_tiles()) bn = 32;   // tiny layers: 4x the tiles
    else if (bn == 128 && !scalar && bk == 32 && rr_cdiv(M, BM) * rr_cdiv(k, 128) <= mid_tiles()) bn = 64;
    int blocks = rr_cdiv(M, BM) * rr_cdiv(k, bn);
    const int nk = scalar ? rr_cdiv(a.Kg, bk) : rr_cdiv(c, bk) * r * s;
    int ks = (bias == nullptr && !relu && k % 4 == 0 && k <= 1024) ? pick_ksplit(blocks, nk) : 1;
    if (bs != nullptr && bs->relu_bias) ks = 1;      // the masked store needs the complete value in one workgroup
    // padded filter on a tiny map, many images (the stage-2 head on 3x3 RoI maps): one output pixel per M tile, padding
    // taps skipped (ConvArgs::pos_major).  Not with statistics in the epilogue (their slab is sized by ceil(M/128) tiles).
    // The variant exists for the pipelined 128- and 64-column kernels (K-step 32, 32-bit buffer offsets).
    if (conv_pos_major() && !scalar && stat_slab == nullptr && bs == nullptr && (pad_h > 0 || pad_w > 0) && r * s > 1
        && a.DH * a.DW <= 16 && n >= 16 * BM && bk == 32 && bn >= 64 && conv_pipe() >= 2
        && (long)n * h * wd * c * 4 < (1l << 31) && (long)k * r * s * c * 4 < (1l << 31)) {
        a.pos_major = 1;
        ks = 1;
        blocks = a.DH * a.DW * rr_cdiv(n, BM) * rr_cdiv(k, bn);
    }
    if (ks > 1) {
        a.ksplit = ks;
        // the split-K destination and (filled by colstats_kernel after the launch) the statistics slab: one zero-fill launch
        const bool zslab = stat_slab != nullptr && bs == nullptr;
        RR_CHECK_HIP(rr_zero2(accumulate ? nullptr : y, accumulate ? 0 : sizeof(float) * (size_t)M * k, zslab ? stat_slab : nullptr,
                              zslab ? rr_conv_stat_slab_bytes(n, a.DH, a.DW, k) : 0, stream), "rr_conv_fprop");
    }
    // The epilogue reads the producer's tensors through buffer descriptors with the row inside the tile in the SCALAR offset,
    // which the hardware's range check does not cover: in a partial last tile it would read up to 11 rows past the end of
    // the tensor.  Fused sums therefore only for M % 128 == 0 (every size the training configurations produce); other
    // sizes take the separate reduce pass below, and the masked-store mode — which has no such fallback — is refused.
    const bool tiles_full = M % BM == 0;
    RR_CHECK_ARG(bs == nullptr || !bs->relu_bias || tiles_full, "rr_conv_dgrad_s1_relubias: N*H*W = %ld must be a multiple of 128", M);
    if (bs != nullptr && ks == 1 && tiles_full) {      // sums in the epilogue; with split-K the complete values exist only afterwards
        a.stat_slab = bs->slab;
        a.bs_y = bs->y; a.bs_z = bs->z; a.bs_mean = bs->mean; a.bs_invstd = bs->invstd; a.bs_msc = bs->msc; a.bs_msh = bs->msh;
        a.bs_relu_bias = bs->relu_bias;
    }
    int rc = launch_igemm<0>(a, bn, scalar, blocks, 1, ks, stream, "rr_conv_fprop");
    if (rc == RR_OK && bs != nullptr) {
        if (ks == 1 && tiles_full) return rr_bn_reduce_slab(bs->slab, (int)rr_cdiv(M, BM), k, bs->sums, stream);
        return rr_bn_bwd_reduce(y, bs->z, bs->y, bs->mean, bs->invstd, bs->msc, bs->msh, bs->sums, M, k, 1, stream);
    }
    if (rc == RR_OK && ks > 1 && stat_slab != nullptr) {
        const int lanes = 256 / (k / 4);
        int sblocks = rr_cdiv(M, (long)lanes * 8);
        if (sblocks > 256) sblocks = 256;
        hipLaunchKernelGGL(colstats_kernel, dim3(sblocks), dim3(256), 0, stream, y, M, k, stat_slab);
        RR_CHECK_LAUNCH("rr_conv_fprop(stats)");
    }
    return rc;
}

extern "C" int rr_conv_fprop(const float *x, const float *w, const float *bias, float *y, double *stat_slab,
                             int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h,
                             int pad_w, int relu, hipStream_t stream)
{
    return fprop_impl(x, w, bias, y, stat_slab, n, h, wd, c, k, r, s, stride, pad_h, pad_w, relu, 0, stream);
}

// wt[c][R-1-r][S-1-s][k] = w[k][r][s][c]: 32x32 tiles of the (k, c) plane through LDS, one tap per grid.z
__global__ __launch_bounds__(256) void weight_flip_transpose_kernel(const float *w, float *wt, int K, int C, int RS)
{
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        tile[i][tx] = (k < K && c < C) ? w[((long)k * RS + tap) * C + c] : 0.f;
    }
    __syncthreads();
    const int ftap = RS - 1 - tap;   // (R-1-r)*S + (S-1-s)
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        if (k < K && c < C) wt[((long)c * RS + ftap) * K + k] = tile[tx][i];
    }
}

extern "C" int rr_weight_flip_transpose(const float *w, float *wt, int k, int c, int r, int s, hipStream_t stream)
{
    RR_CHECK_ARG(k > 0 && c > 0 && r > 0 && s > 0 && r * s < 65536, "rr_weight_flip_transpose: bad dims");
    hipLaunchKernelGGL(weight_flip_transpose_kernel, dim3(rr_cdiv(c, 32), rr_cdiv(k, 32), r * s), dim3(256), 0, stream, w, wt,
                       k, c, r * s);
    RR_CHECK_LAUNCH("rr_weight_flip_transpose");
    return RR_OK;
}

// All filters of a model in one launch: `table` holds one int4 per 32 x 32 tile {element offset of the filter in the flat
// buffers, K, C, packed (tap << 20 | k-tile << 10 | c-tile)} (built once by the host layer, rrnet_amd/flat.py); the
// flipped / transposed copy lands at the SAME offset of `wt_flat`.  Replaces one tiny launch per layer and step on the
// critical path of backward (73 in the headline step, 153 with bf16 stride-1 data gradients at every size).
__global__ __launch_bounds__(256) void weight_flip_transpose_batch_kernel(const float *flat, float *wt_flat, const int4 *table,
                                                                          unsigned short *w16_flat, unsigned short *wt16_flat)
{
    __shared__ float tile[32][33];
    const int4 d = table[blockIdx.x];
    const int K = d.y, C = d.z;
    const int tap = d.w >> 20, k0 = ((d.w >> 10) & 1023) * 32, c0 = (d.w & 1023) * 32;
    const float *w = flat + d.x;
    float *wt = wt_flat + d.x;
    // RS travels in the top bits of K (filters have <= 4096 output channels): K = RS << 16 | K
    const int RS = K >> 16, Kk = K & 0xffff;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        const bool in = k < Kk && c < C;
        const float v = in ? w[((long)k * RS + tap) * C + c] : 0.f;
        tile[i][tx] = v;
        if (in && w16_flat) w16_flat[d.x + ((long)k * RS + tap) * C + c] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
    __syncthreads();
    const int ftap = RS - 1 - tap;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        if (k < Kk && c < C) {
            const float v = tile[tx][i];
            wt[((long)c * RS + ftap) * Kk + k] = v;
            if (wt16_flat) wt16_flat[d.x + ((long)c * RS + ftap) * Kk + k] = __builtin_bit_cast(unsigned short, (__bf16)v);
        }
    }
}

extern "C" int rr_weight_flip_transpose_batch(const float *flat, float *wt_flat, const int *table, int ntiles, hipStream_t stream)
{
    RR_CHECK_ARG(flat && wt_flat && table && ntiles >= 0, "rr_weight_flip_transpose_batch: null argument");
    if (ntiles == 0) return RR_OK;
    hipLaunchKernelGGL(weight_flip_transpose_batch_kernel, dim3(ntiles), dim3(256), 0, stream, flat, wt_flat,
                       reinterpret_cast<const int4 *>(table), (unsigned short *)nullptr, (unsigned short *)nullptr);
    RR_CHECK_LAUNCH("rr_weight_flip_transpose_batch");
    return RR_OK;
}

extern "C" int rr_weight_flip_transpose_batch_bf16(const float *flat, float *wt_flat, unsigned short *w16_flat,
                                                   unsigned short *wt16_flat, const int *table, int ntiles, hipStream_t stream)
{
    RR_CHECK_ARG(flat && wt_flat && table && ntiles >= 0, "rr_weight_flip_transpose_batch_bf16: null argument");
    if (ntiles == 0) return RR_OK;
    hipLaunchKernelGGL(weight_flip_transpose_batch_kernel, dim3(ntiles), dim3(256), 0, stream, flat, wt_flat,
                       reinterpret_cast<const int4 *>(table), w16_flat, wt16_flat);
    RR_CHECK_LAUNCH("rr_weight_flip_transpose_batch_bf16");
    return RR_OK;
}

extern "C" int rr_conv_dgrad_s1(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                int r, int s, int pad_h, int pad_w, int accumulate, hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1: pad must be in [0, kernel)");
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1: empty dy");
    // a stride-1 data gradient IS a forward convolution of dy with the flipped, transposed filter
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate,
                      stream);
}

// out[m][k], m = (n,p,q), k = (r*S + s)*C + c for k < R*S*C and 0 up to KP: the taps of a small-channel convolution laid
// out as one row per output pixel, so that the 7x7 stride-2 stem (C = 3: 147 -> KP = 160) becomes a 1x1 convolution on the
// vector kernels (fprop) and a 128 x KP weight-gradient GEMM over all pixels (wgrad) instead of the scalar-gather paths.
// One thread = one output float4; HBM-bound on the writes (the reads hit L2: every input pixel is wanted R*S/stride^2 times).
__global__ __launch_bounds__(256) void conv_pack_taps_kernel(const float *x, f32x4 *out, int H, int W, int C, int S, int stride,
                                                             int pad_h, int pad_w, int P, int Q, int KP4, int Kg, long total)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k4 = (int)(i % KP4);
        long m = i / KP4;
        const int q = (int)(m % Q); m /= Q;
        const int p = (int)(m % P);
        const long n = m / P;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k4 * 4 + e;
            if (k < Kg) {
                const int tap = k / C, c = k - tap * C;
                const int r = tap / S, sx = tap - r * S;
                const int ih = p * stride - pad_h + r, iw = q * stride - pad_w + sx;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v[e] = x[((n * H + ih) * W + iw) * C + c];
            }
        }
        out[i] = v;
    }
}

extern "C" int rr_conv_pack_taps(const float *x, float *out, int n, int h, int wd, int c, int r, int s, int stride, int pad_h,
                                 int pad_w, int kp, hipStream_t stream)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && r > 0 && s > 0 && stride > 0 && kp % 4 == 0 && kp >= r * s * c,
                 "rr_conv_pack_taps: bad dims (kp must be a multiple of 4 and >= r*s*c)");
    const int p = (h + 2 * pad_h - r) / stride + 1, q = (wd + 2 * pad_w - s) / stride + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_pack_taps: empty output");
    const long total = (long)n * p * q * (kp / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(conv_pack_taps_kernel, dim3((int)blocks), dim3(256), 0, stream, x, reinterpret_cast<f32x4 *>(out), h, wd, c,
                       s, stride, pad_h, pad_w, p, q, kp / 4, r * s * c, total);
    RR_CHECK_LAUNCH("rr_conv_pack_taps");
    return RR_OK;
}

extern "C" int rr_conv_dgrad_s1_bnsum(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                      int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_y,
                                      const float *prod_z, const float *prod_mean, const float *prod_invstd,
                                      const float *prod_mask_scale, const float *prod_mask_shift, double *slab,
                                      double *sums, hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1_bnsum: pad must be in [0, kernel)");
    RR_CHECK_ARG(prod_y && prod_mean && prod_invstd && slab && sums && (!prod_mask_scale == !prod_mask_shift),
                 "rr_conv_dgrad_s1_bnsum: the producer's y / mean / invstd and the two buffers are required");
    RR_CHECK_ARG(c % 4 == 0 && c <= 1024, "rr_conv_dgrad_s1_bnsum: C=%d must be a multiple of 4 and <= 1024", c);
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1_bnsum: empty dy");
    const BnSumArgs bs{prod_y, prod_z, prod_mean, prod_invstd, prod_mask_scale, prod_mask_shift, slab, sums, 0};
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate,
                      stream, &bs);
}

extern "C" int rr_conv_dgrad_s1_relubias(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                         int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_z, double *slab,
                                         double *sums, hipStream_t stream)
{
    RR_CHECK_ARG(pad_h < r && pad_w < s && pad_h >= 0 && pad_w >= 0, "rr_conv_dgrad_s1_relubias: pad must be in [0, kernel)");
    RR_CHECK_ARG(prod_z && slab && sums, "rr_conv_dgrad_s1_relubias: the producer's output and the two buffers are required");
    RR_CHECK_ARG(c % 4 == 0 && k % 4 == 0 && c <= 1024, "rr_conv_dgrad_s1_relubias: C=%d, K=%d must be multiples of 4", c, k);
    const int p = h + 2 * pad_h - r + 1, q = wd + 2 * pad_w - s + 1;
    RR_CHECK_ARG(p > 0 && q > 0, "rr_conv_dgrad_s1_relubias: empty dy");
    const BnSumArgs bs{prod_z, prod_z, nullptr, nullptr, nullptr, nullptr, slab, sums, 1};
    return fprop_impl(dy, wt, nullptr, dx, nullptr, n, p, q, k, c, r, s, 1, r - 1 - pad_h, s - 1 - pad_w, 0, accumulate, stream,
                      &bs);
}

extern "C" int rr_conv_dgrad(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k,
                             int r, int s, int stride, int pad_h, int pad_w, int accumulate,
                             hipStream_t stream)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && k > 0 && r > 0 && s > 0 && stride > 0, "rr_conv_dgrad: bad dims");
    ConvArgs a{};
    a.src = dy; a.w = w; a.dst = dx; a.bias = nullptr; a.stat_slab = nullptr;
    a.N = n;
    a.SH = (h + 2 * pad_h - r) / stride + 1; a.SW = (wd + 2 * pad_w - s) / stride + 1; a.SC = k;
    a.DH = h; a.DW = wd; a.DC = c;
    a.R = r; a.S = s; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
    a.relu = 0; a.accumulate = accumulate; a.ksplit = 1; a.zero = zero_page();
    const long M = (long)n * h * wd;
    RR_CHECK_ARG(M < (1l << 31), "rr_conv_dgrad: tensor too large");
    a.M = (int)M; a.Kg = r * s * k; a.wK = k; a.wC = c;
    const bool scalar = (k % 4) != 0 || (c % 4) != 0 || r * s > 64 || stride > 2;
    const int bk = conv_bk();
    int bn = c > 64 ? 128 : (c > 32 ? (!scalar && bk == 32 ? 64 : 128) : 32);
    if (bn == 128 && !scalar && stride == 1 && rr_cdiv(M, BM) * rr_cdiv(c, 128) <= small_tiles()) bn = 32;
    else if (bn == 128 && !scalar && bk == 32 && stride == 1 && rr_cdiv(M, BM) * rr_cdiv(c, 128) <= mid_tiles()) bn = 64;
    int blocks = rr_cdiv(M, BM) * rr_cdiv(c, bn);
    int gy = 1;
    int nk = scalar ? rr_cdiv(a.Kg, bk) : rr_cdiv(k, bk) * r * s;
    if (stride == 2 && !scalar) {     // parity-decomposed: 4 classes of ceil(h/2) x ceil(w/2) pixels each
        a.parity = 1;
        gy = 4;
        int taps[4], ord[4] = {0, 1, 2, 3};
        for (int cl = 0; cl < 4; ++cl) {
            const int r0 = ((cl >> 1) + pad_h) & 1, s0 = ((cl & 1) + pad_w) & 1;
            taps[cl] = (r0 < r ? (r - r0 + 1) / 2 : 0) * (s0 < s ? (s - s0 + 1) / 2 : 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j)
                if (taps[ord[j]] > taps[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
        a.parity_order = ord[0] | (ord[1] << 2) | (ord[2] << 4) | (ord[3] << 6);
        // classes without any tap (1x1 stride 2: three of four) produce zeros: one memset instead of workgroups
        // that run an empty K loop and store zeros element by element
        // (when accumulating they add nothing: no workgroups at all)
        int live = 4;
        while (live > 1 && taps[ord[live - 1]] == 0) --live;
        if (live < 4) {
            if (!accumulate) RR_CHECK_HIP(rr_zero2(dx, sizeof(float) * (size_t)M * c, nullptr, 0, stream), "rr_conv_dgrad");
            gy = live;
        }
        blocks = rr_cdiv((long)n * ((h + 1) / 2) * ((wd + 1) / 2), BM) * rr_cdiv(c, bn);
        nk = rr_cdiv(k, bk) * ((r + 1) / 2) * ((s + 1) / 2);
    }
    int ks = a.parity ? 1 : pick_ksplit(blocks * gy, nk);
    if (ks > 1) {
        a.ksplit = ks;
        if (!accumulate) RR_CHECK_HIP(rr_zero2(dx, sizeof(float) * (size_t)M * c, nullptr, 0, stream), "rr_conv_dgrad");
    }
    return launch_igemm<1>(a, bn, scalar, blocks, gy, ks, stream, "rr_conv_dgrad");
}

extern "C" int rr_conv_wgrad(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k,
                             int r, int s, int stride, int pad_h, int pad_w, int out_h, int out_w,
                             hipStream_t stream)
{
    RR_CHECK_ARG(n > 0 && h > 0 && wd > 0 && c > 0 && k > 0 && r > 0 && s > 0 && stride > 0, "rr_conv_wgrad: bad dims");
    WgradArgs a{};
    a.x = x; a.dy = dy; a.dw = dw; a.zero = zero_page();
    a.N = n; a.H = h; a.W = wd; a.C = c; a.K = k; a.R = r; a.S = s;
    // out_h/out_w > 0 override the symmetric-padding output size (pad_h/pad_w are the LEADING pads; taps that
    // fall past the far edge are masked), which is how asymmetric padding is expressed
    a.P = out_h > 0 ? out_h : (h + 2 * pad_h - r) / stride + 1;
    a.Q = out_w > 0 ? out_w : (wd + 2 * pad_w - s) / stride + 1;
    a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
    const long M = (long)n * a.P * a.Q;
    RR_CHECK_ARG(M > 0 && M < (1l << 31), "rr_conv_wgrad: bad pixel count");
    a.M = (int)M;
    const int bmw = k > 32 ? 128 : 32;
    const int bn = c > 32 ? 128 : 32;
    a.mt = rr_cdiv(k, bmw); a.nt = rr_cdiv(c, bn);
    const int tiles = a.mt * a.nt * r * s;
    const int total_chunks = rr_cdiv(M, BK);
    // Split the pixel (K) dimension so that tiles*splits fills the 512 resident-workgroup slots
    // (256 CUs x 2) exactly once: equal-length workgroups in more than one round pay a whole extra
    // round for any remainder.  Never fewer than 8 K-steps per split.
    // pipelined 128x128 variants: 32-bit buffer offsets, both tensors below 2 GiB
    const bool as = (k % 4) != 0, bs = (c % 4) != 0;
    const bool pipe_ok = bmw == 128 && bn == 128 && !as && !bs && conv_pipe() && M * k * 4 < (1l << 31) &&
                         (long)n * h * wd * c * 4 < (1l << 31);
    const int wmode = !pipe_ok ? 0 : (a.Q % BK == 0 && wgrad_aligned() ? (wgrad_aligned() == 2 ? 2 : 3) : 1);
    const int slots = wmode == 3 ? 768 : 512;       // resident workgroups: 256 CUs x 3 with one LDS image, else x 2
    int splits = tiles < slots ? slots / tiles : 1;
    if (splits > rr_cdiv(total_chunks, 8)) splits = rr_cdiv(total_chunks, 8);
    if (splits < 1) splits = 1;
    a.chunks_per_split = rr_cdiv(total_chunks, splits);
    splits = rr_cdiv(total_chunks, a.chunks_per_split);
    const int blocks = tiles * splits;
    const size_t lds = sizeof(float) * (wmode == 3 ? 1 : 2) * (BK * bmw + BK * bn);
#define WG(BMv, BNv)                                                                                                  \
    (as ? (bs ? launch(conv_wgrad_kernel<BMv, BNv, true, true, 0>, blocks, lds, stream, a, "rr_conv_wgrad")      \
              : launch(conv_wgrad_kernel<BMv, BNv, true, false, 0>, blocks, lds, stream, a, "rr_conv_wgrad"))    \
        : (bs ? launch(conv_wgrad_kernel<BMv, BNv, false, true, 0>, blocks, lds, stream, a, "rr_conv_wgrad")     \
              : launch(conv_wgrad_kernel<BMv, BNv, false, false, 0>, blocks, lds, stream, a, "rr_conv_wgrad")))
    if (wmode == 3) return launch(conv_wgrad_kernel<128, 128, false, false, 3>, blocks, lds, stream, a, "rr_conv_wgrad");
    if (wmode == 2) return launch(conv_wgrad_kernel<128, 128, false, false, 2>, blocks, lds, stream, a, "rr_conv_wgrad");
    if (wmode == 1) return launch(conv_wgrad_kernel<128, 128, false, false, 1>, blocks, lds, stream, a, "rr_conv_wgrad");
    if (bmw == 128) return bn == 128 ? WG(128, 128) : WG(128, 32);
    return bn == 128 ? WG(32, 128) : WG(32, 32);
#undef WG
}

// shared with csrc/conv_bf16.hip: one occupancy model / tile policy for both precisions
int rr_conv_pick_ksplit(int blocks, int nk) { return pick_ksplit(blocks, nk); }
int rr_conv_small_tiles() { return small_tiles(); }
int rr_conv_mid_tiles() { return mid_tiles(); }
