// Bit-exact wavefront-parallel Soft-NMS for gfx950.
//
// Replaces the reference's serial Cython loop
//   /root/reference/ext/nms/nms/cpu_nms.pyx:17-120 (cpu_soft_nms)
// One workgroup per (image, class) segment; the segment's boxes live in LDS (SoA) for the
// whole run.  Per outer step i (the reference's `for i in range(N)`):
//   1. block-wide arg-max of the scores over [i, N), lowest index wins ties   (pyx:44-52, strict `<`)
//   2. swap rows i <-> maxpos                                                  (pyx:54-66)
//   3. every j in (i, N) decays exactly once, independently                    (pyx:76-104)
//   4. the reference's swap-with-last compaction (pyx:108-115) is a Hoare partition: the
//      k-th dead slot from the left below the new N receives the k-th alive row from the
//      right end.  Done with two block-wide prefix counts over LDS flags.
// Arithmetic mirrors the C that Cython generates (see oracle/soft_nms.c): `+ 1` is a double
// `+ 1.0`, area / iw / ih / ua are rounded to float from double expressions, the gaussian
// weight is (float)exp((double)q).  This file is compiled with -ffp-contract=off.
//
// This kernel is latency-bound, not HBM- or MFMA-bound: algorithmic traffic is
// N*stride*4 bytes in + out per segment.
#include "common.h"
#include "rrnet_hip.h"

namespace {

struct SegView {
    float *x1, *y1, *x2, *y2, *s;
    unsigned short *slot, *src;
    unsigned char *dead;
};

__device__ __forceinline__ float fmax_ref(float a, float b) { return a >= b ? a : b; }
__device__ __forceinline__ float fmin_ref(float a, float b) { return a <= b ? a : b; }

template <int T>
__device__ __forceinline__ int block_sum_i(int v, int *red /* [T/64 + 1] */)
{
    v = wave_sum_i(v);
    if (T == 64) {
        __syncthreads();  // single wave: orders this step's LDS writes before the next step's reads
        return v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) tot += red[w];
    return tot;
}

// exclusive block-wide prefix count of a 0/1 flag; also returns the block total.
template <int T>
__device__ __forceinline__ int block_excl_count(bool flag, int *red, int &total)
{
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    const int wtot = __popcll(m);
    if (T == 64) {
        total = wtot;
        return within;
    }
    __syncthreads();
    if (lane == 0) red[wave] = wtot;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
        const int c = red[w];
        if (w < wave) before += c;
        tot += c;
    }
    total = tot;
    return before + within;
}

template <int T>
__device__ void soft_nms_segment(SegView v, const int n, const float sigma, const float Nt,
                                 const float thr, const int method, int *red, int *out_n, int *err)
{
    const int tid = threadIdx.x;
    int N = n;
    __shared__ float red_s[16];
    __shared__ int red_p[16];
    for (int i = 0; i < n; ++i) {
        if (i >= N) break;  // remaining reference iterations only self-swap dead rows
        // ---- 1. arg-max over [i, N), lowest index among equal maxima
        float bs = -__builtin_huge_valf();
        int bp = 0x7fffffff;
        for (int p = i + tid; p < N; p += T) {
            const float sc = v.s[p];
            if (bp == 0x7fffffff || bs < sc) {  // strided scan visits p in increasing order
                bs = sc;
                bp = p;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int op = __shfl_xor(bp, o, 64);
            const bool take = (op != 0x7fffffff) && (bp == 0x7fffffff || bs < os || (bs == os && op < bp));
            if (take) {
                bs = os;
                bp = op;
            }
        }
        if (T > 64) {
            const int wave = tid >> 6, lane = tid & 63;
            __syncthreads();
            if (lane == 0) {
                red_s[wave] = bs;
                red_p[wave] = bp;
            }
            __syncthreads();
            bs = red_s[0];
            bp = red_p[0];
#pragma unroll
            for (int w = 1; w < T / 64; ++w) {
                const float os = red_s[w];
                const int op = red_p[w];
                const bool take = (op != 0x7fffffff) && (bp == 0x7fffffff || bs < os || (bs == os && op < bp));
                if (take) {
                    bs = os;
                    bp = op;
                }
            }
        }
        const int maxpos = bp;
        // ---- 2. swap i <-> maxpos; every thread keeps the selected box in registers
        const float tx1 = v.x1[maxpos], ty1 = v.y1[maxpos], tx2 = v.x2[maxpos], ty2 = v.y2[maxpos];
        const float ts = v.s[maxpos];
        __syncthreads();
        if (tid == 0 && maxpos != i) {
            v.x1[maxpos] = v.x1[i]; v.y1[maxpos] = v.y1[i]; v.x2[maxpos] = v.x2[i];
            v.y2[maxpos] = v.y2[i]; v.s[maxpos] = v.s[i];
            v.x1[i] = tx1; v.y1[i] = ty1; v.x2[i] = tx2; v.y2[i] = ty2; v.s[i] = ts;
        }
        __syncthreads();
        // ---- 3. decay (i, N)
        int my_dead = 0;
        int my_err = 0;
        for (int p = i + 1 + tid; p < N; p += T) {
            const float x1 = v.x1[p], y1 = v.y1[p], x2 = v.x2[p], y2 = v.y2[p];
            unsigned char dead = 0;
            const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
            const float iw = (float)((double)(fmin_ref(tx2, x2) - fmax_ref(tx1, x1)) + 1.0);
            if (iw > 0.0f) {
                const float ih = (float)((double)(fmin_ref(ty2, y2) - fmax_ref(ty1, y1)) + 1.0);
                if (ih > 0.0f) {
                    const float ua = (float)(((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0)) +
                                              (double)area) - (double)(iw * ih));
                    if (ua == 0.0f) my_err = 1;
                    const float ov = (iw * ih) / ua;
                    float weight;
                    if (method == 1) {
                        weight = (ov > Nt) ? (float)(1.0 - (double)ov) : 1.0f;
                    } else if (method == 2) {
                        const float q = (-(ov * ov)) / sigma;
                        weight = (float)exp((double)q);
                    } else {
                        weight = (ov > Nt) ? 0.0f : 1.0f;
                    }
                    const float ns = weight * v.s[p];
                    v.s[p] = ns;
                    if (ns < thr) dead = 1;
                }
            }
            v.dead[p] = dead;
            my_dead += dead;
        }
        if (my_err) *err = 1;
        const int D = block_sum_i<T>(my_dead, red);
        if (D == 0) continue;  // (block_sum_i's barriers also order the score writes for step 1)
        // ---- 4. compaction == the reference's swap-with-last loop
        __syncthreads();
        const int newN = N - D;
        // 4a. dead slots in [i+1, newN), ascending
        int run = 0;
        for (int base = i + 1; base < newN; base += T) {
            const int p = base + tid;
            const bool f = (p < newN) && v.dead[p];
            int tot;
            const int r = block_excl_count<T>(f, red, tot);
            if (f) v.slot[run + r] = (unsigned short)p;
            run += tot;
        }
        const int nmove = run;
        // 4b. alive rows in [newN, N), descending
        run = 0;
        for (int top = N - 1; top >= newN; top -= T) {
            const int p = top - tid;
            const bool f = (p >= newN) && !v.dead[p];
            int tot;
            const int r = block_excl_count<T>(f, red, tot);
            if (f) v.src[run + r] = (unsigned short)p;
            run += tot;
        }
        __syncthreads();
        for (int k = tid; k < nmove; k += T) {
            const int d = v.slot[k], a = v.src[k];
            v.x1[d] = v.x1[a]; v.y1[d] = v.y1[a]; v.x2[d] = v.x2[a]; v.y2[d] = v.y2[a]; v.s[d] = v.s[a];
        }
        N = newN;
        __syncthreads();
    }
    if (tid == 0) *out_n = N;
}

// ---------------------------------------------------------------------------------------------------------------------
// Register-resident form (round 4).  The LDS-resident loop above spends ~3.5 us per outer step on an arg-max pass over
// LDS, five barriers and a compaction that physically moves rows.  Here every lane OWNS NB boxes for the whole run
// (coordinates, score, precomputed area and the box's current POSITION in the reference's array in registers); nothing
// is moved, only positions change:
//   * swap(i, maxpos)                       -> the two boxes exchange their position numbers (two compares per box);
//   * decay + "is it dead" + the lane's best remaining candidate are ONE pass over the lane's registers;
//   * the block-wide arg-max is a DPP wave reduction of a sortable (score, ~position) key — ties go to the lowest
//     position, cpu_nms.pyx:44-52 — plus, for T > 64, one LDS exchange behind ONE barrier (double-buffered slots);
//     the winner's coordinates travel with its key, so the next step starts right behind that barrier;
//   * the expensive half of the decay (double-precision union, IEEE division, double exp) runs once per OVERLAPPING box
//     of the busiest lane, not once per register slot;
//   * the reference's swap-with-last compaction (pyx:108-115 = a Hoare partition, see above) only renumbers: dead
//     positions are bits of an LDS mask, a surviving box above the new N finds its hole by two popcount scans, and so
//     does — redundantly in every lane — the already selected next box: no second reduction unless the maximal score
//     was tied.  Steps without a death never touch it;
//   * two waves per SIMD (one wave alone issues a vector instruction every 4 cycles, two waves every 2).
// Measured on one box, interleaved (round 4): 1.23 us per outer step at N = 150 and 2.28 at N = 1500 against 2.19 / 3.36
// for the LDS-resident loop (0.94 / 1.58 on steps without deaths); the floor of the bare reduction + exchange chain is
// 0.64 us (tools/softnms_floor.hip).  A third variant (one DPP prefix scan of the mask + an LDS hole table instead of the
// popcount searches) was built, bit-exact, and measured no faster: removed.
// Same arithmetic, same order of selection, same final rows: bit-exact with the LDS form and the reference
// (tests/test_softnms_gpu.py runs every golden through both).  Latency-bound: tools/bench_softnms.py quotes its
// us per outer step against the measured floor of the bare reduction + exchange chain.
__device__ __forceinline__ unsigned sortable_key(float s)
{
    unsigned u = __float_as_uint(s);
    if (u == 0x80000000u) u = 0u;                      // -0.0 == +0.0 in the reference's float compare
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned dpp_max_u32(unsigned v)
{
    // row_shr 1, 2, 4, 8 inside the 16-lane rows, row_bcast 15 / 31 across them: lane 63 ends with the wave maximum.
    // Lanes without a source keep their own value (old = v); max is idempotent, so no bank masks are needed.
    unsigned o;
#define RR_DPP_STEP(ctrl, rmask)                                                  \
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false); \
    v = o > v ? o : v;
    RR_DPP_STEP(0x111, 0xf) RR_DPP_STEP(0x112, 0xf) RR_DPP_STEP(0x114, 0xf) RR_DPP_STEP(0x118, 0xf)
    RR_DPP_STEP(0x142, 0xa) RR_DPP_STEP(0x143, 0xc)
#undef RR_DPP_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

struct RegCand {            // a candidate for the next selection: key = (sortable score, ~position), the box itself
    unsigned ks, kp;
    float x1, y1, x2, y2;
    unsigned tied;          // more than one candidate carries the maximal score (the position decided)
};

// wave-wide best candidate (highest score, then lowest position) -> every lane holds it
__device__ __forceinline__ RegCand wave_best(RegCand c)
{
    const unsigned ms = dpp_max_u32(c.ks);
    unsigned long long tie = __ballot(c.ks == ms);
    int src;
    const unsigned tied = __popcll(tie) > 1;
    if (!tied) {
        src = __ffsll((long long)tie) - 1;
    } else {                                        // equal scores: the lowest position (largest ~position) wins
        const unsigned mp = dpp_max_u32(c.ks == ms ? c.kp : 0u);
        tie = __ballot(c.ks == ms && c.kp == mp);
        src = __ffsll((long long)tie) - 1;
    }
    RegCand r;
    r.ks = ms;
    r.tied = tied;
    r.kp = (unsigned)__builtin_amdgcn_readlane((int)c.kp, src);
    r.x1 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(c.x1), src));
    r.y1 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(c.y1), src));
    r.x2 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(c.x2), src));
    r.y2 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(c.y2), src));
    return r;
}

#ifdef RR_SNMS_STAMP
// development build only (tools/softnms_stamps.sh): cycles of wave 0 per phase of the register kernel's outer step, summed over the launch
__device__ unsigned long long g_snms_stamp[8];
#define SN_DECL unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime()
#define SN_T(i) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); st_acc[i] += t__ - st_last; st_last = t__; } while (0)
#define SN_FLUSH do { if (tid == 0) { for (int q_ = 0; q_ < 7; ++q_) atomicAdd(&g_snms_stamp[q_], st_acc[q_]); atomicAdd(&g_snms_stamp[7], 1ull); } } while (0)
#else
#define SN_DECL
#define SN_T(i)
#define SN_FLUSH
#endif
constexpr int SMALL_SEG_REG = 256;          // one box per lane of four waves
constexpr int REG_DEAD = 0x7fffffff;         // position of a box that is out of the game (dead, or a padding slot)
constexpr int REG_MASK_WORDS = 288;          // 32-bit words of one dead-position mask: positions < 9216
// Round 5: a death step usually kills a handful of boxes.  Their positions go to a short LDS LIST (REG_LIST entries, two buffers)
// and the renumbering works on that list — O(deaths^2) broadcast reads — instead of popcount scans over the position mask, which
// for a 9000-box segment walked up to 288 words per moved box (and per lane for the pre-selected next box): ~6 of the 8.4 us per
// outer step.  Steps with more deaths than the list holds keep the mask path.
constexpr int REG_LIST = 64;
constexpr int REG_LDS_WORDS = 2 * 16 * 8 + 2 * REG_MASK_WORDS + 2 * REG_LIST + 4;

// lds: [2][16] exchange slots of 8 words + [2][REG_MASK_WORDS] dead-position masks (zero on entry)
template <int T, int NB>
__device__ __forceinline__ void soft_nms_registers(float *b, const int stride, const int n, const float sigma, const float Nt, const float thr,
                                   const int method, unsigned *lds, int *out_n, int *err)
{
    constexpr int W = T / 64;
    unsigned *slots = lds;                       // [2][16][8]
    unsigned *masks = lds + 2 * 16 * 8;          // [2][REG_MASK_WORDS] dead positions of the current / the previous death step (mask path)
    int *dlist = reinterpret_cast<int *>(masks + 2 * REG_MASK_WORDS);      // [2][REG_LIST] dead positions of a death step (list path)
    int *dcount = dlist + 2 * REG_LIST;          // [2] entries of each list (zero on entry)
    bool mask_dirty = false;                     // the other mask buffer holds bits of an earlier mask-path step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float x1[NB], y1[NB], x2[NB], y2[NB], sc[NB], ar[NB];
    unsigned key[NB];                            // sortable_key(sc[j]), refreshed when the score changes
    int pos[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int p = j * T + tid;
        if (p < n) {
            const float *r = b + (size_t)p * stride;
            x1[j] = r[0]; y1[j] = r[1]; x2[j] = r[2]; y2[j] = r[3]; sc[j] = r[4];
            pos[j] = p;
            ar[j] = (float)(((double)(x2[j] - x1[j]) + 1.0) * ((double)(y2[j] - y1[j]) + 1.0));
            key[j] = sortable_key(sc[j]);
        } else {
            x1[j] = y1[j] = x2[j] = y2[j] = sc[j] = ar[j] = 0.f;
            pos[j] = REG_DEAD;
            key[j] = 0u;
        }
    }
    int N = n, xbuf = 0, mbuf = 0;

    // block-wide best candidate among the lane candidates `c`; `dead` = deaths of this lane in the current step (summed)
    auto block_best = [&](RegCand c, int dead, int &dead_total) -> RegCand {
        RegCand r = wave_best(c);
        int d = 0;                                 // wave total of the lanes' death counts (0..NB each): NB ballots, no shuffles
#pragma unroll
        for (int q = 1; q <= NB; ++q) d += __popcll(__ballot(dead >= q));
        if (T == 64) {
            dead_total = d;
            return r;
        }
        // slot = [score key, ~position, deaths, tied | x1, y1, x2, y2]: two 16-byte halves.  Every wave's first half is
        // read unconditionally (W independent ds_read_b128: one LDS round trip), the winner's box in a second one.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 *sl = reinterpret_cast<u32x4 *>(slots + (xbuf * 16 + wave) * 8);
        if (lane == 0) {
            sl[0] = u32x4{r.ks, r.kp, (unsigned)d, r.tied};
            sl[1] = u32x4{__float_as_uint(r.x1), __float_as_uint(r.y1), __float_as_uint(r.x2), __float_as_uint(r.y2)};
        }
        __syncthreads();
        const u32x4 *s0 = reinterpret_cast<const u32x4 *>(slots + xbuf * 16 * 8);
        xbuf ^= 1;                                 // the other set of slots next time: no second barrier per step
        u32x4 h[W];
#pragma unroll
        for (int w = 0; w < W; ++w) h[w] = s0[2 * w];
        unsigned bs = h[0][0], bp = h[0][1], tied = h[0][3];
        int bw = 0, tot = (int)h[0][2];
#pragma unroll
        for (int w = 1; w < W; ++w) {
            const unsigned os = h[w][0], op = h[w][1];
            tot += (int)h[w][2];
            const bool same = os == bs && bs != 0u;                   // two waves share the maximal score
            const bool better = os > bs || (os == bs && op > bp);
            tied = same ? 1u : (os > bs ? h[w][3] : tied);
            bs = better ? os : bs;
            bp = better ? op : bp;
            bw = better ? w : bw;
        }
        dead_total = tot;
        const u32x4 box = s0[2 * bw + 1];
        RegCand q;
        q.ks = bs; q.kp = bp; q.tied = tied;
        q.x1 = __uint_as_float(box[0]); q.y1 = __uint_as_float(box[1]);
        q.x2 = __uint_as_float(box[2]); q.y2 = __uint_as_float(box[3]);
        return q;
    };
    // the lane's best candidate among its boxes at positions (lo, N)
    auto lane_best = [&](int lo) -> RegCand {
        RegCand c;
        c.ks = 0u; c.kp = 0u; c.x1 = c.y1 = c.x2 = c.y2 = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (pos[j] > lo && pos[j] < N) {
                const unsigned ks = key[j], kp = ~(unsigned)pos[j];
                if (ks > c.ks || (ks == c.ks && kp > c.kp)) {
                    c.ks = ks; c.kp = kp; c.x1 = x1[j]; c.y1 = y1[j]; c.x2 = x2[j]; c.y2 = y2[j];
                }
            }
        }
        return c;
    };

    int dummy;
    RegCand sel = block_best(lane_best(-1), 0, dummy);
    int my_err = 0;
    SN_DECL;
    for (int i = 0; i < N; ++i) {
        SN_T(6);
        const int maxpos = (int)~sel.kp;
        const float tx1 = sel.x1, ty1 = sel.y1, tx2 = sel.x2, ty2 = sel.y2;
        const double tarea = ((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0);
        // ---- swap i <-> maxpos (position numbers only) and the cheap half of the decay: which of my boxes in (i, N) overlap
        // the selected one at all (iw > 0 and ih > 0)?  Typically a handful of the segment's boxes do.
        unsigned ovl = 0u;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int p = pos[j];
            p = (p == maxpos) ? i : ((p == i) ? maxpos : p);
            pos[j] = p;
            // (float)((double)d + 1.0) == d + 1.0f (one correctly rounded add).  v_min / v_max instead of the reference's
            // `a <= b ? a : b`: they differ only in the sign of a zero result, which the "+ 1" erases (finite inputs)
            const float iw = (__builtin_fminf(tx2, x2[j]) - __builtin_fmaxf(tx1, x1[j])) + 1.0f;
            const float ih = (__builtin_fminf(ty2, y2[j]) - __builtin_fmaxf(ty1, y1[j])) + 1.0f;
            if (p > i && p < N && iw > 0.0f && ih > 0.0f) ovl |= 1u << j;
        }
        // ---- the expensive half (double-precision union, IEEE division, double exp) runs once per OVERLAPPING box of the
        // busiest lane, not once per register slot that holds an overlapping box somewhere in the wave: every lane picks
        // its lowest pending slot, gathers that box with selects (no dynamically indexed registers), updates its score
        SN_T(0);
        unsigned deadbits = 0u;
        while (__any(ovl != 0u)) {
            if (ovl != 0u) {
                const int j = __ffs((int)ovl) - 1;
                ovl &= ovl - 1u;
                float bx1 = x1[0], by1 = y1[0], bx2 = x2[0], by2 = y2[0], bsc = sc[0], bar = ar[0];
#pragma unroll
                for (int jj = 1; jj < NB; ++jj) {
                    const bool t = j == jj;
                    bx1 = t ? x1[jj] : bx1; by1 = t ? y1[jj] : by1; bx2 = t ? x2[jj] : bx2; by2 = t ? y2[jj] : by2;
                    bsc = t ? sc[jj] : bsc; bar = t ? ar[jj] : bar;
                }
                const float iw = (__builtin_fminf(tx2, bx2) - __builtin_fmaxf(tx1, bx1)) + 1.0f;
                const float ih = (__builtin_fminf(ty2, by2) - __builtin_fmaxf(ty1, by1)) + 1.0f;
                const float ua = (float)((tarea + (double)bar) - (double)(iw * ih));
                if (ua == 0.0f) my_err = 1;
                const float ov = (iw * ih) / ua;
                float weight;
                if (method == 1) {
                    weight = (ov > Nt) ? (float)(1.0 - (double)ov) : 1.0f;
                } else if (method == 2) {
                    const float q = (-(ov * ov)) / sigma;
                    weight = (float)exp((double)q);
                } else {
                    weight = (ov > Nt) ? 0.0f : 1.0f;
                }
                const float ns = weight * bsc;
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) {
                    sc[jj] = (j == jj) ? ns : sc[jj];
                    key[jj] = (j == jj) ? sortable_key(ns) : key[jj];
                }
                if (ns < thr) deadbits |= 1u << j;
            }
        }
        // ---- the lane's best surviving candidate for the next step
        SN_T(1);
        RegCand c;
        c.ks = 0u; c.kp = 0u; c.x1 = c.y1 = c.x2 = c.y2 = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int p = pos[j];
            if (p > i && p < N && !(deadbits & (1u << j))) {
                const unsigned ks = key[j], kp = ~(unsigned)p;
                if (ks > c.ks || (ks == c.ks && kp > c.kp)) {
                    c.ks = ks; c.kp = kp; c.x1 = x1[j]; c.y1 = y1[j]; c.x2 = x2[j]; c.y2 = y2[j];
                }
            }
        }
        const int ndead = __popc(deadbits);
        int D;
        SN_T(2);
        sel = block_best(c, ndead, D);
        SN_T(3);
        if (D == 0) continue;
        // ---- renumbering = the reference's swap-with-last compaction: the k-th dead position from the left below the new
        // N receives the k-th surviving box from the right end.
        const int newN = N - D;
        if (D <= REG_LIST) {
            // list path: the step's dead positions, unordered, in LDS
            int *dl = dlist + mbuf * REG_LIST;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (deadbits & (1u << j)) dl[atomicAdd(&dcount[mbuf], 1)] = pos[j];
            __syncthreads();
            auto new_pos = [&](int p) -> int {
                int above = 0;                                    // dead positions in (p, N)
                for (int e = 0; e < D; ++e) above += dl[e] > p ? 1 : 0;
                const int r = (N - 1 - p) - above;                // rank from the right among the survivors of [newN, N)
                for (int e = 0; e < D; ++e) {                     // the hole with exactly r holes to its left
                    const int h = dl[e];
                    if (h >= newN) continue;
                    int below = 0;
                    for (int f = 0; f < D; ++f) below += dl[f] < h ? 1 : 0;
                    if (below == r) return h;
                }
                return p;
            };
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int p = pos[j];
                if (deadbits & (1u << j)) pos[j] = REG_DEAD;
                else if (p >= newN && p < N) pos[j] = new_pos(p);
            }
            if (sel.tied || sel.ks == 0u) {
                N = newN;
                sel = block_best(lane_best(i), 0, dummy);
            } else {
                const int q = (int)~sel.kp;
                if (q >= newN && q < N) sel.kp = ~(unsigned)new_pos(q);
                N = newN;
            }
            // the OTHER list's counter (the previous death step's) is reset now: every wave left that list at least one barrier ago
            mbuf ^= 1;
            if (tid == 0) dcount[mbuf] = 0;
            if (mask_dirty) {                                       // (block-uniform) bits of an earlier mask-path step in that buffer
                for (int w = tid; w < REG_MASK_WORDS; w += T) masks[mbuf * REG_MASK_WORDS + w] = 0u;
                mask_dirty = false;
            }
            continue;
        }
        // mask path (more deaths than the list holds): dead positions of this step -> bits of an LDS mask
        unsigned *mk = masks + mbuf * REG_MASK_WORDS;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (deadbits & (1u << j)) atomicOr(&mk[pos[j] >> 5], 1u << (pos[j] & 31));
        __syncthreads();
        // new position of a survivor that sits at p in [newN, N): its rank r from the right among the survivors of that
        // range is the index, from the left, of the dead position in (i, newN) it moves to
        auto new_pos = [&](int p) -> int {
            int deadabove = 0;
            if (p + 1 < N) {
                const int w0 = (p + 1) >> 5, w1 = (N - 1) >> 5;
                for (int w = w0; w <= w1; ++w) {
                    unsigned m = mk[w];
                    if (w == w0) m &= ~0u << ((p + 1) & 31);
                    if (w == w1 && ((N & 31) != 0)) m &= (1u << (N & 31)) - 1u;
                    deadabove += __popc(m);
                }
            }
            int r = (N - 1 - p) - deadabove;
            const int lo = i + 1;
            const int w0 = lo >> 5, w1 = (newN - 1) >> 5;
            for (int w = w0; w <= w1; ++w) {
                unsigned m = mk[w];
                if (w == w0) m &= ~0u << (lo & 31);
                if (w == w1 && ((newN & 31) != 0)) m &= (1u << (newN & 31)) - 1u;
                const int cnt = __popc(m);
                if (r < cnt) {
                    for (int q = 0; q < r; ++q) m &= m - 1u;
                    return (w << 5) + (__ffs((int)m) - 1);
                }
                r -= cnt;
            }
            return p;
        };
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int p = pos[j];
            if (deadbits & (1u << j)) pos[j] = REG_DEAD;
            else if (p >= newN && p < N) pos[j] = new_pos(p);
        }
        // The next selection was made on the old numbering.  Its score is still the maximum; unless that maximum was
        // shared (a tie is decided by position, and positions just changed: select again, rare) only its own position
        // may have moved — every lane renumbers it for itself, no second reduction, no further barrier.
        if (sel.tied || sel.ks == 0u) {
            N = newN;
            sel = block_best(lane_best(i), 0, dummy);
        } else {
            const int q = (int)~sel.kp;
            if (q >= newN && q < N) sel.kp = ~(unsigned)new_pos(q);
            N = newN;
        }
        // the OTHER buffers (the previous death step's) are cleared now: every wave left them at least one barrier ago; this
        // mask is read until the next barrier and will be cleared by the next death step
        mbuf ^= 1;
        if (tid == 0) dcount[mbuf] = 0;
        for (int w = tid; w < REG_MASK_WORDS; w += T) masks[mbuf * REG_MASK_WORDS + w] = 0u;
        mask_dirty = true;                                          // (the mask just used, cleared by the next death step whichever path it takes)
    }
    SN_FLUSH;
    if (my_err) *err = 1;
    // rows [0, N) in selection order = position order
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (pos[j] < N) {
            float *r = b + (size_t)pos[j] * stride;
            r[0] = x1[j]; r[1] = y1[j]; r[2] = x2[j]; r[3] = y2[j]; r[4] = sc[j];
        }
    }
    if (tid == 0) *out_n = N;
}

constexpr int SMALL_SEG = 192;   // segments up to this many boxes run on one wave
constexpr int MID_SEG = 512;     // first launch of the two-launch split: LDS for this many boxes (12.8 KB)

// LDS-resident: dynamic LDS = n_max * 25 bytes (rounded), boxes are loaded AoS->SoA and stored back.
template <int T>
__global__ __launch_bounds__(T) void soft_nms_kernel(float *boxes, const int *seg_off, const int *seg_len, int stride,
                                                     float sigma, float Nt, float thr, int method,
                                                     int cap, int *n_out, int *err, float *gws, int n_lo, int n_hi)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int red[20];
    const int seg = blockIdx.x;
    const int off = seg_off[seg];
    const int n = seg_len ? seg_len[seg] : seg_off[seg + 1] - off;
    // two-launch split (soft_nms_launch): this launch takes the segments of n_lo < n <= n_hi boxes
    if (n <= n_lo || n > n_hi) return;
    float *b = boxes + (size_t)off * stride;
    SegView v;
    if (gws == nullptr) {
        float *f = reinterpret_cast<float *>(smem);
        v.x1 = f; v.y1 = f + cap; v.x2 = f + 2 * cap; v.y2 = f + 3 * cap; v.s = f + 4 * cap;
        v.slot = reinterpret_cast<unsigned short *>(f + 5 * cap);
        v.src = v.slot + cap;
        v.dead = reinterpret_cast<unsigned char *>(v.src + cap);
    } else {  // segment too large for LDS: same algorithm on a global workspace (25 B per box)
        float *f = gws + (size_t)off * 7;
        v.x1 = f; v.y1 = f + n; v.x2 = f + 2 * n; v.y2 = f + 3 * n; v.s = f + 4 * n;
        v.slot = reinterpret_cast<unsigned short *>(f + 5 * n);
        v.src = v.slot + n;
        v.dead = reinterpret_cast<unsigned char *>(v.src + n);
    }
    for (int p = threadIdx.x; p < n; p += T) {
        const float *r = b + (size_t)p * stride;
        v.x1[p] = r[0]; v.y1[p] = r[1]; v.x2[p] = r[2]; v.y2[p] = r[3]; v.s[p] = r[4];
    }
    __syncthreads();
    __shared__ int outn;
    if (threadIdx.x == 0) outn = n;
    if (T > 64 && n <= SMALL_SEG) {
        // The launch is sized for the LONGEST possible segment (the caller's bound, e.g. K = 1500), the typical
        // (frame, class) segment holds ~150 boxes: those run on the first wave alone — same algorithm, wave
        // shuffles instead of block barriers in every one of its ~N steps (the barriers were most of the step).
        __syncthreads();
        if (threadIdx.x >= 64) return;
        soft_nms_segment<64>(v, n, sigma, Nt, thr, method, red, &outn, err);
        const int nn1 = outn;
        for (int p = threadIdx.x; p < nn1; p += 64) {
            float *r = b + (size_t)p * stride;
            r[0] = v.x1[p]; r[1] = v.y1[p]; r[2] = v.x2[p]; r[3] = v.y2[p]; r[4] = v.s[p];
        }
        if (threadIdx.x == 0) n_out[seg] = nn1;
        return;
    }
    soft_nms_segment<T>(v, n, sigma, Nt, thr, method, red, &outn, err);
    __syncthreads();
    const int nn = outn;
    // only columns 0..4 of the surviving rows are written back; column 5+ never moves (pyx:55-66)
    for (int p = threadIdx.x; p < nn; p += T) {
        float *r = b + (size_t)p * stride;
        r[0] = v.x1[p]; r[1] = v.y1[p]; r[2] = v.x2[p]; r[3] = v.y2[p]; r[4] = v.s[p];
    }
    if (threadIdx.x == 0) n_out[seg] = nn;
}

// One workgroup per segment, boxes in registers (soft_nms_registers).  Segments of up to 192 boxes inside a launch sized for
// longer ones run on the first wave alone (no barriers at all in their ~N steps).
template <int T, int NB>
__global__ __launch_bounds__(T) void soft_nms_reg_kernel(float *boxes, const int *seg_off, const int *seg_len, int stride,
                                                         float sigma, float Nt, float thr, int method, int *n_out, int *err)
{
    __shared__ __align__(16) unsigned lds[REG_LDS_WORDS];
    const int seg = blockIdx.x;
    const int off = seg_off[seg];
    const int n = seg_len ? seg_len[seg] : seg_off[seg + 1] - off;
    for (int w = threadIdx.x; w < REG_LDS_WORDS; w += T) lds[w] = 0u;
    __syncthreads();
    float *b = boxes + (size_t)off * stride;
    if (T > 256 && n <= SMALL_SEG_REG) {         // a short segment in a launch sized for long ones: the first four waves, one box each
        if (threadIdx.x >= 256) return;
        soft_nms_registers<256, 1>(b, stride, n, sigma, Nt, thr, method, lds, &n_out[seg], err);
        return;
    }
    soft_nms_registers<T, NB>(b, stride, n, sigma, Nt, thr, method, lds, &n_out[seg], err);
}

}  // namespace

#ifdef RR_SNMS_STAMP
extern "C" int rr_snms_stamps(unsigned long long *host_out, int reset)
{
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_snms_stamp), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_snms_stamp), z, sizeof(z)) != hipSuccess) return 2;
    }
    return 0;
}
#endif

extern "C" size_t rr_soft_nms_workspace_bytes(int total_boxes, int max_seg_boxes)
{
    // LDS path covers segments up to RR_SOFT_NMS_LDS_MAX boxes; beyond that 28 B per box of global scratch.
    return max_seg_boxes > RR_SOFT_NMS_LDS_MAX ? (size_t)total_boxes * 28 + 64 : 0;
}

static int soft_nms_launch(float *boxes, const int *seg_off, const int *seg_len, int nseg, int max_seg_boxes,
                           int stride, float sigma, float Nt, float threshold, int method,
                           int *n_out, int *err_flag, void *workspace, hipStream_t stream)
{
    RR_CHECK_ARG(stride >= 5, "rr_soft_nms_segments: stride %d < 5", stride);
    RR_CHECK_ARG(nseg >= 0 && max_seg_boxes >= 0, "rr_soft_nms_segments: negative size");
    RR_CHECK_ARG(max_seg_boxes < 65536, "rr_soft_nms_segments: segment of %d boxes (limit 65535)", max_seg_boxes);
    if (nseg == 0) return RR_OK;
    RR_CHECK_ARG(!(method == 2 && sigma == 0.0f), "rr_soft_nms_segments: sigma == 0 (reference raises ZeroDivisionError)");
    const bool in_lds = max_seg_boxes <= RR_SOFT_NMS_LDS_MAX;
    RR_CHECK_ARG(in_lds || workspace != nullptr, "rr_soft_nms_segments: workspace required for %d-box segments", max_seg_boxes);
    const int cap = (max_seg_boxes + 3) & ~3;
    const size_t lds = in_lds ? (size_t)cap * 25 + 16 : 0;
    float *gws = in_lds ? nullptr : reinterpret_cast<float *>(workspace);
#define LAUNCH(T, LDS, CAP, LO, HI)                                                                    \
    do {                                                                                              \
        if ((LDS) > 48 * 1024)                                                                        \
            hipFuncSetAttribute(reinterpret_cast<const void *>(soft_nms_kernel<T>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));              \
        hipLaunchKernelGGL(soft_nms_kernel<T>, dim3(nseg), dim3(T), (LDS), stream, boxes, seg_off,    \
                           seg_len, stride, sigma, Nt, threshold, method, (CAP), n_out, err_flag, gws, (LO), (HI)); \
    } while (0)
    constexpr int ALL = 0x7fffffff;
    // Regime: a few segments (an image's classes at evaluation, rrnet_operator.py:246-284) are a latency problem — the
    // register-resident kernel; a batch of more long segments than two per CU is a throughput problem, where the LDS loop's
    // 256-thread workgroups (4 per CU) retire more segments per unit time (1280 x 1500 boxes: 5.5 against 6.0 ms)
    const bool throughput_regime = nseg > 512 && max_seg_boxes > 256;
    // (measured, one segment: 2500 boxes 4.97 against 5.57 ms for the LDS loop; at 5000 / 9000 boxes — six / nine boxes per
    // lane of a 1024-thread workgroup, 128 registers per lane — the register kernel spills and loses, 15.7 against 14.2 and
    // 52 against 38 ms: it takes segments of up to 3072 boxes)
    // (round 5, with the list renumbering: one 9000-box segment 29 ms in registers — 18 boxes per lane of 512 threads, 256 registers,
    // no scratch — against 38 ms for the LDS loop on its global workspace; a batch of such segments likewise: the LDS loop has no
    // LDS for them and runs out of L2)
    if (max_seg_boxes <= 9216 && (!throughput_regime || !in_lds)) {
        // register-resident kernels: T x NB boxes per segment
#define LAUNCH_REG(T, NB)                                                                                              \
        hipLaunchKernelGGL((soft_nms_reg_kernel<T, NB>), dim3(nseg), dim3(T), 0, stream, boxes, seg_off, seg_len, stride, sigma, \
                           Nt, threshold, method, n_out, err_flag)
        // One wave alone on a SIMD issues a vector instruction every 4 cycles, two waves one every 2: the configurations put
        // (at least) two waves on every SIMD of the CU wherever the segment is long enough to feed them
        // (measured: 1024 x 2 for 1500 boxes, four waves per SIMD, is 30 % slower than 512 x 3: a 16-wave barrier and exchange)
        if (max_seg_boxes <= 256) LAUNCH_REG(256, 1);
        else if (max_seg_boxes <= 1536) LAUNCH_REG(512, 3);
        else if (max_seg_boxes <= 3072) LAUNCH_REG(1024, 3);
        else if (max_seg_boxes <= 6144) LAUNCH_REG(512, 12);
        else LAUNCH_REG(512, 18);
#undef LAUNCH_REG
        RR_CHECK_LAUNCH("rr_soft_nms_segments");
        return RR_OK;
    }
    if (max_seg_boxes <= SMALL_SEG) LAUNCH(64, lds, cap, -1, ALL);
    else if (in_lds && max_seg_boxes > MID_SEG) {
        // The caller's bound is the LONGEST possible segment (K = 1500 at inference: 37.5 KB of LDS per workgroup = four
        // segments per CU) while a typical (frame, class) segment holds ~150 boxes.  Two launches: the segments of up to
        // MID_SEG boxes with 12.8 KB of LDS each (all 1280 segments of a 128-frame batch are resident at once: the launch
        // lasts as long as its longest segment instead of a round and a quarter), then the larger ones with the LDS
        // they need (workgroups of the others return at once).
        LAUNCH(256, (size_t)MID_SEG * 25 + 16, MID_SEG, -1, MID_SEG);
        if (max_seg_boxes <= 2560) LAUNCH(256, lds, cap, MID_SEG, ALL);
        else LAUNCH(1024, lds, cap, MID_SEG, ALL);
    } else if (max_seg_boxes <= 2560) LAUNCH(256, lds, cap, -1, ALL);
    else LAUNCH(1024, lds, cap, -1, ALL);
#undef LAUNCH
    RR_CHECK_LAUNCH("rr_soft_nms_segments");
    return RR_OK;
}

extern "C" int rr_soft_nms_segments(float *boxes, const int *seg_off, int nseg, int max_seg_boxes,
                                    int stride, float sigma, float Nt, float threshold, int method,
                                    int *n_out, int *err_flag, void *workspace, hipStream_t stream)
{
    return soft_nms_launch(boxes, seg_off, nullptr, nseg, max_seg_boxes, stride, sigma, Nt, threshold, method, n_out,
                           err_flag, workspace, stream);
}

extern "C" int rr_soft_nms_ragged(float *boxes, const int *seg_off, const int *seg_len, int nseg, int max_seg_boxes,
                                  int stride, float sigma, float Nt, float threshold, int method,
                                  int *n_out, int *err_flag, void *workspace, hipStream_t stream)
{
    RR_CHECK_ARG(seg_len != nullptr, "rr_soft_nms_ragged: seg_len is null");
    return soft_nms_launch(boxes, seg_off, seg_len, nseg, max_seg_boxes, stride, sigma, Nt, threshold, method, n_out,
                           err_flag, workspace, stream);
}
