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

}  // namespace

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
