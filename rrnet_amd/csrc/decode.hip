// Heat-map decode for gfx950: sigmoid -> top-K -> gather -> box assembly, one workgroup per image.
//
// Replaces models/rrnet.py:93-138 (RRNet._topk + _gather_feat + _transpose_and_gather_feat +
// transform_bbox): torch.topk over [B,C,H*W], a second torch.topk over [B,C*K], five gathers and
// two NHWC permute copies in the reference.  The two-level top-k equals one global top-K over the
// C*H*W scores of an image (every member of the global top-K is inside its class's top-K), which is
// what this kernel computes with a 3-pass radix select (11+11+10 bits, LDS histograms), an
// LDS-resident bitonic sort of the K winners and a fused gather of offset / wh.
// Ordering: score descending; equal scores ordered by the reference's flat index c*H*W + y*W + x
// ascending (torch.topk leaves the order of ties unspecified).
// HBM-bound: algorithmic bytes = C*H*W*4 per image (the map is read once from HBM, the later
// passes hit L2) + K*(2+2)*4 gathered + K*6*4 written.
//
// Also here: the optional 3x3 peak filter (operators/centernet_operator.py:204-210 `_ctnet_nms`,
// dead code in the reference, named by north_star) and the per-class grouping / packing helpers
// that turn the reference's per-image x per-class Python loops (models/rrnet.py:37-46,56-80,
// operators/rrnet_operator.py:211-232) into batched launches.
#include "common.h"
#include "rrnet_hip.h"

namespace {

constexpr int DT = 1024;  // threads per image

__device__ __forceinline__ unsigned int f2ord(float f)
{
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ float score_of(float v, int is_logits) { return is_logits ? 1.0f / (1.0f + expf(-v)) : v; }

// wave 0: find the bin (scanning from the top) where the running count reaches `need`.
// result[0] = bin, result[1] = count strictly above that bin.
__device__ void find_bin(const int *hist, int nbins, int need, int *result)
{
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const int per = nbins / 64;
    const int hi = nbins - 1 - lane * per;  // this lane owns bins hi, hi-1, ..., hi-per+1
    int mine = 0;
    for (int k = 0; k < per; ++k) mine += hist[hi - k];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const int excl = incl - mine;
    if (excl < need && incl >= need) {
        int run = excl;
        for (int k = 0; k < per; ++k) {
            const int h = hist[hi - k];
            if (run + h >= need) {
                result[0] = hi - k;
                result[1] = run;
                break;
            }
            run += h;
        }
    }
}

constexpr int DEC_CAP = 8192;      // candidate slots of the fast path (64 KB of LDS)
constexpr int DEC_SAMPLES = 16384;  // strided sample that places the candidate threshold
constexpr int SCAN_T = 256;         // threads of a scan workgroup (multi-workgroup path)
constexpr int SCAN_PER = 32;        // floats per thread of a scan workgroup (8 float4; decode_scan_kernel is written for 8)

// Per-frame scratch of the multi-workgroup path (rr_decode_workspace_bytes): the sampled threshold, the
// candidate counter and DEC_CAP candidate slots.
typedef float4 f32x4s;

struct DecodeFrameWs {
    unsigned int thr;    // ordered key of the candidate threshold (0: every element is a candidate)
    int cnt;             // candidates appended so far (may exceed DEC_CAP: then the exact path runs)
    int pad[2];
    unsigned long long cand[DEC_CAP];   // (ordered raw value << 32) | flat NHWC index
};

// `_ctnet_nms` (operators/centernet_operator.py:204-210): an element survives iff its score equals the maximum of
// its 3x3 window (same class, window clipped at the border = max_pool2d's -inf padding).  The comparison is on
// SCORES like the reference's (distinct logits may share one fp32 sigmoid); raw values decide first because the
// score is monotone in them, expf runs only for a neighbour with a larger raw value.
__device__ __forceinline__ bool is_peak3x3(const float *x, long i, int C, int H, int W, int is_logits)
{
    const long pix = i / C;
    const int xw = (int)(pix % W), y = (int)(pix / W);
    const float v = x[i];
    float sv = 0.f;
    bool have_sv = false;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = xw + dx;
            if (xx < 0 || xx >= W || (dx == 0 && dy == 0)) continue;
            const float u = x[i + ((long)dy * W + dx) * C];
            if (u > v) {
                if (!is_logits) return false;
                if (!have_sv) { sv = score_of(v, 1); have_sv = true; }
                if (score_of(u, 1) > sv) return false;
            }
        }
    }
    return true;
}

// value the top-K ranks element i by: its score, or 0 where the optional peak filter removes it (heat * keep)
__device__ __forceinline__ float elem_score(const float *x, long i, int is_logits, int peak, int C, int H, int W)
{
    if (peak && !is_peak3x3(x, i, C, H, W, is_logits)) return 0.f;
    return score_of(x[i], is_logits);
}

// Places the candidate threshold of one frame from a strided sample so that ~2K..3K of the n raw values (of the
// surviving peaks when the filter is on) lie at or above it.  All threads of the workgroup call it; returns the
// ordered key (0 = take everything) or 0xffffffff when the fast path cannot be used (K too close to the capacity).
__device__ unsigned int place_threshold(const float *x, long n, int K, int peak, int is_logits, int C, int H, int W,
                                        int *hist, int *res)
{
    const int tid = threadIdx.x;
    if (n <= DEC_CAP) return 0u;
    // Large maps: the sample is DEC_SAMPLES / 32 whole 128-byte lines (32 consecutive floats each), evenly spaced — a
    // strided sample touches one line per element (0.54 GB fetched per 128 frames of 270x480x10 in round 2, as much as
    // the scan itself); lines cost 1/32 of that.  Maps too small for well-spread lines keep the element stride.
    // Consecutive values are correlated (neighbouring pixels of a peak), so a line sample is noisier than a strided one
    // of the same size: it is four times larger (2048 lines = 64 K values = 256 KB, 5 % of a 270x480x10 map) and aims
    // 25 % lower — a threshold that is too low only lengthens the candidate list, one that is too high sends the frame
    // down the exact path (4.7 ms instead of 60 us for the whole launch).
    constexpr long LINE_SAMPLES = 4l * DEC_SAMPLES;
    const bool by_line = n >= 8l * LINE_SAMPLES;
    const long lstride = by_line ? ((n / (LINE_SAMPLES / 32)) & ~31l) : 0;
    const long stride = n / DEC_SAMPLES > 0 ? n / DEC_SAMPLES : 1;
    const long ns = by_line ? LINE_SAMPLES : (n + stride - 1) / stride;
    long target = 2l * K > K + 1024l ? 2l * K : K + 1024l;
    if (by_line) target += target / 4;
    if (target > DEC_CAP / 2) target = DEC_CAP / 2;
    if (target < K) return 0xffffffffu;
    int need = (int)((double)target * (double)ns / (double)n);
    if (need < 8) need = 8;
    if (need > ns) return 0xffffffffu;
    unsigned int prefix = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int shift = pass == 0 ? 21 : 10;
        for (int i = tid; i < 2048; i += blockDim.x) hist[i] = 0;
        if (tid == 0) { res[0] = -1; res[1] = 0; }
        __syncthreads();
        for (long j = tid; j < ns; j += blockDim.x) {
            const long i = by_line ? (j >> 5) * lstride + (j & 31) : j * stride;
            const unsigned int o = f2ord(x[i]);
            if (pass == 0 || (o >> 21) == (prefix >> 21)) {
                if (!peak || is_peak3x3(x, i, C, H, W, is_logits)) atomicAdd(&hist[(o >> shift) & 2047], 1);
            }
        }
        __syncthreads();
        find_bin(hist, 2048, need, res);
        __syncthreads();
        const int bin = res[0], above = res[1];
        __syncthreads();
        if (bin < 0) return 0u;               // fewer surviving samples than `need`: take every survivor
        prefix |= (unsigned int)bin << shift;
        need -= above;
    }
    return prefix;                            // low 10 bits zero: at or just below the sampled rank
}

// m candidates (ordered raw value, flat index) in keys[0..m) -> (score, reference flat index) keys, sorted
// score-descending; returns m, or -1 when an element outside the candidate set could tie with the K-th score
// (then the exact path decides).
__device__ int finish_candidates(unsigned long long *keys, int m, unsigned int t0, int is_logits, int peak, int C, long HW,
                                 int K)
{
    const int tid = threadIdx.x;
    for (int p = tid; p < m; p += DT) {
        const unsigned long long cand = keys[p];
        const long i = (long)(cand & 0xffffffffull);
        const float sc = score_of(ord2f((unsigned int)(cand >> 32)), is_logits);
        const unsigned int ref = (unsigned int)((i % C) * HW + i / C);
        keys[p] = ((unsigned long long)f2ord(sc) << 32) | (unsigned long long)(0xffffffffu - ref);
    }
    int mp = 2;
    while (mp < m) mp <<= 1;
    for (int p = m + tid; p < mp; p += DT) keys[p] = 0ull;
    __syncthreads();
    for (int k2 = 2; k2 <= mp; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < mp / 2; t += DT) {
                const int lo = ((t / j) * 2 * j) + (t % j);
                const int hi2 = lo + j;
                const bool desc = ((lo & k2) == 0);
                const unsigned long long a = keys[lo], b = keys[hi2];
                if ((a < b) == desc) { keys[lo] = b; keys[hi2] = a; }
            }
            __syncthreads();
        }
    }
    const unsigned int kth = (unsigned int)(keys[K - 1] >> 32);
    if (t0 != 0) {
        const unsigned int edge = f2ord(score_of(ord2f(t0), is_logits));
        if (kth <= edge) return -1;           // an outsider could tie with the K-th score
    }
    if (peak && kth <= f2ord(0.f)) return -1; // filtered-out elements rank as 0: they could tie / outrank
    return m;
}

// Single-workgroup fast path (small maps, or no workspace): threshold, ONE scan into LDS, finish.
// Raw values are compared (sigmoid is monotone), so expf runs on the candidates only.
__device__ int decode_fast_path(const float *x, long n, int is_logits, int peak, int C, int H, int W, int K,
                                unsigned long long *keys, int *hist, int *res, int *cnt)
{
    const int tid = threadIdx.x;
    const unsigned int t0 = place_threshold(x, n, K, peak, is_logits, C, H, W, hist, res);
    if (t0 == 0xffffffffu) return -1;
    if (tid == 0) *cnt = 0;
    __syncthreads();
    for (long i = tid; i < n; i += DT) {
        const unsigned int o = f2ord(x[i]);
        if (o >= t0 && (!peak || is_peak3x3(x, i, C, H, W, is_logits))) {
            const int p = atomicAdd(cnt, 1);
            if (p < DEC_CAP) keys[p] = ((unsigned long long)o << 32) | (unsigned long long)i;
        }
    }
    __syncthreads();
    const int m = *cnt;
    if (m < K || m > DEC_CAP) return -1;
    return finish_candidates(keys, m, t0, is_logits, peak, C, (long)H * W, K);
}

// ---- multi-workgroup path: (1) threshold per frame, (2) all CUs stream the maps and append candidates,
// ---- (3) one workgroup per frame sorts its candidates and assembles the boxes -------------------------------
__global__ __launch_bounds__(DT) void decode_threshold_kernel(const float *hm, int is_logits, int peak, int H, int W, int C,
                                                              int K, DecodeFrameWs *ws)
{
    __shared__ int hist[2048];
    __shared__ int res[2];
    const long n = (long)H * W * C;
    const unsigned int t0 = place_threshold(hm + (long)blockIdx.x * n, n, K, peak, is_logits, C, H, W, hist, res);
    if (threadIdx.x == 0) {
        ws[blockIdx.x].thr = t0;
        ws[blockIdx.x].cnt = 0;
    }
}

// rare path of the scan (≈0.2 % of the elements): kept out of line so that the streaming loop stays small, fully
// unrolled and in registers (inlined 32 times it pushed the loaded float4s into scratch memory and doubled the traffic)
__device__ __noinline__ void scan_push(const float *x, unsigned int o, long i, int peak, int is_logits, int C, int H, int W,
                                       unsigned long long *lcand, int *lcnt, int lcap, DecodeFrameWs *f)
{
    if (peak && !is_peak3x3(x, i, C, H, W, is_logits)) return;
    const unsigned long long key = ((unsigned long long)o << 32) | (unsigned long long)i;
    const int p = atomicAdd(lcnt, 1);
    if (p < lcap) {
        lcand[p] = key;
    } else {                                       // a workgroup with > lcap candidates (t0 == 0, flat maps): direct
        const int q = atomicAdd(&f->cnt, 1);
        if (q < DEC_CAP) f->cand[q] = key;
    }
}

__global__ __launch_bounds__(SCAN_T) void decode_scan_kernel(const float *hm, int is_logits, int peak, int H, int W, int C,
                                                             DecodeFrameWs *ws)
{
    // candidates of this workgroup are gathered in LDS (LDS atomics return in ~100 cycles; a returning global atomic
    // per candidate parked every wave for microseconds) and appended to the frame's list with ONE global atomic
    constexpr int LCAP = 512;
    __shared__ unsigned long long lcand[LCAP];
    __shared__ int lcnt, gbase;
    const long n = (long)H * W * C;
    const float *x = hm + (long)blockIdx.y * n;
    DecodeFrameWs *f = ws + blockIdx.y;
    const long base = (long)blockIdx.x * (SCAN_T * SCAN_PER);
    const bool full = (n & 3) == 0 && base + SCAN_T * SCAN_PER <= n;
    f32x4s v0, v1, v2, v3, v4, v5, v6, v7;
    if (full) {                                    // the streaming loads go out before the threshold is even known
        const f32x4s *p = reinterpret_cast<const f32x4s *>(x + base) + threadIdx.x;
        v0 = p[0 * SCAN_T]; v1 = p[1 * SCAN_T]; v2 = p[2 * SCAN_T]; v3 = p[3 * SCAN_T];
        v4 = p[4 * SCAN_T]; v5 = p[5 * SCAN_T]; v6 = p[6 * SCAN_T]; v7 = p[7 * SCAN_T];
    }
    const unsigned int t0 = f->thr;
    if (t0 == 0xffffffffu) return;
    if (threadIdx.x == 0) lcnt = 0;
    __syncthreads();
    if (full) {
#define RR_SCAN4(V, R)                                                                                              \
        {                                                                                                           \
            const long i0 = base + ((long)(R) * SCAN_T + threadIdx.x) * 4;                                          \
            const unsigned int o0 = f2ord(V.x), o1 = f2ord(V.y), o2 = f2ord(V.z), o3 = f2ord(V.w);                  \
            if (o0 >= t0) scan_push(x, o0, i0, peak, is_logits, C, H, W, lcand, &lcnt, LCAP, f);                    \
            if (o1 >= t0) scan_push(x, o1, i0 + 1, peak, is_logits, C, H, W, lcand, &lcnt, LCAP, f);                \
            if (o2 >= t0) scan_push(x, o2, i0 + 2, peak, is_logits, C, H, W, lcand, &lcnt, LCAP, f);                \
            if (o3 >= t0) scan_push(x, o3, i0 + 3, peak, is_logits, C, H, W, lcand, &lcnt, LCAP, f);                \
        }
        RR_SCAN4(v0, 0) RR_SCAN4(v1, 1) RR_SCAN4(v2, 2) RR_SCAN4(v3, 3)
        RR_SCAN4(v4, 4) RR_SCAN4(v5, 5) RR_SCAN4(v6, 6) RR_SCAN4(v7, 7)
#undef RR_SCAN4
    } else {
        const long end = base + SCAN_T * SCAN_PER < n ? base + SCAN_T * SCAN_PER : n;
        for (long i = base + threadIdx.x; i < end; i += SCAN_T) {
            const unsigned int o = f2ord(x[i]);
            if (o >= t0) scan_push(x, o, i, peak, is_logits, C, H, W, lcand, &lcnt, LCAP, f);
        }
    }
    __syncthreads();
    const int m = lcnt < LCAP ? lcnt : LCAP;
    if (m == 0) return;
    if (threadIdx.x == 0) gbase = atomicAdd(&f->cnt, m);
    __syncthreads();
    for (int p = threadIdx.x; p < m; p += SCAN_T)
        if (gbase + p < DEC_CAP) f->cand[gbase + p] = lcand[p];
}

__global__ __launch_bounds__(DT) void decode_topk_kernel(const float *hm, int is_logits, int peak, const float *wh,
                                                         const float *off, int H, int W, int C, int K, int KP, float *out,
                                                         int *pix_out, const DecodeFrameWs *ws, int box_mode, float scale)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);  // [max(KP, DEC_CAP)]
    int *hist = reinterpret_cast<int *>(keys + (KP > DEC_CAP ? KP : DEC_CAP)); // [2048]
    __shared__ int res[2];
    __shared__ int cnt_gt;

    const int tid = threadIdx.x;
    const long HW = (long)H * W;
    const long n = HW * C;
    const float *x = hm + (long)blockIdx.x * n;

    int fast;
    if (ws) {                                     // candidates were collected by decode_scan_kernel
        const DecodeFrameWs *f = ws + blockIdx.x;
        const int m = f->cnt;
        const unsigned int t0 = f->thr;
        fast = -1;
        if (t0 != 0xffffffffu && m >= K && m <= DEC_CAP) {
            for (int p = tid; p < m; p += DT) keys[p] = f->cand[p];
            __syncthreads();
            fast = finish_candidates(keys, m, t0, is_logits, peak, C, HW, K);
        }
    } else {
        fast = decode_fast_path(x, n, is_logits, peak, C, H, W, K, keys, hist, res, &cnt_gt);
    }
    __syncthreads();
    if (fast < 0) {
    // ---- radix select of the K-th largest 64-bit key (ordered score << 32 | ~reference index): passes 0-2 fix
    // the score word (11+11+10 bits), passes 3-5 the index word among the elements that tie with the K-th score,
    // so ties across the K-th rank go to the lowest reference index like everywhere else.  Keys are unique:
    // exactly K elements are >= the selected key.
    unsigned int hi_pref = 0, lo_pref = 0;   // bits fixed so far (top-aligned) of the two words
    int need = K;
    for (int pass = 0; pass < 6; ++pass) {
        const int sub = pass % 3;
        const int shift = sub == 0 ? 21 : (sub == 1 ? 10 : 0);
        const int nb = sub == 2 ? 1024 : 2048;
        for (int i = tid; i < 2048; i += DT) hist[i] = 0;
        __syncthreads();
        for (long i = tid; i < n; i += DT) {
            const unsigned int o = f2ord(elem_score(x, i, is_logits, peak, C, H, W));
            unsigned int word, pref;
            if (pass < 3) {
                word = o; pref = hi_pref;
            } else {
                if (o != hi_pref) continue;
                word = 0xffffffffu - (unsigned int)((i % C) * HW + i / C); pref = lo_pref;
            }
            const bool in = sub == 0 ? true : (sub == 1 ? (word >> 21) == (pref >> 21) : (word >> 10) == (pref >> 10));
            if (in) atomicAdd(&hist[(word >> shift) & (nb - 1)], 1);
        }
        __syncthreads();
        find_bin(hist, nb, need, res);
        __syncthreads();
        if (pass < 3) hi_pref |= (unsigned int)res[0] << shift; else lo_pref |= (unsigned int)res[0] << shift;
        need -= res[1];
        __syncthreads();
    }
    const unsigned long long kth = ((unsigned long long)hi_pref << 32) | (unsigned long long)lo_pref;

    // ---- collect the K keys >= kth (any order: sorted next)
    if (tid == 0) cnt_gt = 0;
    for (int i = K + tid; i < KP; i += DT) keys[i] = 0ull;
    __syncthreads();
    for (long i = tid; i < n; i += DT) {
        const unsigned int o = f2ord(elem_score(x, i, is_logits, peak, C, H, W));
        if (o < hi_pref) continue;
        const unsigned int ref = (unsigned int)((i % C) * HW + i / C);
        const unsigned long long key = ((unsigned long long)o << 32) | (unsigned long long)(0xffffffffu - ref);
        if (key >= kth) {
            const int p = atomicAdd(&cnt_gt, 1);
            if (p < K) keys[p] = key;
        }
    }
    __syncthreads();

    // ---- bitonic sort, descending
    for (int k2 = 2; k2 <= KP; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < KP / 2; t += DT) {
                const int lo = ((t / j) * 2 * j) + (t % j);
                const int hi2 = lo + j;
                const bool desc = ((lo & k2) == 0);
                const unsigned long long a = keys[lo], b = keys[hi2];
                if ((a < b) == desc) { keys[lo] = b; keys[hi2] = a; }
            }
            __syncthreads();
        }
    }
    }   // exact path

    // ---- gather + box assembly (models/rrnet.py:122-137)
    for (int k = tid; k < K; k += DT) {
        const unsigned long long key = keys[k];
        const float score = ord2f((unsigned int)(key >> 32));
        const unsigned int ref = 0xffffffffu - (unsigned int)(key & 0xffffffffu);
        const int cls = (int)(ref / HW);
        const long pix = ref % HW;
        const float xs0 = (float)(pix % W), ys0 = (float)(pix / W);
        const float *po = off + ((long)blockIdx.x * HW + pix) * 2;
        const float *pw = wh + ((long)blockIdx.x * HW + pix) * 2;
        const float xs = xs0 + po[0], ys = ys0 + po[1];
        float *o = out + ((long)blockIdx.x * K + k) * 6;
        if (box_mode == 0) {      // RRNet (models/rrnet.py:122-137): wh clamped at 0, x1,y1,x2,y2 in feature coordinates
            const float w_ = fmaxf(pw[0], 0.f), h_ = fmaxf(pw[1], 0.f);
            const float px = xs - w_ / 2.f, py = ys - h_ / 2.f;
            o[0] = px; o[1] = py; o[2] = w_ + px; o[3] = h_ + py; o[4] = score; o[5] = (float)cls;
        } else {                  // CenterNet (operators/centernet_operator.py:152-178): no clamp, x,y,w,h * scale, cls+1
            const float w_ = pw[0], h_ = pw[1];
            o[0] = (xs - w_ / 2.f) * scale; o[1] = (ys - h_ / 2.f) * scale; o[2] = w_ * scale; o[3] = h_ * scale;
            o[4] = score; o[5] = (float)cls + 1.f;
        }
        if (pix_out) pix_out[(long)blockIdx.x * K + k] = (int)pix;
    }
}

// For every packed RoI: the heat-map pixel it was decoded from, found by matching its six floats against
// the image's decode rows (grouping / hard NMS / packing copy rows bit for bit).  -1 if no row matches.
__global__ void roi_provenance_kernel(const float *rois, const float *scores, const float *clses, int R, const float *decoded,
                                      const int *pix, int K, int *roi_pix)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float *q = rois + (long)r * 5;
    const int b = (int)q[0];
    const float *d = decoded + (long)b * K * 6;
    int found = -1;
    for (int k = 0; k < K; ++k) {
        const float *e = d + k * 6;
        if (e[4] == scores[r] && e[5] == clses[r] && e[0] == q[1] && e[1] == q[2] && e[2] == q[3] && e[3] == q[4]) {
            found = pix[(long)b * K + k];
            break;
        }
    }
    roi_pix[r] = found;
}

// Backward of the box assembly (models/rrnet.py:122-137): x1 = xs + off_x - w/2, x2 = x1 + w, w = max(wh_x, 0)
// => d off_x = dx1 + dx2, d wh_x = (dx2 - dx1)/2 where wh_x >= 0; same for y.  Scatter with float atomics.
__global__ void proposal_bwd_kernel(const float *droi, const float *rois, const int *roi_pix, int R, const float *wh, long HW,
                                    float *dwh, float *doff)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int p = roi_pix[r];
    if (p < 0) return;
    const int b = (int)rois[(long)r * 5];
    const float *g = droi + (long)r * 4;
    const long o = ((long)b * HW + p) * 2;
    const float gx = g[0] + g[2], gy = g[1] + g[3];
    if (gx != 0.f) atomicAdd(doff + o, gx);
    if (gy != 0.f) atomicAdd(doff + o + 1, gy);
    const float hx = 0.5f * (g[2] - g[0]), hy = 0.5f * (g[3] - g[1]);
    if (wh[o] >= 0.f && hx != 0.f) atomicAdd(dwh + o, hx);
    if (wh[o + 1] >= 0.f && hy != 0.f) atomicAdd(dwh + o + 1, hy);
}

// scores[b,y,x,c] = sigmoid(hm) if it equals the maximum of its 3x3 window (same class) else 0:
// operators/centernet_operator.py:204-210 as a stand-alone map (the decode applies the same test to its candidates
// only, see is_peak3x3).  One expf per element; neighbours are compared on raw values first.
__global__ void peak3x3_kernel(const float *hm, int is_logits, float *scores, int B, int H, int W, int C)
{
    const long per = (long)H * W * C;
    const long total = (long)B * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / per;
        const long j = i - b * per;
        const float *x = hm + b * per;
        scores[i] = is_peak3x3(x, j, C, H, W, is_logits) ? score_of(x[j], is_logits) : 0.f;
    }
}

// ---- stable grouping of each image's rows by class (rows arrive score-descending) ------------
// boxes [B,K,6] -> grouped [B,K,6] with classes ascending, order inside a class preserved;
// seg_off [B*NC+1] row offsets of every (image, class) segment in the flattened [B*K] row space.
__global__ __launch_bounds__(256) void group_by_class_kernel(const float *boxes, int K, int NC, int cls_base, int KP,
                                                             float *grouped, int *seg_off, int *seg_len)
{
    extern __shared__ int sm[];   // count[NC], then (LDS sort path) keys[KP]
    int *count = sm;
    unsigned int *keys = reinterpret_cast<unsigned int *>(sm + NC);
    __shared__ int n_valid;
    const int b = blockIdx.x;
    const int n = K;
    const float *src = boxes + (long)b * K * 6;
    float *dst = grouped + (long)b * K * 6;
    for (int c = threadIdx.x; c < NC; c += 256) count[c] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = (int)src[i * 6 + 5] - cls_base;
        if (c >= 0 && c < NC) atomicAdd(&count[c], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < NC; ++c) {
            const int cnt = count[c];
            count[c] = run;                    // becomes the segment's start row
            seg_off[b * NC + c] = b * K + run;
            if (seg_len) seg_len[b * NC + c] = cnt;
            run += cnt;
        }
        // rows of classes outside the range are dropped and leave a gap at the end of the image's K-row block:
        // seg_off[s+1] - seg_off[s] over-counts the image's last class by that gap, seg_len is exact
        if (b == gridDim.x - 1) seg_off[(b + 1) * NC] = b * K + run;
        n_valid = run;
    }
    __syncthreads();
    if (KP > 0) {
        // stable grouping = ascending sort of (class << 22 | row): LDS bitonic network, rows of classes outside
        // [cls_base, cls_base+NC) sort to the end and are dropped
        for (int i = threadIdx.x; i < KP; i += 256) {
            unsigned int key = 0xffffffffu;
            if (i < n) {
                const int c = (int)src[i * 6 + 5] - cls_base;
                if (c >= 0 && c < NC) key = ((unsigned int)c << 22) | (unsigned int)i;
            }
            keys[i] = key;
        }
        __syncthreads();
        for (int k2 = 2; k2 <= KP; k2 <<= 1) {
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < KP / 2; t += 256) {
                    const int lo = ((t / j) * 2 * j) + (t % j);
                    const int hi2 = lo + j;
                    const bool asc = ((lo & k2) == 0);
                    const unsigned int a = keys[lo], c = keys[hi2];
                    if ((a > c) == asc) { keys[lo] = c; keys[hi2] = a; }
                }
                __syncthreads();
            }
        }
        const int nv = n_valid;
        for (int e = threadIdx.x; e < nv * 6; e += 256) {
            const int r = e / 6, col = e - r * 6;
            dst[e] = src[(keys[r] & 0x3fffffu) * 6 + col];
        }
        return;
    }
    // very long inputs (beyond the LDS sort): one thread per class keeps the order stable
    for (int c = threadIdx.x; c < NC; c += 256) {
        int w = count[c];
        for (int i = 0; i < n; ++i) {
            if ((int)src[i * 6 + 5] - cls_base == c) {
#pragma unroll
                for (int e = 0; e < 6; ++e) dst[w * 6 + e] = src[i * 6 + e];
                ++w;
            }
        }
    }
}

// exclusive prefix of n_out over segments -> row offsets of the packed output (single block)
__global__ void seg_prefix_kernel(const int *n_out, int nseg, int *out_off /*[nseg+1]*/)
{
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nseg; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nseg ? n_out[i] : 0;
        // simple Hillis-Steele in LDS
        __shared__ int buf[1024];
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nseg) out_off[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_off[nseg] = carry;
}

// kept rows of every segment -> packed rois [R,5] = (image, x1,y1,x2,y2), scores [R], classes [R]
__global__ void pack_segments_kernel(const float *grouped, const int *seg_off, const int *n_out, const int *out_off,
                                     int segs_per_image, float *rois, float *scores, float *clses, float *rows6)
{
    const int s = blockIdx.x;
    const int n = n_out[s];
    const float *src = grouped + (long)seg_off[s] * 6;
    const int o0 = out_off[s];
    const float img = (float)(s / segs_per_image);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float *r = src + i * 6;
        if (rois) {
            float *q = rois + (long)(o0 + i) * 5;
            q[0] = img; q[1] = r[0]; q[2] = r[1]; q[3] = r[2]; q[4] = r[3];
            scores[o0 + i] = r[4];
            clses[o0 + i] = r[5];
        }
        if (rows6) {
            float *q = rows6 + (long)(o0 + i) * 6;
#pragma unroll
            for (int e = 0; e < 6; ++e) q[e] = r[e];
        }
    }
}

}  // namespace

extern "C" size_t rr_decode_workspace_bytes(int b) { return b > 0 ? (size_t)b * sizeof(DecodeFrameWs) : 0; }

extern "C" int rr_decode_topk(const float *hm, int is_logits, int peak_filter, const float *wh, const float *off, int b,
                              int h, int w, int c, int k, int box_mode, float scale, float *out, int *pix_out,
                              void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    RR_CHECK_ARG(b > 0 && h > 0 && w > 0 && c > 0, "rr_decode_topk: bad dims");
    RR_CHECK_ARG(k > 0 && k <= 4096 && (long)k <= (long)h * w * c, "rr_decode_topk: k=%d out of range (1..min(4096, C*H*W))", k);
    RR_CHECK_ARG((long)h * w * c < (1l << 31), "rr_decode_topk: map too large");
    int kp = 2;
    while (kp < k) kp <<= 1;
    const long n = (long)h * w * c;
    // maps worth spreading over the chip: threshold per frame, a streaming scan by all CUs, then one workgroup
    // per frame; small maps (or no workspace) stay in the single-workgroup kernel
    DecodeFrameWs *ws = nullptr;
    if (workspace && n >= 65536) {
        RR_CHECK_ARG(workspace_bytes >= rr_decode_workspace_bytes(b), "rr_decode_topk: workspace of %zu bytes, need %zu",
                     workspace_bytes, rr_decode_workspace_bytes(b));
        ws = static_cast<DecodeFrameWs *>(workspace);
        hipLaunchKernelGGL(decode_threshold_kernel, dim3(b), dim3(DT), 0, stream, hm, is_logits, peak_filter, h, w, c, k, ws);
        const int chunks = rr_cdiv(n, SCAN_T * SCAN_PER);
        hipLaunchKernelGGL(decode_scan_kernel, dim3(chunks, b), dim3(SCAN_T), 0, stream, hm, is_logits, peak_filter, h, w, c, ws);
    }
    const size_t lds = (size_t)(kp > DEC_CAP ? kp : DEC_CAP) * 8 + 2048 * 4 + (DT / 64 + 1) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(decode_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    RR_CHECK_ARG(box_mode == 0 || box_mode == 1, "rr_decode_topk: box_mode %d", box_mode);
    hipLaunchKernelGGL(decode_topk_kernel, dim3(b), dim3(DT), lds, stream, hm, is_logits, peak_filter, wh, off, h, w, c, k, kp,
                       out, pix_out, ws, box_mode, scale);
    RR_CHECK_LAUNCH("rr_decode_topk");
    return RR_OK;
}

extern "C" int rr_roi_provenance(const float *rois, const float *scores, const float *clses, int r, const float *decoded,
                                 const int *pix, int k, int *roi_pix, hipStream_t stream)
{
    if (r <= 0) return RR_OK;
    hipLaunchKernelGGL(roi_provenance_kernel, dim3(rr_cdiv(r, 128)), dim3(128), 0, stream, rois, scores, clses, r, decoded, pix,
                       k, roi_pix);
    RR_CHECK_LAUNCH("rr_roi_provenance");
    return RR_OK;
}

extern "C" int rr_proposal_bwd(const float *droi, const float *rois, const int *roi_pix, int r, const float *wh, int b, int h,
                               int w, float *dwh, float *doff, hipStream_t stream)
{
    RR_CHECK_ARG(b > 0 && h > 0 && w > 0, "rr_proposal_bwd: bad dims");
    hipMemsetAsync(dwh, 0, sizeof(float) * (size_t)b * h * w * 2, stream);
    hipMemsetAsync(doff, 0, sizeof(float) * (size_t)b * h * w * 2, stream);
    if (r > 0)
        hipLaunchKernelGGL(proposal_bwd_kernel, dim3(rr_cdiv(r, 128)), dim3(128), 0, stream, droi, rois, roi_pix, r, wh,
                           (long)h * w, dwh, doff);
    RR_CHECK_LAUNCH("rr_proposal_bwd");
    return RR_OK;
}

extern "C" int rr_peak3x3(const float *hm, int is_logits, float *scores, int b, int h, int w, int c, hipStream_t stream)
{
    const long total = (long)b * h * w * c;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(peak3x3_kernel, dim3((int)blocks), dim3(256), 0, stream, hm, is_logits, scores, b, h, w, c);
    RR_CHECK_LAUNCH("rr_peak3x3");
    return RR_OK;
}

extern "C" int rr_group_by_class(const float *boxes, int b, int k, int num_classes, int cls_base,
                                 float *grouped, int *seg_off, int *seg_len, hipStream_t stream)
{
    RR_CHECK_ARG(b > 0 && k > 0 && num_classes > 0 && num_classes <= 1024, "rr_group_by_class: bad dims");
    int kp = 2;
    while (kp < k) kp <<= 1;
    if (kp > 16384 || num_classes > 1023) kp = 0;   // LDS sort path: <= 64 KB of keys, class id in 10 bits
    const size_t lds = (size_t)(num_classes + kp) * sizeof(int);
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(group_by_class_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(group_by_class_kernel, dim3(b), dim3(256), lds, stream, boxes, k, num_classes, cls_base, kp,
                       grouped, seg_off, seg_len);
    RR_CHECK_LAUNCH("rr_group_by_class");
    return RR_OK;
}

extern "C" int rr_pack_segments(const float *grouped, const int *seg_off, const int *n_out, int nseg, int segs_per_image,
                                int *out_off, float *rois, float *scores, float *clses, float *rows6, int phase,
                                hipStream_t stream)
{
    RR_CHECK_ARG(nseg > 0 && segs_per_image > 0, "rr_pack_segments: bad dims");
    if (phase == 0) {   // offsets only: the caller reads out_off[nseg] to size the outputs
        hipLaunchKernelGGL(seg_prefix_kernel, dim3(1), dim3(1024), 0, stream, n_out, nseg, out_off);
    } else {
        hipLaunchKernelGGL(pack_segments_kernel, dim3(nseg), dim3(128), 0, stream, grouped, seg_off, n_out, out_off,
                           segs_per_image, rois, scores, clses, rows6);
    }
    RR_CHECK_LAUNCH("rr_pack_segments");
    return RR_OK;
}
