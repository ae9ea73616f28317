// Inference post-process of RRNet for gfx950: re-regressed boxes, score filter and the final per-frame
// ordering, batched over frames and (frame, class) segments.
//
// Replaces, for a whole batch of frames per launch,
//   operators/rrnet_operator.py:188-209  generate_bbox  (stage-2 boxes from RoIs + regression)
//   operators/rrnet_operator.py:266-267  `pred_bbox[pred_bbox[:, 4] > 0.01]`
//   operators/rrnet_operator.py:222-223  xywh -> xyxy in front of soft_nms
//   operators/rrnet_operator.py:231      xyxy -> xywh behind it
//   operators/rrnet_operator.py:272-279  sort by score (descending) before and after _ext_nms
// The reference runs these as ~20 small torch kernels + a D2H copy + Python loops per frame and class.
// Built with -ffp-contract=off: every arithmetic step is a separate fp32 rounding like the eager torch ops.
// HBM traffic is R*(5+4+2)*4 B in and R*6*4 B out per call: latency-bound, not bandwidth-bound.
#include "common.h"
#include "rrnet_hip.h"

namespace {

// One workgroup per stage-1 (frame, class) segment: rows [seg_off[s], seg_off[s+1]) of the packed RoI list.
// out rows (x1, y1, x2, y2, score, cls+1) of the boxes passing `score > thr` are written, order preserved, to the
// front of the same row range; seg_len[s] = how many.
__global__ __launch_bounds__(256) void refine_boxes_kernel(const float *rois, const float *reg, const float *scores,
                                                           const float *clses, const int *seg_off, float scale, float thr,
                                                           float *out6, int *seg_len)
{
    __shared__ int wave_cnt[4];
    __shared__ int run;
    const int s = blockIdx.x;
    const int r0 = seg_off[s], n = seg_off[s + 1] - r0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        bool keep = false;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f, sc = 0.f, cl = 0.f;
        if (i < n) {
            const long r = r0 + i;
            const float *q = rois + r * 5;
            const float *g = reg + r * 4;
            // generate_bbox: xyxy * scale -> xywh -> w,h += 1 -> centre / size update -> xywh
            const float x = q[1] * scale, y = q[2] * scale;
            const float w = (q[3] * scale - x) + 1.0f, h = (q[4] * scale - y) + 1.0f;
            const float cx = (g[0] * w + x) + w / 2.0f;
            const float cy = (g[1] * h + y) + h / 2.0f;
            const float ow = expf(g[2]) * w, oh = expf(g[3]) * h;
            o0 = cx - ow / 2.0f;
            o1 = cy - oh / 2.0f;
            o2 = o0 + ow;           // _ext_nms: x2 = x + w
            o3 = o1 + oh;
            sc = scores[r];
            cl = clses[r] + 1.0f;
            keep = sc > thr;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int pos = run + __popcll(m & ((1ull << lane) - 1ull));
        for (int wv = 0; wv < wave; ++wv) pos += wave_cnt[wv];
        if (keep) {
            float *o = out6 + (long)(r0 + pos) * 6;
            o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = sc; o[5] = cl;
        }
        __syncthreads();
        if (threadIdx.x == 0) run += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) seg_len[s] = run;
}

// One workgroup per frame: the kept rows of the frame's segments (class order, as np.concatenate at
// rrnet_operator.py:225 leaves them) -> xywh -> stable sort by score descending -> out rows at
// frame_off = out_off[frame * segs_per_frame].
__global__ __launch_bounds__(256) void finalize_frames_kernel(const float *boxes6, const int *seg_off, const int *n_out,
                                                              const int *out_off, int segs_per_frame, int KP, float *out6)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);   // [KP]
    const int f = blockIdx.x;
    const int s0 = f * segs_per_frame;
    const int o0 = out_off[s0];
    const int total = out_off[s0 + segs_per_frame] - o0;
    for (int i = threadIdx.x; i < KP; i += 256) keys[i] = 0ull;
    __syncthreads();
    for (int s = s0; s < s0 + segs_per_frame; ++s) {
        const int n = n_out[s], src0 = seg_off[s], dst0 = out_off[s] - o0;
        for (int i = threadIdx.x; i < n; i += 256) {
            const unsigned int sc = __float_as_uint(boxes6[(long)(src0 + i) * 6 + 4]);
            const unsigned int ord = (sc & 0x80000000u) ? ~sc : (sc | 0x80000000u);
            // high word: score order; low word: ~position in the frame's concatenation (earlier row wins a tie)
            keys[dst0 + i] = ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned int)(dst0 + i));
        }
    }
    __syncthreads();
    for (int k2 = 2; k2 <= KP; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < KP / 2; t += 256) {
                const int lo = ((t / j) * 2 * j) + (t % j);
                const int hi = lo + j;
                const bool desc = ((lo & k2) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int k = threadIdx.x; k < total; k += 256) {
        const int pos = (int)(0xffffffffu - (unsigned int)(keys[k] & 0xffffffffull));
        // segment holding that position: segs_per_frame is small (classes), linear search
        int s = s0;
        while (s + 1 < s0 + segs_per_frame && out_off[s + 1] - o0 <= pos) ++s;
        const float *r = boxes6 + (long)(seg_off[s] + (pos - (out_off[s] - o0))) * 6;
        float *o = out6 + (long)(o0 + k) * 6;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2] - r[0]; o[3] = r[3] - r[1]; o[4] = r[4]; o[5] = r[5];
    }
}

// rows [n,6] -> out [n,6] ordered by score (column 4) descending, equal scores in input order: the two
// `torch.sort(pred_bbox[:, 4], descending=True)` of operators/rrnet_operator.py:272-279 on the device (one workgroup,
// LDS bitonic sort of (score, ~position) keys; n <= 16384).
__global__ __launch_bounds__(1024) void sort_rows_kernel(const float *rows, int n, int KP, float *out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
    for (int i = threadIdx.x; i < KP; i += 1024) {
        unsigned long long key = 0ull;
        if (i < n) {
            const unsigned int sc = __float_as_uint(rows[(long)i * 6 + 4]);
            const unsigned int ord = (sc & 0x80000000u) ? ~sc : (sc | 0x80000000u);
            key = ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned int)i);
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= KP; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < KP / 2; t += 1024) {
                const int lo = ((t / j) * 2 * j) + (t % j);
                const int hi = lo + j;
                const bool desc = ((lo & k2) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int k = threadIdx.x; k < n; k += 1024) {
        const int pos = (int)(0xffffffffu - (unsigned int)(keys[k] & 0xffffffffull));
#pragma unroll
        for (int e = 0; e < 6; ++e) out[(long)k * 6 + e] = rows[(long)pos * 6 + e];
    }
}

}  // namespace

extern "C" int rr_sort_rows_by_score(const float *rows6, int n, float *out6, hipStream_t stream)
{
    RR_CHECK_ARG(n >= 0 && n <= 16384, "rr_sort_rows_by_score: %d rows (limit 16384)", n);
    if (n == 0) return RR_OK;
    int kp = 2;
    while (kp < n) kp <<= 1;
    const size_t lds = (size_t)kp * 8;
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(sort_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(sort_rows_kernel, dim3(1), dim3(1024), lds, stream, rows6, n, kp, out6);
    RR_CHECK_LAUNCH("rr_sort_rows_by_score");
    return RR_OK;
}

extern "C" int rr_refine_boxes(const float *rois, const float *reg, const float *scores, const float *clses,
                               const int *seg_off, int nseg, float scale, float score_thr, float *out6, int *seg_len,
                               hipStream_t stream)
{
    RR_CHECK_ARG(nseg >= 0, "rr_refine_boxes: negative segment count");
    if (nseg == 0) return RR_OK;
    hipLaunchKernelGGL(refine_boxes_kernel, dim3(nseg), dim3(256), 0, stream, rois, reg, scores, clses, seg_off, scale,
                       score_thr, out6, seg_len);
    RR_CHECK_LAUNCH("rr_refine_boxes");
    return RR_OK;
}

extern "C" int rr_finalize_frames(const float *boxes6, const int *seg_off, const int *n_out, const int *out_off,
                                  int nframes, int segs_per_frame, int max_frame_boxes, float *out6, hipStream_t stream)
{
    RR_CHECK_ARG(nframes >= 0 && segs_per_frame > 0, "rr_finalize_frames: bad dims");
    RR_CHECK_ARG(max_frame_boxes >= 0 && max_frame_boxes <= 16384, "rr_finalize_frames: %d boxes per frame (limit 16384)",
                 max_frame_boxes);
    if (nframes == 0) return RR_OK;
    int kp = 2;
    while (kp < max_frame_boxes) kp <<= 1;
    const size_t lds = (size_t)kp * 8;
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(finalize_frames_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
    hipLaunchKernelGGL(finalize_frames_kernel, dim3(nframes), dim3(256), lds, stream, boxes6, seg_off, n_out, out_off,
                       segs_per_frame, kp, out6);
    RR_CHECK_LAUNCH("rr_finalize_frames");
    return RR_OK;
}
