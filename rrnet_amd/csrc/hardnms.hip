// Batched hard NMS for gfx950: one workgroup per (image, class) segment, boxes in LDS.
//
// Replaces the per-image x per-class Python loop around torchvision.ops.nms at
// models/rrnet.py:56-80 (default stage-1 path, IoU > 0.7).  torchvision is a third-party
// dependency that is not vendored by the reference (parity unpinned): this follows the
// torchvision-0.3 CUDA definition — rows visited in descending score order, IoU without the +1
// convention, a row is suppressed when IoU > thresh with an earlier kept row.
// Rows of a segment must already be score-descending (rr_decode_topk + rr_group_by_class give
// that).  Compiled with -ffp-contract=off so the IoU compare matches the CPU oracle bit for bit.
// Latency-bound (one barrier per kept box); algorithmic traffic = n*6*4 B in + kept rows out.
#include "common.h"
#include "rrnet_hip.h"

namespace {
constexpr int NT = 256;

__global__ __launch_bounds__(NT) void hard_nms_kernel(float *boxes, const int *seg_off, float thresh, int cap, int *n_out)
{
    extern __shared__ __align__(16) float sm[];
    float *bx = sm;                                    // [cap][6]
    unsigned char *flag = reinterpret_cast<unsigned char *>(sm + (size_t)cap * 6);  // [cap]
    __shared__ int wsum[NT / 64];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int off = seg_off[seg];
    const int n = seg_off[seg + 1] - off;
    float *g = boxes + (long)off * 6;
    for (int i = tid; i < n * 6; i += NT) bx[i] = g[i];
    for (int i = tid; i < n; i += NT) flag[i] = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (flag[i]) continue;
        const float ix1 = bx[i * 6 + 0], iy1 = bx[i * 6 + 1], ix2 = bx[i * 6 + 2], iy2 = bx[i * 6 + 3];
        const float ai = (ix2 - ix1) * (iy2 - iy1);
        for (int j = i + 1 + tid; j < n; j += NT) {
            if (flag[j]) continue;
            const float jx1 = bx[j * 6 + 0], jy1 = bx[j * 6 + 1], jx2 = bx[j * 6 + 2], jy2 = bx[j * 6 + 3];
            const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
            const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
            const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
            const float inter = w * h;
            const float aj = (jx2 - jx1) * (jy2 - jy1);
            const float ovr = inter / (ai + aj - inter);
            if (ovr > thresh) flag[j] = 1;
        }
        __syncthreads();
    }
    __syncthreads();
    // stable compaction of the kept rows to the front of the segment
    int run = 0;
    const int lane = tid & 63, wave = tid >> 6;
    for (int base = 0; base < n; base += NT) {
        const int i = base + tid;
        const bool keep = i < n && !flag[i];
        const unsigned long long m = __ballot(keep);
        const int within = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int before = 0, tot = 0;
        for (int w = 0; w < NT / 64; ++w) {
            if (w < wave) before += wsum[w];
            tot += wsum[w];
        }
        if (keep) {
            float *d = g + (long)(run + before + within) * 6;
#pragma unroll
            for (int e = 0; e < 6; ++e) d[e] = bx[i * 6 + e];
        }
        run += tot;
    }
    if (tid == 0) n_out[seg] = run;
}
}  // namespace

extern "C" int rr_hard_nms_segments(float *boxes, const int *seg_off, int nseg, int max_seg_boxes, float thresh,
                                    int *n_out, hipStream_t stream)
{
    RR_CHECK_ARG(nseg >= 0 && max_seg_boxes >= 0, "rr_hard_nms_segments: negative size");
    RR_CHECK_ARG(max_seg_boxes <= 6000, "rr_hard_nms_segments: segment of %d boxes (limit 6000)", max_seg_boxes);
    if (nseg == 0) return RR_OK;
    const int cap = (max_seg_boxes + 3) & ~3;
    const size_t lds = (size_t)cap * 25 + 16;
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(hard_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(hard_nms_kernel, dim3(nseg), dim3(NT), lds, stream, boxes, seg_off, thresh, cap, n_out);
    RR_CHECK_LAUNCH("rr_hard_nms_segments");
    return RR_OK;
}
