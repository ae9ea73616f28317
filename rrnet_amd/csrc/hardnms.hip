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

__global__ __launch_bounds__(NT) void hard_nms_kernel(float *boxes, const int *seg_off, const int *seg_len, float thresh,
                                                      int cap, int *n_out)
{
    extern __shared__ __align__(16) float sm[];
    float *bx = sm;                                    // [cap][6]
    unsigned char *flag = reinterpret_cast<unsigned char *>(sm + (size_t)cap * 6);  // [cap]
    __shared__ int wsum[NT / 64];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int off = seg_off[seg];
    const int n = seg_len ? seg_len[seg] : seg_off[seg + 1] - off;
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

// ---- the reference's own hard-NMS family (ext/nms/nms/nms_kernel.cu, cpu_nms.pyx:129-176, py_cpu_nms.py) ----------
// Legacy "+1" pixel convention; boxes arrive score-descending.  Same two-phase design as nms_kernel.cu, whose
// 64-bit mask words are exactly one wavefront wide here: phase 1 fills mask[i][w] = boxes of column block w that
// box i suppresses (upper triangle only); phase 2 walks the boxes in order with the running "removed" bit set in LDS
// — on the device, so the n*n/8 bytes of mask never cross PCIe (the reference copies them to the host and reduces there).
// inclusive = 0: suppress when IoU > thresh (nms_kernel.cu:71, py_cpu_nms.py:29); 1: IoU >= thresh (cpu_nms.pyx:170).
__device__ __forceinline__ float iou_plus1(const float *a, const float *b)
{
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
    const float inter = width * height;
    const float sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    const float sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return inter / (sa + sb - inter);
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float *boxes, int n, int stride, float thresh, int inclusive,
                                                      unsigned long long *mask)
{
    const int row_blk = blockIdx.y, col_blk = blockIdx.x;
    const int col_blocks = (n + 63) / 64;
    const int i = row_blk * 64 + threadIdx.x;
    if (col_blk < row_blk) {                       // lower triangle: never read by the reduction
        return;
    }
    __shared__ float cb[64 * 4];
    const int cj = col_blk * 64 + threadIdx.x;
    if (cj < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cb[threadIdx.x * 4 + e] = boxes[(long)cj * stride + e];
    }
    __syncthreads();
    if (i >= n) return;
    float me[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) me[e] = boxes[(long)i * stride + e];
    const int ncol = n - col_blk * 64 < 64 ? n - col_blk * 64 : 64;
    unsigned long long t = 0;
    for (int j = (row_blk == col_blk ? threadIdx.x + 1 : 0); j < ncol; ++j) {
        const float ovr = iou_plus1(me, cb + j * 4);
        if (inclusive ? ovr >= thresh : ovr > thresh) t |= 1ull << j;
    }
    mask[(long)i * col_blocks + col_blk] = t;
}

__global__ __launch_bounds__(256) void nms_reduce_kernel(const unsigned long long *mask, int n, int *keep, int *num_out)
{
    extern __shared__ unsigned long long remv[];   // [col_blocks]
    const int col_blocks = (n + 63) / 64;
    for (int j = threadIdx.x; j < col_blocks; j += 256) remv[j] = 0ull;
    __syncthreads();
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if ((remv[i >> 6] >> (i & 63)) & 1ull) continue;       // uniform: every thread reads the same word
        if (threadIdx.x == 0) keep[k] = i;
        ++k;
        __syncthreads();                                        // all reads of remv[i >> 6] done before it changes
        for (int j = (i >> 6) + threadIdx.x; j < col_blocks; j += 256) remv[j] |= mask[(long)i * col_blocks + j];
        __syncthreads();
    }
    if (threadIdx.x == 0) *num_out = k;
}
}  // namespace

extern "C" size_t rr_nms_workspace_bytes(int n)
{
    const size_t cb = (size_t)(n + 63) / 64;
    return (size_t)n * cb * sizeof(unsigned long long);
}

extern "C" int rr_nms_sorted(const float *boxes, int n, int stride, float thresh, int inclusive, void *workspace,
                             int *keep, int *num_out, hipStream_t stream)
{
    RR_CHECK_ARG(n >= 0 && stride >= 4, "rr_nms_sorted: bad dims");
    RR_CHECK_ARG(n <= 64 * 16384, "rr_nms_sorted: %d boxes (limit 1048576)", n);
    if (n == 0) {
        hipMemsetAsync(num_out, 0, sizeof(int), stream);
        return RR_OK;
    }
    RR_CHECK_ARG(workspace != nullptr, "rr_nms_sorted: workspace required (rr_nms_workspace_bytes)");
    const int cb = (n + 63) / 64;
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, stream, boxes, n, stride, thresh, inclusive, mask);
    RR_CHECK_LAUNCH("rr_nms_sorted(mask)");
    const size_t lds = (size_t)cb * 8;
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(nms_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(256), lds, stream, mask, n, keep, num_out);
    RR_CHECK_LAUNCH("rr_nms_sorted(reduce)");
    return RR_OK;
}

// Drop-in for `_nms` of ext/nms/nms/gpu_nms.hpp:1-2 (bound by gpu_nms.pyx:13-29): HOST pointers in and out, boxes
// pre-sorted by score, synchronous, allocates and frees its own device scratch like the reference does.
extern "C" void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
                     float nms_overlap_thresh, int device_id)
{
    *num_out = 0;
    if (boxes_num <= 0) return;
    int cur = 0;
    hipGetDevice(&cur);
    if (cur != device_id) hipSetDevice(device_id);
    float *boxes_dev = nullptr;
    void *ws = nullptr;
    int *keep_dev = nullptr;
    const size_t bbytes = (size_t)boxes_num * boxes_dim * sizeof(float);
    if (hipMalloc(&boxes_dev, bbytes) != hipSuccess || hipMalloc(&ws, rr_nms_workspace_bytes(boxes_num)) != hipSuccess ||
        hipMalloc(&keep_dev, ((size_t)boxes_num + 1) * sizeof(int)) != hipSuccess) {
        rr_set_error("_nms: hipMalloc failed");
    } else {
        hipMemcpy(boxes_dev, boxes_host, bbytes, hipMemcpyHostToDevice);
        if (rr_nms_sorted(boxes_dev, boxes_num, boxes_dim, nms_overlap_thresh, 0, ws, keep_dev + 1, keep_dev, nullptr) == RR_OK) {
            hipStreamSynchronize(nullptr);
            hipMemcpy(num_out, keep_dev, sizeof(int), hipMemcpyDeviceToHost);
            hipMemcpy(keep_out, keep_dev + 1, (size_t)*num_out * sizeof(int), hipMemcpyDeviceToHost);
        }
    }
    hipFree(boxes_dev);
    hipFree(ws);
    hipFree(keep_dev);
}

extern "C" int rr_hard_nms_segments(float *boxes, const int *seg_off, const int *seg_len, int nseg, int max_seg_boxes,
                                    float thresh, int *n_out, hipStream_t stream)
{
    RR_CHECK_ARG(nseg >= 0 && max_seg_boxes >= 0, "rr_hard_nms_segments: negative size");
    RR_CHECK_ARG(max_seg_boxes <= 6000, "rr_hard_nms_segments: segment of %d boxes (limit 6000)", max_seg_boxes);
    if (nseg == 0) return RR_OK;
    const int cap = (max_seg_boxes + 3) & ~3;
    const size_t lds = (size_t)cap * 25 + 16;
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(hard_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(hard_nms_kernel, dim3(nseg), dim3(NT), lds, stream, boxes, seg_off, seg_len, thresh, cap, n_out);
    RR_CHECK_LAUNCH("rr_hard_nms_segments");
    return RR_OK;
}
