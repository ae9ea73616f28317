/*
 * rrnet_hip.h — C ABI of librrnet_hip.so: the MI355X (gfx950) implementation of RRNet's
 * detection hot path.  Every entry point is `extern "C"`, takes plain device pointers and
 * sizes plus a hipStream_t, allocates nothing (scratch comes from an explicit
 * rr_*_workspace_bytes query), is asynchronous on the given stream and returns 0 or a
 * negative error code (text via rr_last_error(), thread-local).
 *
 * Each declaration cites the reference interface it replaces (file:line under
 * /root/reference).  INTEGRATION.md shows the reference-side binding (ctypes) a maintainer
 * would add.  Layouts: activations NHWC fp32 ("channels_last"), conv weights OHWI fp32
 * (= torch channels_last of the reference's [Cout,Cin,kh,kw] parameters), boxes row-major.
 */
#ifndef RRNET_HIP_H
#define RRNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_ABI_VERSION 1

#ifndef __HIP_INCLUDE_HIP_HIP_RUNTIME_API_H__
typedef struct ihipStream_t *hipStream_t;
#endif

/* ---- library ------------------------------------------------------------------------ */
const char *rr_last_error(void);
int rr_abi_version(void);

/* ---- Soft-NMS ----------------------------------------------------------------------- *
 * Replaces ext/nms/nms/cpu_nms.pyx:17-120 `cpu_soft_nms(boxes, sigma, Nt, threshold, method)`
 * (bound by ext/nms/nms_wrapper.py:13-19) and its per-class drivers
 * operators/rrnet_operator.py:211-232, models/rrnet.py:56-80, utils/metrics/metrics.py:308-324.
 * `boxes` holds nseg segments back to back, `stride` floats per row (>= 5; columns 0..4 =
 * x1,y1,x2,y2,score are permuted in place, further columns are left where they are, as in
 * the reference).  Segment s = rows [seg_off[s], seg_off[s+1]).  On return rows
 * [seg_off[s], seg_off[s] + n_out[s]) are the kept detections in the reference's order,
 * bit for bit.  method: 1 linear, 2 gaussian, else hard (weight 0).  *err_flag is set to 1
 * where the reference would raise ZeroDivisionError (union area == 0); it must be zeroed by
 * the caller.  Segments above RR_SOFT_NMS_LDS_MAX boxes need `workspace`
 * (rr_soft_nms_workspace_bytes); seg_off, n_out, err_flag are device pointers. */
#define RR_SOFT_NMS_LDS_MAX 6000
size_t rr_soft_nms_workspace_bytes(int total_boxes, int max_seg_boxes);
int rr_soft_nms_segments(float *boxes, const int *seg_off, int nseg, int max_seg_boxes, int stride,
                         float sigma, float Nt, float threshold, int method, int *n_out,
                         int *err_flag, void *workspace, hipStream_t stream);

/* ---- Convolutions (NHWC fp32, implicit GEMM on v_mfma_f32_32x32x2_f32) --------------- *
 * Replace what the reference delegates to cuDNN through nn.Conv2d:
 * backbones/hourglass.py:17,21,25,48,143,167,173 (ResidualBlock / ConvBNRelu / stem / inter),
 * detectors/centernet_detector.py:62,73,85 (3x3, 17x1, 1x17, 1x1 heads),
 * detectors/fasterrcnn_detector.py:11 + backbones/resnet.py:22-28 (stage-2 Bottleneck).
 * x [n,h,w,c], w [k][r][s][c] (OHWI), y [n,p,q,k], p = (h + 2*pad_h - r)/stride + 1.
 * fprop: optional bias[k], optional fused ReLU, optional per-block BatchNorm partial sums
 *   (`stat_slab`, rr_conv_stat_slab_bytes bytes: [ceil(n*p*q/128)][2][k] doubles = column
 *   sums and sums of squares of y, reduced by rr_bn_finalize).
 * dgrad: dx [n,h,w,c] = conv-transpose of dy [n,p,q,k]; accumulate != 0 adds into dx.
 * wgrad: dw [k][r][s][c] += x (*) dy, split over the n*p*q pixels and summed with float
 *   atomics (the caller zeroes dw once per step; the flat gradient buffer is that target). */
size_t rr_conv_stat_slab_bytes(int n, int p, int q, int k);
int rr_conv_fprop(const float *x, const float *w, const float *bias, float *y, double *stat_slab,
                  int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                  int relu, hipStream_t stream);
int rr_conv_dgrad(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k,
                  int r, int s, int stride, int pad_h, int pad_w, int accumulate, hipStream_t stream);
int rr_conv_wgrad(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k,
                  int r, int s, int stride, int pad_h, int pad_w, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RRNET_HIP_H */
