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
/* Same, with explicit segment lengths: segment s = rows [seg_off[s], seg_off[s] + seg_len[s]) (rows between the
 * length and the next offset are ignored) — the layout rr_refine_boxes leaves behind. */
int rr_soft_nms_ragged(float *boxes, const int *seg_off, const int *seg_len, int nseg, int max_seg_boxes, int stride,
                       float sigma, float Nt, float threshold, int method, int *n_out, int *err_flag,
                       void *workspace, hipStream_t stream);


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
 *   atomics (the caller zeroes dw once per step; the flat gradient buffer is that target).
 *   out_h/out_w > 0 give dy's spatial size explicitly (asymmetric padding: pad_h/pad_w are the leading
 *   pads); 0 derives it from the symmetric formula. */
size_t rr_conv_stat_slab_bytes(int n, int p, int q, int k);
int rr_conv_fprop(const float *x, const float *w, const float *bias, float *y, double *stat_slab,
                  int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                  int relu, hipStream_t stream);
int rr_conv_dgrad(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k,
                  int r, int s, int stride, int pad_h, int pad_w, int accumulate, hipStream_t stream);
/* Stride-1 data gradient through the forward kernel (its [n][k] weight fragments are read 16 bytes at a time):
 * rr_weight_flip_transpose writes wt[c][r-1-i][s-1-j][k] = w[k][i][j][c] (k*r*s*c floats of caller scratch), then
 * rr_conv_dgrad_s1(dy [n,p,q,k], wt) = dx [n,h,w,c], p = h + 2*pad_h - r + 1.  Same result as rr_conv_dgrad
 * (different summation order inside the fp32 accumulation chain). */
int rr_weight_flip_transpose(const float *w, float *wt, int k, int c, int r, int s, hipStream_t stream);
/* The same for every filter of a flat parameter buffer in ONE launch (host layer: rrnet_amd/flat.py keeps the flipped copies
 * of all stride-1 layers in a second flat buffer, refreshed once per optimizer step): table = ntiles x int4 {element offset
 * of the filter in flat / wt_flat, (r*s) << 16 | k, c, tap << 20 | k_tile << 10 | c_tile}, one 32 x 32 tile of the (k, c)
 * plane of one tap per workgroup.  Filters of up to 65535 output channels, 1024 tiles per side, r*s < 2048. */
int rr_weight_flip_transpose_batch(const float *flat, float *wt_flat, const int *table, int ntiles, hipStream_t stream);
/* ... and, for the bf16-operand convolutions, bf16 copies of both (w16_flat: the filters as they are, wt16_flat: flipped /
 * transposed; bf16 elements at the same element offsets; either may be NULL). */
int rr_weight_flip_transpose_batch_bf16(const float *flat, float *wt_flat, unsigned short *w16_flat, unsigned short *wt16_flat,
                                        const int *table, int ntiles, hipStream_t stream);
int rr_conv_dgrad_s1(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                     int r, int s, int pad_h, int pad_w, int accumulate, hipStream_t stream);
/* Small-channel convolutions (the 7x7 stride-2 stem on a 3-channel image, backbones/hourglass.py:143): rr_conv_pack_taps
 * writes out [n,p,q,kp], out[..,k] = x[n, p*stride-pad_h+r, q*stride-pad_w+s, c] for k = (r*S+s)*c_in + c < r*s*c and 0 up
 * to kp (a multiple of 4, the caller pads to 32).  The convolution then is a 1x1 rr_conv_fprop over kp channels with the
 * OHWI weight rows zero-padded to kp, and its weight gradient a 1x1 rr_conv_wgrad — both on the vector MFMA kernels. */
int rr_conv_pack_taps(const float *x, float *out, int n, int h, int wd, int c, int r, int s, int stride, int pad_h,
                      int pad_w, int kp, hipStream_t stream);
/* rr_conv_dgrad_s1 that ALSO returns the BatchNorm-backward sums of the layer that produced the tensor whose gradient
 * it writes (the conv -> bn -> relu layer in front of this convolution, backbones/hourglass.py:31-40): the epilogue has
 * the finished gradient dz = dx in registers, reads the producer's pre-BN output prod_y [n,h,w,c] (and prod_z, its
 * post-activation output, when the ReLU mask cannot be recomputed as prod_y*mask_scale+mask_shift > 0; with neither
 * the producer has no ReLU and every element counts) and emits per
 * block sum(dz*mask) and sum(dz*mask*xhat) to `slab` (rr_conv_stat_slab_bytes(n,h,wd,c) bytes), reduced into
 * sums [2][c] (zeroed by the caller) — exactly what rr_bn_bwd_reduce(dx, prod_z, prod_y, ...) returns, without its
 * pass over dx and y.  accumulate != 0: dx += ..., the sums are taken of the final values (the last contributor of a
 * gradient fan-in).  Layers that run split-K, and sizes with n*h*wd not a multiple of 128 (the epilogue addresses whole
 * 128-row tiles), fall back to rr_bn_bwd_reduce internally: same contract. */
int rr_conv_dgrad_s1_bnsum(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                           int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_y,
                           const float *prod_z, const float *prod_mean, const float *prod_invstd,
                           const float *prod_mask_scale, const float *prod_mask_shift, double *slab,
                           double *sums, hipStream_t stream);
/* The same idea for a producer of the form conv + bias + ReLU (the heads' 3x3 layers, detectors/centernet_detector.py:62,
 * whose output prod_z [n,h,w,c] is this convolution's input): dx = relu-masked gradient (dx * (prod_z > 0) is what is
 * STORED) and sums[0..c) = its column sums = the producer's bias gradient — what rr_bias_relu_bwd computes in a pass of
 * its own.  slab: rr_conv_stat_slab_bytes(n,h,wd,c) bytes; sums [2][c] doubles, zeroed by the caller (the second row is
 * not meaningful).  K and C multiples of 4 (the host layer zero-pads a 10- or 2-channel dy to 12 / 4); n*h*wd a multiple
 * of 128 (refused otherwise: the host layer then runs rr_conv_dgrad + rr_bias_relu_bwd).
 * accumulate != 0: dx holds the gradients of the producer's other consumers; the mask is applied to the SUM (the last
 * contributor of a fan-in: the three heads' 3x3 layers behind relu(feature), models/centernet.py:20-24), which also serves
 * a bare ReLU producer (its backward then is the identity on this tensor). */
int rr_conv_dgrad_s1_relubias(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                              int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_z, double *slab,
                              double *sums, hipStream_t stream);
int rr_conv_wgrad(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k,
                  int r, int s, int stride, int pad_h, int pad_w, int out_h, int out_w, hipStream_t stream);

/* ---- bf16-operand convolutions (BASELINE config 4, "bf16"; csrc/conv_bf16.hip) ------------------------------------- *
 * The reference is fp32-only (backbones/hourglass.py:12-61,127-199 -> nn.Conv2d), so this precision is builder-defined
 * and opt-in (cfg.Model.bf16): SAME tensors, layouts and semantics as the entry points above without the suffix — fp32
 * activations / weights / gradients in HBM — but the two operands of every product are rounded to bf16
 * (round-to-nearest-even) on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
 * Contract: result == the fp32 entry point run on bf16-rounded operands, up to the summation order.
 * Vector shapes only (C % 4 == 0, R*S <= 64, for rr_conv_wgrad_bf16 also K % 4 == 0; every tensor < 2 GiB): the host
 * layer keeps the fp32 entry points for the rest (stride-2 data gradients, the 17-tap WH head, the 3-channel stem's
 * unpacked form).  rr_conv_dgrad_s1*_bf16 take the flipped / transposed filter of rr_weight_flip_transpose (fp32). */
/* w_bf16 / wt_bf16 (may be NULL): the same filter already rounded to bf16, same [k][r][s][c] element order (the host layer
 * keeps bf16 copies of all filters, refreshed once per optimizer step: rr_weight_flip_transpose_batch).  Used when
 * c % 8 == 0: 16-byte loads of 8 channels, no converts for the B operand; identical results. */
int rr_conv_fprop_bf16(const float *x, const float *w, const float *bias, float *y, double *stat_slab,
                       int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                       int relu, const unsigned short *w_bf16, hipStream_t stream);
int rr_conv_dgrad_s1_bf16(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                          int r, int s, int pad_h, int pad_w, int accumulate, const unsigned short *wt_bf16,
                          hipStream_t stream);
int rr_conv_dgrad_s1_bnsum_bf16(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_y,
                                const float *prod_z, const float *prod_mean, const float *prod_invstd,
                                const float *prod_mask_scale, const float *prod_mask_shift, double *slab,
                                double *sums, const unsigned short *wt_bf16, hipStream_t stream);
int rr_conv_dgrad_s1_relubias_bf16(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                   int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_z,
                                   double *slab, double *sums, hipStream_t stream);
int rr_conv_wgrad_bf16(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k,
                       int r, int s, int stride, int pad_h, int pad_w, int out_h, int out_w, hipStream_t stream);

/* ---- convolutions on 16-bit ACTIVATIONS (round 5; config 4, cfg.Model.bf16; csrc/conv16.hip) ------------------------
 * Same convolutions (reference: nn.Conv2d of /root/reference/backbones/hourglass.py:12-61,127-199,
 * detectors/centernet_detector.py:80-93), same arithmetic contract as the *_bf16 entry points above (result == the fp32
 * entry point on bf16-rounded operands up to the summation order; fp32 accumulation) — but both operands are bf16 tensors
 * IN HBM: x / dy as written by their producers (rr_bn_apply_b16, rr_bn_bwd_apply_b16, rr_to_bf16), the filter's bf16
 * copy (rr_weight_flip_transpose_batch_bf16: plain for the forward, flipped / transposed for the data gradient).  They
 * reach LDS by LDS-DMA; 256 channels x 256 pixels per workgroup on v_mfma_f32_16x16x32_bf16.
 * Shapes: rr_conv16_supported (C % 64 == 0, K % 128 == 0, R*S <= 16, stride 1 or 2; input < 2 GiB); the host layer keeps
 * the *_bf16 entry points for the rest.  y / dx (fp32) and y16 / dx16 (the same values rounded to bf16, for a consumer
 * that is again a convolution): either may be NULL, not both.  stat_slab: [ceil(M / 256)][2][K] doubles
 * (rr_conv16_stat_slab_bytes), reduced by rr_bn_reduce_slab / rr_bn_stats_finalize like the fp32 kernel's. */
int rr_conv16_supported(int c, int k, int r, int s, int stride);
size_t rr_conv16_stat_slab_bytes(int n, int p, int q, int k);
int rr_conv16_fprop(const unsigned short *x, const unsigned short *w, const float *bias, float *y, unsigned short *y16,
                    double *stat_slab, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                    int relu, hipStream_t stream);
int rr_conv16_dgrad_s1(const unsigned short *dy, const unsigned short *wt, float *dx, unsigned short *dx16, int n, int h,
                       int wd, int c, int k, int r, int s, int pad_h, int pad_w, int accumulate, hipStream_t stream);
/* rr_conv16_dgrad_s1 with the backward of the ReLU in front of the convolution applied in the epilogue (rr_conv_dgrad_s1_relubias without the
 * column sums: the producer is a bare ReLU, functional._ReLU): dx = (dx_conv [+ dx]) * (relu_out > 0). */
int rr_conv16_dgrad_s1_relumask(const unsigned short *dy, const unsigned short *wt, float *dx, int n, int h, int wd, int c, int k,
                                int r, int s, int pad_h, int pad_w, int accumulate, const float *relu_out, hipStream_t stream);
/* Stride-2 data gradient (rr_conv_dgrad_s2_bf16's contract: dx [n,h,w,c] = / += the gradient of a stride-2 convolution with filter
 * w [k][r][s][c] (fp32, rounded to bf16 while its parity-class sub-filters are packed into wsub: k*r*s*c bf16 of caller scratch);
 * dy bf16 [n,p,q,k]).  K % 64 == 0, C % 128 == 0, R*S <= 16, non-negative leading pads in every class (3x3 pad 1, 1x1 pad 0). */
int rr_conv16_dgrad_s2(const unsigned short *dy, const float *w, float *dx, int n, int h, int wd, int c, int k, int r, int s,
                       int pad_h, int pad_w, int accumulate, unsigned short *wsub, hipStream_t stream);
/* dw [k][r][s][c] fp32 += x (*) dy, both bf16 (rr_conv_wgrad's contract on bf16-rounded operands; fp32 atomics).
 * Shapes: rr_conv16_wgrad_supported (K % 128 == 0 — the last 256-filter tile may run half empty —, C % 128 == 0, stride 1 or 2; tensors < 2 GiB). */
int rr_conv16_wgrad_supported(int c, int k, int r, int s, int stride);
int rr_conv16_wgrad(const unsigned short *x, const unsigned short *dy, float *dw, int n, int h, int wd, int c, int k,
                    int r, int s, int stride, int pad_h, int pad_w, hipStream_t stream);

/* Producers of the bf16 images (csrc/elementwise.hip): rr_bn_apply / rr_bn_bwd_apply (reference: nn.BatchNorm2d + ReLU + residual
 * add of backbones/hourglass.py:31-40 and their autograd backward) with a second, bf16 output holding the same values
 * rounded to nearest even — written in the same pass, so the convolution that consumes the tensor never reads the fp32
 * one.  out / dx (fp32) may be NULL when nothing else reads them.  rr_to_bf16: the plain conversion, for operands that
 * come from elsewhere (fan-in sums, the up-sample add, head gradients). */
int rr_bn_apply_b16(const float *y, const unsigned short *y16, const float *scale, const float *shift, const float *res,
                    const unsigned short *res16, const float *res_scale, const float *res_shift, float *out,
                    unsigned short *out16, long total, int c, int relu, hipStream_t stream);
int rr_bn_bwd_apply_b16(const float *dz, const float *z, const unsigned short *z16, const float *y, const unsigned short *y16,
                        const float *mean, const float *invstd, const float *gamma, const float *mask_scale,
                        const float *mask_shift, const double *sums, double count, const double *count_dev, float *dx,
                        unsigned short *dx16, float *g_out, int g_accumulate, float *dgamma, float *dbeta, long total,
                        int c, hipStream_t stream);
/* y16 / res16 / z16: the convolution's pre-BN output / the residual / the layer's own output (the ReLU mask's source) when
 * it exists ONLY as its bf16 image (a tensor every consumer of which reads bf16: rrnet_amd.ops.phantom_f32); give y or
 * y16 (exactly one), res or res16, z or z16 (at most one each).
 * rr_bn_bwd_reduce_b16: rr_bn_bwd_reduce with z and / or y from their images (sums pre-zeroed).  rr_upsample2x_add_b16: the hourglass
 * up-path add (backbones/hourglass.py:121-124, even sizes) with either operand as fp32 or bf16 and the result as fp32
 * and / or bf16.  rr_from_bf16: the widening copy (a consumer without a bf16 form materialises the fp32 tensor). */
int rr_bn_bwd_reduce_b16(const float *dz, const float *z, const unsigned short *z16, const float *y, const unsigned short *y16,
                         const float *mean, const float *invstd, const float *mask_scale, const float *mask_shift,
                         double *sums, long npix, int c, hipStream_t stream);
int rr_upsample2x_add_b16(const float *up1, const unsigned short *up1_16, const float *low, const unsigned short *low_16,
                          float *out, unsigned short *out16, int n, int h, int w, int c, hipStream_t stream);
int rr_from_bf16(const unsigned short *x, float *out, long total, hipStream_t stream);
int rr_to_bf16(const float *x, unsigned short *out, long total, hipStream_t stream);

/* Data gradient of a head's narrow 1x1 convolution (K = 10 / 2 / 34 output channels behind a 3x3 conv + bias + ReLU:
 * /root/reference/detectors/centernet_detector.py:62,73,85-93), the producer's ReLU mask and bias gradient in the same
 * pass — rr_conv_dgrad_s1_relubias's contract for r = s = 1 without the channel padding:
 *   dx[m][c] = (prod_z[m][c] > 0) * ([dx[m][c] +] sum_k dy[m][k] * w[k][c]);  sums[c] += sum_m dx[m][c]  (doubles, pre-zeroed).
 * dy [m][k], w [k][c] (the layer's own OHWI filter), k <= 36, c a divisor of 1024; dx16 (may be NULL): bf16 image of dx.
 * HBM-bound element-wise kernel (csrc/elementwise.hip). */
int rr_head_dgrad_relubias(const float *dy, const float *w, float *dx, unsigned short *dx16, const float *prod_z,
                           double *sums, long m, int c, int k, int accumulate, hipStream_t stream);

/* ---- split-operand convolutions ("f16x3", cfg.Model.conv_math; csrc/conv_bf16.hip) -------------------------------
 * fp32 in, fp32 out, fp32 accumulation, as rr_conv_fprop / rr_conv_dgrad / rr_conv_wgrad (reference: nn.Conv2d in
 * /root/reference/backbones/hourglass.py:12-61); inside the kernel each operand is the sum of two fp16 values (22
 * significant bits, after a power-of-two scaling taken from the tensor's largest magnitude) and the three products
 * hi*hi + hi*lo + lo*hi run on the 16-bit matrix instructions.  amax_*: DEVICE words holding the bit pattern of
 * max|tensor|, produced by rr_absmax_bits (atomicMax into a word the caller zeroed; several calls may share a word to
 * take the maximum over several tensors).  Shapes as the _bf16 entry points.  w_split / wt_split (optional; C % 8 == 0,
 * K > 64): the filter (for the data gradients: the flipped / transposed one) already split into its two fp16 parts by
 * rr_weight_split_f16 with the SAME amax word — 2*k*r*s*c halves, hi image then lo image — instead of tile by tile inside
 * the convolution (48 of its 149 vector instructions per K-step and thread). */
int rr_weight_split_f16(const float *w, long n, const unsigned *amax_w, unsigned short *out, hipStream_t stream);
int rr_absmax_bits(const float *x, long n, unsigned *out, hipStream_t stream);
int rr_conv_fprop_f16x3(const float *x, const float *w, const float *bias, float *y, double *stat_slab,
                        int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h,
                        int pad_w, int relu, const unsigned *amax_x, const unsigned *amax_w,
                        const unsigned short *w_split, hipStream_t stream);
int rr_conv_dgrad_s1_f16x3(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                           int r, int s, int pad_h, int pad_w, int accumulate, const unsigned *amax_dy,
                           const unsigned *amax_w, const unsigned short *wt_split, hipStream_t stream);
int rr_conv_dgrad_s1_bnsum_f16x3(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                 int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_y,
                                 const float *prod_z, const float *prod_mean, const float *prod_invstd,
                                 const float *prod_mask_scale, const float *prod_mask_shift, double *slab,
                                 double *sums, const unsigned *amax_dy, const unsigned *amax_w,
                                 const unsigned short *wt_split, hipStream_t stream);
int rr_conv_dgrad_s1_relubias_f16x3(const float *dy, const float *wt, float *dx, int n, int h, int wd, int c, int k,
                                    int r, int s, int pad_h, int pad_w, int accumulate, const float *prod_z,
                                    double *slab, double *sums, const unsigned *amax_dy, const unsigned *amax_w,
                                    hipStream_t stream);
int rr_conv_dgrad_s2_f16x3(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k,
                           int r, int s, int pad_h, int pad_w, int accumulate, float *wsub, const unsigned *amax_dy,
                           const unsigned *amax_w, hipStream_t stream);
int rr_conv_wgrad_f16x3(const float *x, const float *dy, float *dw, int n, int h, int wd, int c, int k,
                        int r, int s, int stride, int pad_h, int pad_w, int out_h, int out_w, const unsigned *amax_x,
                        const unsigned *amax_dy, hipStream_t stream);

/* Stride-2 data gradient (rr_conv_dgrad with stride 2) on the bf16 forward kernel: one launch per output parity class
 * (h % 2, w % 2) with that class's sub-filter (1 / 2 / 2 / 4 taps of a 3x3), written to every second pixel of dx
 * [n,h,w,c]; classes no tap reaches are zeroed (unless accumulating).  w OHWI fp32; wsub: k*r*s*c floats of caller
 * scratch (the packed sub-filters).  C, K multiples of 4. */
int rr_conv_dgrad_s2_bf16(const float *dy, const float *w, float *dx, int n, int h, int wd, int c, int k,
                          int r, int s, int pad_h, int pad_w, int accumulate, float *wsub, hipStream_t stream);

/* ---- BatchNorm / ReLU / residual / up-path / Adam (HBM-bound NHWC elementwise) -------- *
 * Replace nn.BatchNorm2d (SyncBatchNorm via operators/rrnet_operator.py:27) + ReLU + residual
 * add of backbones/hourglass.py:18-19,22,26,34-40,51,59-60 and backbones/resnet.py:23-50;
 * nn.Upsample x2 + bilinear(align_corners) resize + add of backbones/hourglass.py:113,121-124;
 * F.adaptive_avg_pool2d of detectors/fasterrcnn_detector.py:15; optim.Adam of
 * operators/rrnet_operator.py:29,138.  `total` = element count, `c` = channels (NHWC inner).
 * Training statistics: rr_conv_fprop's slab -> rr_bn_reduce_slab (adds into a zeroed) sums[2][c] (the SyncBN
 * exchange all-reduces exactly this buffer plus the sample count) -> rr_bn_finalize ->
 * mean/invstd (saved for backward), scale/shift (for rr_bn_apply), running stats updated with
 * `momentum` and the unbiased variance, num_batches_tracked (optional int64 counter) incremented.
 * rr_bn_apply: out = relu?(y*scale+shift [+ res | + res*res_scale+res_shift]).
 * rr_bn_bwd_reduce: sums[2][c] = per-channel sum(dy), sum(dy*xhat), dy = dz*(z>0) if z; with z NULL and
 *   mask_scale/mask_shift given the ReLU mask is recomputed as (y*scale+shift > 0) — layers without a
 *   residual input need not re-read their output.  sums_zeroed != 0: the caller hands in zeroed sums (the host
 *   layer takes them from one pre-zeroed pool: 163 memset launches per step less).
 * rr_bn_bwd_apply: dx = gamma*invstd*(dy - sums0/count - xhat*sums1/count); g_out (optional)
 *   receives dy for the residual branch; dgamma/dbeta (optional) are accumulated from sums.
 *   `count_dev` (optional, both finalize and bwd_apply): sample count read from device memory
 *   instead of `count` — the SyncBN exchange all-reduces it next to the sums.
 * rr_sum_n: out = (sum of n<=8 tensors) * (z>0 if z): gradient fan-in of a multiply-used tensor.
 * rr_bias_relu_bwd: dy_masked = dy*(z>0) (if z), dbias += column sums (conv+bias+ReLU heads,
 *   detectors/centernet_detector.py:85-93).
 * rr_upsample_add: out[n,h,w] = up1[n,h,w] + bilinear_ac(nearest2x(low))[h,w]; the exact 2x
 *   case never builds the intermediate image. */
int rr_bn_reduce_slab(const double *slab, int mtiles, int c, double *sums, hipStream_t stream);
int rr_bn_finalize(const double *sums, double count, const double *count_dev, const float *gamma, const float *beta,
                   float *running_mean, float *running_var, float momentum, float eps, float *mean,
                   float *invstd, float *scale, float *shift, int c, long *num_batches_tracked, hipStream_t stream);
/* SyncBN (data parallel; /root/reference/operators/distributed_wrapper.py:40-61 converts every BatchNorm): the three small steps
 * around a statistics exchange without extra launches.  rr_bn_reduce_slab_count = rr_bn_reduce_slab that also stores the local
 * sample count into *count_slot (a slot of the buffer that goes through the all-reduce);  rr_bn_finalize_count = rr_bn_finalize
 * with the (exchanged) count read from the device and copied to *count_out for the backward;  rr_bn_affine_grad adds the LOCAL
 * BatchNorm-backward sums into dbeta / dgamma before they are exchanged. */
int rr_bn_reduce_slab_count(const double *slab, int mtiles, int c, double *sums, double count, double *count_slot,
                            hipStream_t stream);
int rr_bn_finalize_count(const double *sums, const double *count_dev, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, float momentum, float eps, float *mean,
                         float *invstd, float *scale, float *shift, int c, long *num_batches_tracked,
                         double *count_out, hipStream_t stream);
int rr_bn_affine_grad(const double *sums, float *dgamma, float *dbeta, int c, hipStream_t stream);
/* rr_bn_reduce_slab + rr_bn_finalize in one launch, for the single-process case (no SyncBN exchange of the sums in
 * between); fixed summation order (no atomics). */
int rr_bn_stats_finalize(const double *slab, int mtiles, double count, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, float momentum, float eps, float *mean,
                         float *invstd, float *scale, float *shift, int c, long *num_batches_tracked,
                         hipStream_t stream);
int rr_bn_eval_coeffs(const float *gamma, const float *beta, const float *running_mean,
                      const float *running_var, float eps, float *scale, float *shift, int c,
                      hipStream_t stream);
int rr_bn_apply(const float *y, const float *scale, const float *shift, const float *res,
                const float *res_scale, const float *res_shift, float *out, long total, int c, int relu,
                hipStream_t stream);
/* rr_bn_apply that also leaves max |out| (bit pattern, atomicMax into the zeroed word *amax_out): the operand scale of
 * the split-operand convolution that consumes `out` (rr_conv_*_f16x3) without a pass of rr_absmax_bits. */
int rr_bn_apply_amax(const float *y, const float *scale, const float *shift, const float *res,
                     const float *res_scale, const float *res_shift, float *out, long total, int c, int relu,
                     unsigned *amax_out, hipStream_t stream);
int rr_bn_bwd_reduce(const float *dz, const float *z, const float *y, const float *mean,
                     const float *invstd, const float *mask_scale, const float *mask_shift, double *sums,
                     long npix, int c, int sums_zeroed, hipStream_t stream);
int rr_bn_bwd_apply(const float *dz, const float *z, const float *y, const float *mean,
                    const float *invstd, const float *gamma, const float *mask_scale, const float *mask_shift,
                    const double *sums, double count,
                    const double *count_dev, float *dx, float *g_out, float *dgamma, float *dbeta, long total,
                    int c, hipStream_t stream);
/* rr_bn_bwd_apply whose masked gradient (the residual branch's share, g) is ADDED into g_acc — the fan-in buffer of the
 * residual's fan-out when another consumer already left its gradient there (one read-modify-write instead of a write,
 * an add kernel and its three passes). */
int rr_bn_bwd_apply_gacc(const float *dz, const float *z, const float *y, const float *mean,
                    const float *invstd, const float *gamma, const float *mask_scale, const float *mask_shift,
                    const double *sums, double count,
                    const double *count_dev, float *dx, float *g_acc, float *dgamma, float *dbeta, long total,
                    int c, hipStream_t stream);
/* rr_bn_bwd_apply / rr_bn_bwd_apply_gacc (g_accumulate) that also leave max |dx| in the zeroed word *amax_dx (see
 * rr_bn_apply_amax): dx is the operand of the data / weight gradients of the convolution in front of the BatchNorm. */
int rr_bn_bwd_apply_amax(const float *dz, const float *z, const float *y, const float *mean,
                         const float *invstd, const float *gamma, const float *mask_scale, const float *mask_shift,
                         const double *sums, double count, const double *count_dev, float *dx, float *g_out,
                         int g_accumulate, float *dgamma, float *dbeta, long total, int c, unsigned *amax_dx,
                         hipStream_t stream);
int rr_relu_fwd(const float *x, float *out, long total, hipStream_t stream);
/* [npix][k] -> [npix][kp], channels k..kp-1 zero (kp a multiple of 4): the 10- / 2-channel gradients of the hm / offset
 * heads' last 1x1 convolutions (detectors/centernet_detector.py:14-16) enter the vector data-gradient kernel as 12 / 4
 * channels.  (A strided torch copy into a channel slice took 3.6 ms per head at 8x256x256 — 10.8 ms of a 457 ms step.) */
int rr_pad_channels(const float *src, float *dst, long npix, int k, int kp, hipStream_t stream);
int rr_sum_n(const float *const *grads, int n, const float *z, float *out, long total, hipStream_t stream);
int rr_bias_relu_bwd(const float *dy, const float *z, float *dy_masked, float *dbias, long npix, int c,
                     hipStream_t stream);
int rr_upsample_add_fwd(const float *up1, const float *low, float *out, int n, int h, int w, int lh, int lw,
                        int c, hipStream_t stream);
int rr_upsample_add_bwd(const float *dout, float *dlow, int n, int h, int w, int lh, int lw, int c,
                        hipStream_t stream);
/* Bilinear resize with align_corners = True (operators/rrnet_operator.py:263, the multi-scale evaluation's
 * F.interpolate); x NHWC [n,h,w,c] -> out NHWC [n,oh,ow,c]. */
int rr_resize_bilinear_ac(const float *x, float *out, int n, int h, int w, int oh, int ow, int c, hipStream_t stream);
int rr_avgpool_fwd(const float *x, float *out, long r, int hw, int c, hipStream_t stream);
int rr_avgpool_bwd(const float *dout, float *dx, long r, int hw, int c, hipStream_t stream);
/* Inference tail of the stage-2 head (backbones/resnet.py:48-53 + detectors/fasterrcnn_detector.py:15):
 * out[r,c] = mean over the hw positions of relu(y[r,p,c]*scale[c] + shift[c] + res[r,p,c]); y, res NHWC [r,hw,c]. */
int rr_bn_res_relu_avgpool(const float *y, const float *scale, const float *shift, const float *res, float *out,
                           long r, int hw, int c, hipStream_t stream);
/* 1x1 convolution to at most 64 output channels on many rows (the stage-2 head's conv1, backbones/resnet.py:33-35 through
 * detectors/fasterrcnn_detector.py:17-18, at inference on R*9 RoI rows): y[m][n] = relu?(x[m][:] . w[n][:] + bias[n]), x [M][K]
 * (K = 128 or 256, M*K*4 < 2 GiB), w [N][K], y [M][N].  Weights live in registers, rows stream through persistent workgroups. */
int rr_conv1x1_rows(const float *x, const float *w, const float *bias, float *y, long m, int k, int n, int relu,
                    hipStream_t stream);
/* The same tail with the 1x1 convolution in front of it fused in (inference): out [r, n] = mean over the hw rows of a
 * RoI of relu((h [r*hw, k] x w [n, k]^T) * scale + shift + res [r*hw, n]); the convolution's output never reaches
 * HBM.  k = 32 or 64, n a multiple of 4 and <= 256 (backbones/resnet.py:46-53 with planes = 64,
 * fasterrcnn_detector.py:15). */
int rr_conv1x1_bn_res_relu_avgpool(const float *h, const float *w, const float *scale, const float *shift,
                                   const float *res, float *out, long r, int hw, int k, int n, hipStream_t stream);
/* WH head, detectors/centernet_detector.py:26-77 (HCov k x 1 and WCov 1 x k to one channel each,
 * interleaved [W,H]): t [n,h,w,ct] (ct >= 2k, a multiple of 4 keeps the vector conv paths) = 1x1
 * convolution of the 256-channel map with the 2k tap vectors (rows 0..k-1 = HCov taps, k..2k-1 = WCov
 * taps, the rest zero; computed by rr_conv_fprop); out [n,h,w,2]:
 * out[..,0] = bias_w + sum_s t[h, w+s-k/2, k+s], out[..,1] = bias_h + sum_r t[h+r-k/2, w, r]. */
int rr_wh_shift_sum_fwd(const float *t, const float *bias_w, const float *bias_h, float *out, int n, int h,
                        int w, int k, int ct, hipStream_t stream);
int rr_wh_shift_sum_bwd(const float *dout, float *dt, int n, int h, int w, int k, int ct, hipStream_t stream);
int rr_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                 float beta1, float beta2, float eps, int step, float grad_scale, hipStream_t stream);

/* ---- losses --------------------------------------------------------------------------- *
 * rr_focal_loss_fwd/bwd: clamp(sigmoid(x),1e-4,1-1e-4) + focal_loss_for_hm of
 *   operators/rrnet_operator.py:55-57 + modules/loss/functional.py:25-51.  logits and gt share
 *   one layout; sums[3] (double, device) = { sum log(p)(1-p)^2 [g==1], sum log(1-p)p^2(1-g)^4
 *   [g<1], #(g==1) }; loss = -(s0+s1)/s2, or -s1 when s2 == 0.  bwd: dlogits = *gout * gscale *
 *   dloss/dlogits (gout: device scalar).
 * rr_regl1_fwd/bwd: modules/loss/regl1loss.py:9-17.  pred NHWC [b,hw,c]; mask, ind [b,m] float;
 *   target [b,m,c]; sums = { sum|pred*m - t*m|, sum of the expanded mask }; loss = s0/(s1+1e-4).
 *   bwd zeroes dpred [b,hw,c] and scatters with float atomics.
 * rr_stage2_loss: operators/rrnet_operator.py:63-102 — per RoI max IoU (torchvision box_iou
 *   definition) against gt [b,g,gstride>=4] xyxy, positives IoU > 0.5, Faster-RCNN targets with
 *   the +1 width convention, smooth-L1 mean per image / b; images without positives add 0.
 *   rois [r,5] = (image, x1,y1,x2,y2) in feature coords, scaled by `scale`.  Outputs: tgt [r,4],
 *   pos [r], npos [b], loss[0] (double), dreg_unit [r,4] = dloss/dreg, droi_unit [r,4] (optional) =
 *   dloss/droi — the reference's smooth-L1 also differentiates its TARGETS, which depend on the boxes
 *   (rrnet_operator.py:82 with the hard-NMS-selected boxes of models/rrnet.py:70). */
int rr_focal_loss_fwd(const float *logits, const float *gt, long n, double *sums, hipStream_t stream);
int rr_focal_loss_bwd(const float *logits, const float *gt, long n, const double *sums, const float *gout,
                      float gscale, float *dlogits, hipStream_t stream);
int rr_regl1_fwd(const float *pred, const float *mask, const float *ind, const float *target, int b, int m,
                 int c, long hw, double *sums, hipStream_t stream);
int rr_regl1_bwd(const float *pred, const float *mask, const float *ind, const float *target, int b, int m,
                 int c, long hw, const double *sums, const float *gout, float gscale, float *dpred,
                 hipStream_t stream);
int rr_stage2_loss(const float *rois, const float *reg, int r, const float *gt, int b, int g, int gstride,
                   float scale, float *tgt, int *pos, int *npos, double *loss, float *dreg_unit,
                   float *droi_unit, hipStream_t stream);

/* ---- decode / NMS plumbing ------------------------------------------------------------ *
 * rr_decode_topk: models/rrnet.py:93-138 (_topk + gathers + transform_bbox).  hm NHWC
 *   [b,h,w,c] logits (is_logits=1: sigmoid applied) or ready scores (0); wh, off NHWC [b,h,w,2];
 *   out [b,k,6] = x1,y1,x2,y2,score,cls in feature coordinates, score-descending, ties by the
 *   reference's flat index; pix_out (optional) [b,k] = y*w+x of every row.  rr_roi_provenance maps
 *   packed RoIs back to that pixel; rr_proposal_bwd is the backward of the box assembly (d wh, d off maps
 *   from d roi).  k <= min(4096, c*h*w).  peak_filter=1 applies operators/centernet_operator.py:204-210
 *   (`_ctnet_nms`: keep a score only where it equals its 3x3 window maximum, dead code in the reference, named by
 *   north_star) INSIDE the scan: only the candidates above the sampled threshold are tested, no extra pass over
 *   the map; rr_peak3x3 writes the filtered score map itself (is_logits=0: hm holds
 *   ready scores, the reference's call form `_ctnet_nms(heat)`).  workspace (optional, rr_decode_workspace_bytes(b)
 *   bytes of device memory): with it, maps of >= 64 K elements are scanned by all CUs (threshold kernel -> streaming
 *   candidate scan -> one workgroup per frame for sort + box assembly); without it one workgroup per frame does
 *   everything.  Both give identical rows.  box_mode 0: RRNet rows as above (scale unused); box_mode 1: CenterNet
 *   rows of operators/centernet_operator.py:152-178 = (x, y, w, h) * scale, score, cls+1, wh NOT clamped.
 * rr_group_by_class: stable regrouping of each image's k rows by class (classes ascending =
 *   torch.unique order of models/rrnet.py:59); seg_off [b*num_classes+1] row offsets.  Rows whose class lies
 *   outside [cls_base, cls_base+num_classes) are dropped and leave a gap at the end of the image's k-row block, so
 *   seg_off[s+1]-seg_off[s] over-counts the image's last class by the gap: seg_len (optional, [b*num_classes]) holds
 *   the exact lengths and is what the NMS entries should be given (rr_hard_nms_segments, rr_soft_nms_ragged).
 * rr_hard_nms_segments: torchvision.ops.nms as called at models/rrnet.py:69,78; rows of a
 *   segment score-descending, 6 floats per row; kept rows are compacted to the segment front.  seg_len NULL:
 *   lengths from consecutive offsets.
 * rr_pack_segments: phase 0 -> out_off [nseg+1] exclusive prefix of n_out (out_off[nseg] = R);
 *   phase 1 -> rois [R,5], scores [R], clses [R] (models/rrnet.py:37-49) and/or rows6 [R,6]. */
size_t rr_decode_workspace_bytes(int b);
int rr_decode_topk(const float *hm, int is_logits, int peak_filter, const float *wh, const float *off, int b,
                   int h, int w, int c, int k, int box_mode, float scale, float *out, int *pix_out, void *workspace,
                   size_t workspace_bytes, hipStream_t stream);
int rr_roi_provenance(const float *rois, const float *scores, const float *clses, int r, const float *decoded,
                      const int *pix, int k, int *roi_pix, hipStream_t stream);
int rr_proposal_bwd(const float *droi, const float *rois, const int *roi_pix, int r, const float *wh, int b, int h,
                    int w, float *dwh, float *doff, hipStream_t stream);
int rr_peak3x3(const float *hm, int is_logits, float *scores, int b, int h, int w, int c, hipStream_t stream);
int rr_group_by_class(const float *boxes, int b, int k, int num_classes, int cls_base, float *grouped,
                      int *seg_off, int *seg_len, hipStream_t stream);
int rr_hard_nms_segments(float *boxes, const int *seg_off, const int *seg_len, int nseg, int max_seg_boxes,
                         float thresh, int *n_out, hipStream_t stream);

/* ---- the reference's own hard-NMS family: ext/nms/nms/nms_kernel.cu (`_nms`), cpu_nms.pyx:129-176, py_cpu_nms.py,
 * bound by ext/nms/nms_wrapper.py:23-33 `nms(dets, thresh, gpu_id)`.  Legacy "+1" IoU; boxes [n,stride>=4] on the
 * device, already score-descending; inclusive = 0 suppresses at IoU > thresh (nms_kernel.cu:71, py_cpu_nms.py:29),
 * 1 at IoU >= thresh (cpu_nms.pyx:170).  keep [n] (device) receives the kept row indices in order, *num_out their
 * count; workspace = rr_nms_workspace_bytes(n) bytes (the n x ceil(n/64) suppression bit matrix).
 * `_nms` is the reference's C entry itself (gpu_nms.hpp:1-2): host pointers, synchronous, own scratch. */
size_t rr_nms_workspace_bytes(int n);
int rr_nms_sorted(const float *boxes, int n, int stride, float thresh, int inclusive, void *workspace,
                  int *keep, int *num_out, hipStream_t stream);
void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
          float nms_overlap_thresh, int device_id);
int rr_pack_segments(const float *grouped, const int *seg_off, const int *n_out, int nseg, int segs_per_image,
                     int *out_off, float *rois, float *scores, float *clses, float *rows6, int phase,
                     hipStream_t stream);

/* ---- training targets ------------------------------------------------------------ *
 * rr_ctnet_targets: datasets/transforms/functional.py:177-262 (gaussian_radius, gaussian2d, draw_umich_gaussian,
 *   to_heatmap) + the zero padding of datasets/drones_det.py:70-94 (collate_fn_ctnet) for a whole batch.
 *   annos [b,m,anno_stride>=6] = x,y,w,h,score,cls(1-based),.. in image pixels (rows >= counts[i] are padding),
 *   counts [b] int32.  Outputs: hm NHWC [b,img_h/sf,img_w/sf,num_classes] (zeroed here), wh [b,m,2] = (w,h)/sf,
 *   ind [b,m] = cy_int*(img_w/4)+cx_int as float, offset [b,m,2], reg_mask [b,m] (0/1 floats). */
int rr_ctnet_targets(const float *annos, const int *counts, int b, int m, int anno_stride, int img_h, int img_w,
                     int scale_factor, int num_classes, float *hm, float *wh, float *ind, float *offset,
                     float *reg_mask, hipStream_t stream);

/* ---- inference post-process (config 5: decode -> re-regression -> Soft-NMS) ------------------ *
 * rr_refine_boxes: operators/rrnet_operator.py:188-209 `generate_bbox` (stage-2 boxes from the packed RoIs
 *   [r,5] = image,x1,y1,x2,y2 in feature coordinates, the regression [r,4], scores, classes), the score filter
 *   `pred_bbox[:, 4] > score_thr` (:266-267) and the xywh -> xyxy step of `_ext_nms` (:222-223), for all
 *   (frame, class) segments of a batch in one launch.  seg_off [nseg+1] = row offsets of the stage-1 segments in
 *   the packed list (rr_pack_segments phase 0).  out6 rows = x1,y1,x2,y2,score,cls+1, order preserved, written
 *   to the front of each segment's row range; seg_len [nseg] = rows kept.  Feed to rr_soft_nms_ragged.
 * rr_finalize_frames: `np.concatenate` of the per-class results (:225), xyxy -> xywh (:231) and the final
 *   `torch.sort(score, descending)` (:278-279; ties keep concatenation order) per frame.  out_off [nseg+1] =
 *   exclusive prefix of n_out (rr_pack_segments phase 0); frame f's rows land at out6[out_off[f*segs_per_frame]]. */
int rr_refine_boxes(const float *rois, const float *reg, const float *scores, const float *clses,
                    const int *seg_off, int nseg, float scale, float score_thr, float *out6, int *seg_len,
                    hipStream_t stream);
int rr_finalize_frames(const float *boxes6, const int *seg_off, const int *n_out, const int *out_off,
                       int nframes, int segs_per_frame, int max_frame_boxes, float *out6, hipStream_t stream);
/* rows6 [n,6] -> out6 ordered by score (column 4) descending, ties in input order (operators/rrnet_operator.py:272,
 * 278: the torch.sort calls around _ext_nms in the multi-scale evaluation); n <= 16384, out6 != rows6. */
int rr_sort_rows_by_score(const float *rows6, int n, float *out6, hipStream_t stream);

/* ---- RoIAlign --------------------------------------------------------------------------- *
 * torchvision.ops.roi_align(feat, rois, (ph,pw)) at models/rrnet.py:51 (spatial_scale 1,
 * sampling_ratio -1, legacy coordinates).  feat NHWC [b,h,w,c]; out NHWC [r,ph,pw,c]. */
int rr_roi_align_fwd(const float *feat, const float *rois, int r, int h, int w, int c, int ph, int pw,
                     float spatial_scale, int sampling_ratio, const int *order, float *out, hipStream_t stream);
/* order (optional, 3x3 bins): processing position -> RoI index; rr_roi_spatial_order builds it per frame (rois of
 * frame f = rows [frame_off[f], frame_off[f+1]) of the packed list) by (8-pixel row band, x centre), so that RoIs with
 * overlapping footprints run back to back on one XCD and share its L2.  Outputs stay in RoI order. */
int rr_roi_spatial_order(const float *rois, const int *frame_off, int nframes, int *order, hipStream_t stream);
int rr_roi_align_bwd(const float *dout, const float *rois, int r, int b, int h, int w, int c, int ph, int pw,
                     float spatial_scale, int sampling_ratio, float *dfeat, hipStream_t stream);

/* ---- Modulated deformable convolution (DCNv2) -- BASELINE config 4 ------------------------- *
 * Replaces ext/dcn: `dcn_v2_forward` / `dcn_v2_backward` of src/vision.cpp:3-8 (src/cuda/dcn_v2_cuda.cu:42-172,
 * 206-335, kernels src/cuda/dcn_v2_im2col_cuda.cu:125-327), bound by ext/dcn/dcn_v2.py:25,40.
 * x NHWC [n,h,w,c]; offset NHWC [n,p,q,2*dg*r*s] (per deformable group: interleaved (dh,dw) per tap);
 * mask NHWC [n,p,q,dg*r*s]; w OHWI [k][r][s][c]; y NHWC [n,p,q,k].
 * rr_dcn_fwd is one fused gather-GEMM (no column buffer).  The backward is assembled by the host layer:
 *   col  = rr_dcn_im2col(x, offset, mask)            [M, r*s*c], M = n*p*q   (rr_dcn_col_bytes)
 *   dw  += rr_conv_wgrad(col as [1,M,1,r*s*c], dy as [1,M,1,k])              (1x1 layer)
 *   dcol = rr_conv_dgrad(dy, w as [k, r*s*c, 1, 1])
 *   rr_dcn_col2im(x, offset, mask, dcol) -> dx (zeroed + float atomics), doffset, dmask
 * Requires c % 4 == 0, c % dg == 0 and (dg == 1 or (c/dg) % 32 == 0).
 * Layers with stride 1, c % 32 == 0 and k > 32 (3x3 for the two gradients) run on kernels that stage the input block an
 * 8x16 pixel tile can reach in LDS (margin RR_DCN_WINDOW, default 3 pixels; offsets beyond it go to global memory);
 * the others on the kernels that gather the bilinear corners from L2.  Same results either way. */
int rr_dcn_fwd(const float *x, const float *offset, const float *mask, const float *w, const float *bias, float *y,
               int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
               int deformable_groups, hipStream_t stream);
/* Same operation with bf16 matrix operands (samples and weights rounded to bf16 in the kernel, fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16): inputs and outputs stay fp32 tensors.  BASELINE config 4. */
int rr_dcn_fwd_bf16(const float *x, const float *offset, const float *mask, const float *w, const float *bias, float *y,
                    int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                    int deformable_groups, hipStream_t stream);
/* rr_dcn_fwd_bf16 with caller scratch for the weights re-packed to bf16 [tap][32-channel chunk][filter][32]
 * (rr_dcn_wpack_bytes(c,k,r,s) bytes, filled inside by one small kernel per call): on the LDS-window kernel with 256-filter
 * tiles the weight operand then goes global -> LDS by buffer_load ... lds (no staging registers, converts or ds_writes).
 * Layers that kernel does not take run exactly as rr_dcn_fwd_bf16. */
size_t rr_dcn_wpack_bytes(int c, int k, int r, int s);
int rr_dcn_fwd_bf16_packed(const float *x, const float *offset, const float *mask, const float *w, const float *bias,
                           float *y, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                           int dilation, int deformable_groups, void *wpack, hipStream_t stream);
size_t rr_dcn_col_bytes(int n, int h, int wd, int c, int r, int s, int stride, int pad_h, int pad_w, int dilation);
int rr_dcn_im2col(const float *x, const float *offset, const float *mask, float *col, int n, int h, int wd, int c,
                  int r, int s, int stride, int pad_h, int pad_w, int dilation, int deformable_groups,
                  hipStream_t stream);
int rr_dcn_col2im(const float *x, const float *offset, const float *mask, const float *dcol, float *dx,
                  float *doffset, float *dmask, int n, int h, int wd, int c, int r, int s, int stride, int pad_h,
                  int pad_w, int dilation, int deformable_groups, hipStream_t stream);

/* Fused backward (no column buffers): rr_dcn_wgrad adds dY^T x (deformed columns produced in registers) into dw
 * (float atomics; dw pre-zeroed or holding the running gradient); rr_dcn_dgrad keeps the column gradient in the MFMA
 * accumulators and writes dx (zeroed inside; window kernels: contributions pre-summed per pixel block in an LDS image
 * in fixed point, one power-of-two scale per block and channel, then one global float atomic per window element;
 * otherwise float atomics on the bilinear corners), doffset and dmask (plain stores).
 * Replace ext/dcn/src/cuda/dcn_v2_cuda.cu:206-335 + dcn_v2_im2col_cuda.cu:197-327.  Layers they do not take
 * (rr_dcn_fused_bwd_supported == 0) go through the column path above. */
/* rr_dcn_wgrad_bf16 with dY's bf16 image (same element order as dy; its producer's image or one rr_to_bf16 pass): the dY
 * operand reaches LDS by buffer_load ... lds, no conversion inside (round 5).  Same results as rr_dcn_wgrad_bf16.
 * Replaces the dW GEMM of dcn_v2_cuda_backward (ext/dcn/src/cuda/dcn_v2_cuda.cu:283-301). */
int rr_dcn_wgrad_bf16_img(const float *x, const float *offset, const float *mask, const float *dy, const unsigned short *dy_bf16,
                          float *dw, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w,
                          int dilation, int deformable_groups, hipStream_t stream);

/* 1 when rr_dcn_wgrad / rr_dcn_dgrad (and their _bf16 forms) take a layer of this shape, 0 when the host layer has to
 * run the column path (rr_dcn_im2col / rr_dcn_col2im + the conv GEMMs). */
int rr_dcn_fused_bwd_supported(int c, int k, int r, int s, int stride, int deformable_groups);      /* dilation 1 */
/* The same query with the dilation: deformable groups of 32 / 64 / 96 channels exist on the LDS-window kernels only, and a
 * dilated filter's window may not fit (3x3 with dilation >= 3: the data-gradient window exceeds 160 KB). */
int rr_dcn_fused_bwd_supported_dil(int c, int k, int r, int s, int stride, int dilation, int deformable_groups);
int rr_dcn_wgrad(const float *x, const float *offset, const float *mask, const float *dy, float *dw, int n, int h,
                 int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                 int deformable_groups, hipStream_t stream);
int rr_dcn_dgrad(const float *x, const float *offset, const float *mask, const float *w, const float *dy, float *dx,
                 float *doffset, float *dmask, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h,
                 int pad_w, int dilation, int deformable_groups, hipStream_t stream);
/* rr_dcn_dgrad_bf16 with caller scratch for dY rounded to bf16 (rr_dcn_dyb_bytes(n,p,q,k) bytes, written inside by one
 * pass per call): the K sweep re-reads a block's dY tile once per 32-channel chunk — as bf16 those re-reads stay in L2
 * (64 KB per workgroup instead of 128 KB) and the operand staging needs no convert.  Same results bit for bit. */
size_t rr_dcn_dyb_bytes(int n, int p, int q, int k);
int rr_dcn_dgrad_bf16_ws(const float *x, const float *offset, const float *mask, const float *w, const float *dy, float *dx,
                         float *doffset, float *dmask, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad_h,
                         int pad_w, int dilation, int deformable_groups, void *dyb, hipStream_t stream);
/* ... with dY's bf16 image already in HBM (written by dY's producer: rr_head_dgrad_relubias / rr_to_bf16): no conversion
 * inside the call.  dy (fp32) is still read by the d offset / d mask side where the column path is taken. */
int rr_dcn_dgrad_bf16_img(const float *x, const float *offset, const float *mask, const float *w, const float *dy,
                          const unsigned short *dy_bf16, float *dx, float *doffset, float *dmask, int n, int h, int wd,
                          int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                          int deformable_groups, hipStream_t stream);
/* rr_dcn_dgrad_bf16 with BOTH operands of the column-gradient sweep fed by LDS-DMA (round 5): the weights are packed to
 * bf16 inside the call (the forward's [tap][chunk][filter][32] layout), dY comes as the caller's bf16 image (dy_bf16) or is
 * rounded once into the workspace (dy_bf16 NULL).  ws: rr_dcn_dgrad_ws_bytes(n,p,q,c,k,r,s, dy_bf16 != NULL) bytes.
 * Layers the DMA kernel does not take (K % 32 != 0, other filter sizes ...) run rr_dcn_dgrad_bf16_ws's kernel: same results.
 * accumulate != 0: dx holds another consumer's gradient of x and is added to (the kernels scatter with float atomics; the
 * zero fill is skipped) — the host layer's shared fan-in buffer, no separate sum pass.
 * Replaces modulated_deformable_col2im(_coord)_cuda + the dcol GEMM (ext/dcn/src/cuda/dcn_v2_cuda.cu:214-259). */
size_t rr_dcn_dgrad_ws_bytes(int n, int p, int q, int c, int k, int r, int s, int have_dy_bf16);
int rr_dcn_dgrad_bf16_packed(const float *x, const float *offset, const float *mask, const float *w, const float *dy,
                             const unsigned short *dy_bf16, float *dx, float *doffset, float *dmask, int n, int h, int wd, int c,
                             int k, int r, int s, int stride, int pad_h, int pad_w, int dilation, int deformable_groups, int accumulate,
                             void *ws, hipStream_t stream);

/* rr_dcn_dgrad with bf16 matrix operands (dY, W rounded to bf16; fp32 accumulation and scatter): the backward of
 * rr_dcn_fwd_bf16.  Layers the window kernel does not take run rr_dcn_dgrad's fp32 kernel. */
int rr_dcn_dgrad_bf16(const float *x, const float *offset, const float *mask, const float *w, const float *dy, float *dx,
                      float *doffset, float *dmask, int n, int h, int wd, int c, int k, int r, int s, int stride,
                      int pad_h, int pad_w, int dilation, int deformable_groups, hipStream_t stream);

/* rr_dcn_wgrad with bf16 matrix operands (dY and the masked bilinear samples rounded to bf16; fp32 accumulation): a
 * workgroup keeps the [256 filters][9 taps x 32 channels] block of dW in its accumulators and walks over 8x16 pixel
 * blocks whose input window is staged in LDS as in rr_dcn_fwd_bf16.  3x3, stride 1, c % 32 == 0; other layers take the
 * rr_dcn_wgrad kernel. */
int rr_dcn_wgrad_bf16(const float *x, const float *offset, const float *mask, const float *dy, float *dw, int n, int h,
                      int wd, int c, int k, int r, int s, int stride, int pad_h, int pad_w, int dilation,
                      int deformable_groups, hipStream_t stream);

/* DCN module glue (ext/dcn/dcn_v2.py:117-121): om NHWC [m, 3*third] -> offset [m, 2*third] (first two thirds,
 * unchanged = cat(o1, o2)) and mask [m, third] = sigmoid(last third); backward: dom from doffset, dmask and mask. */
int rr_dcn_split_fwd(const float *om, long m, int third, float *offset, float *mask, hipStream_t stream);
int rr_dcn_split_bwd(const float *doffset, const float *dmask, const float *mask, long m, int third, float *dom,
                     hipStream_t stream);

/* ---- deformable PS-RoI pooling (ext/dcn/src/cuda/dcn_v2_psroi_pooling_cuda.cu:59-290; dcn_v2.py:130-300) ------ *
 * x NHWC [b,h,w,c], c = output_dim * group_size^2; rois [n,5] = (image, x1,y1,x2,y2); trans NCHW
 * [n, trans_channels = 2*num_classes, part, part] (ignored when no_trans); out / count NHWC [n, pooled, pooled,
 * output_dim].  Backward: dx (zeroed inside, float atomics) and dtrans (REQUIRED unless no_trans; zeroed inside).
 * A RoI whose image index is negative (backward: or >= b, the batch size it is told) pools nothing: out = count = 0. */
int rr_dcn_psroi_fwd(const float *x, const float *rois, const float *trans, int n, int h, int w, int c,
                     int no_trans, float spatial_scale, int output_dim, int group_size, int pooled_size,
                     int part_size, int sample_per_part, float trans_std, int trans_channels, float *out,
                     float *count, hipStream_t stream);
int rr_dcn_psroi_bwd(const float *dout, const float *x, const float *rois, const float *trans, const float *count,
                     int n, int b, int h, int w, int c, int no_trans, float spatial_scale, int output_dim,
                     int group_size, int pooled_size, int part_size, int sample_per_part, float trans_std,
                     int trans_channels, float *dx, float *dtrans, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RRNET_HIP_H */
