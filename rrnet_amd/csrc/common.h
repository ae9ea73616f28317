// Shared helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define RR_OK 0
#define RR_ERR_ARG (-1)
#define RR_ERR_LAUNCH (-2)
#define RR_ERR_UNSUPPORTED (-3)

void rr_set_error(const char *fmt, ...);

#define RR_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            rr_set_error(__VA_ARGS__);     \
            return RR_ERR_ARG;             \
        }                                  \
    } while (0)

#define RR_CHECK_LAUNCH(name)                                              \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) {                                            \
            rr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return RR_ERR_LAUNCH;                                          \
        }                                                                  \
    } while (0)

// a HIP runtime call whose failure must surface through the C ABI's error code (memsets, function attributes)
#define RR_CHECK_HIP(call, name)                                           \
    do {                                                                   \
        hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) {                                            \
            rr_set_error("%s: %s failed: %s", name, #call, hipGetErrorString(e_)); \
            return RR_ERR_LAUNCH;                                          \
        }                                                                  \
    } while (0)

static inline int rr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// y*scale + shift of a BatchNorm layer as ONE explicit fma.  The forward (bn_apply), the two backward passes that recompute
// the ReLU mask from y instead of reading z (bn_bwd_reduce / bn_bwd_apply) and the data-gradient epilogue that carries the
// sums (conv.hip, rr_conv_dgrad_s1_bnsum) must agree on the sign of this value bit for bit: all four go through here, so
// the agreement does not hang on the compiler contracting (or not contracting) a mul + add the same way in each file.
__device__ __forceinline__ float rr_bn_affine(float y, float sc, float sh) { return __builtin_fmaf(y, sc, sh); }
typedef float rr_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rr_f32x4 rr_bn_affine4(rr_f32x4 y, rr_f32x4 sc, rr_f32x4 sh)
{
    rr_f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(y[e], sc[e], sh[e]);
    return r;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
