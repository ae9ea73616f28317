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

typedef float rr_f32x4_fwd __attribute__((ext_vector_type(4)));
static inline int rr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Zero fill of one or two buffers in ONE launch.  hipMemsetAsync turns into three to four tiny runtime kernels per call
// (fillBufferAligned + copyBuffer: 607 of them for the 165 fills of a train step, rocprofv3); the split-K layers zero their
// output and their statistics slab in front of every launch.  16-byte multiples take the kernel, anything else the runtime.
static __global__ __launch_bounds__(256) void rr_zero2_kernel(rr_f32x4_fwd *a, long na, rr_f32x4_fwd *b, long nb)
{
    const rr_f32x4_fwd z = {0.f, 0.f, 0.f, 0.f};
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na; i += stride) a[i] = z;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) b[i] = z;
}
static inline hipError_t rr_zero2(void *a, size_t abytes, void *b, size_t bbytes, hipStream_t stream)
{
    if (b == nullptr) bbytes = 0;
    if ((abytes | bbytes) % 16 != 0 || (reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b)) % 16 != 0) {
        hipError_t e = abytes ? hipMemsetAsync(a, 0, abytes, stream) : hipSuccess;
        if (e == hipSuccess && bbytes) e = hipMemsetAsync(b, 0, bbytes, stream);
        return e;
    }
    const long na = (long)(abytes / 16), nb = (long)(bbytes / 16);
    if (na + nb == 0) return hipSuccess;
    long blocks = (na + nb + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rr_zero2_kernel, dim3((int)blocks), dim3(256), 0, stream, static_cast<rr_f32x4_fwd *>(a), na,
                       static_cast<rr_f32x4_fwd *>(b), nb);
    return hipGetLastError();
}

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
