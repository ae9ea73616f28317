// Standalone probe: where does a 128x128x32 fp32 MFMA tile loop lose MFMA issue slots on gfx950?
// Variants accumulate features of the real conv kernel's K-step.
//   0: MFMA only (operands in registers)            1: + LDS fragment reads (ds_read_b128, padded rows)
//   2: + ds_write_b128 of a tile + barrier / step   3: + global float4 loads (L2-resident source)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDK = 36;

template <int V>
__global__ __launch_bounds__(256) void probe(const float *src, float *out, int steps)
{
    extern __shared__ __align__(16) float lds[];
    float *As = lds, *Bs = lds + 2 * 128 * LDK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    for (int i = t; i < 4 * 128 * LDK; i += 256) lds[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 fa[2], fb[2];
    for (int i = 0; i < 2; ++i) { fa[i] = (f32x4){1.f + lane, 2.f, 3.f, 4.f}; fb[i] = (f32x4){0.5f, 0.25f, lane * 0.1f, 1.f}; }
    f32x4 ra[4], rb[4];
    const float *g = src + ((size_t)blockIdx.x * 256 + t) * 4;
    for (int j = 0; j < 4; ++j) { ra[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; rb[j] = ra[j]; }
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (V >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ra[j] = *reinterpret_cast<const f32x4 *>(g + (size_t)((s * 8 + j) & 63) * 262144);
                rb[j] = *reinterpret_cast<const f32x4 *>(g + (size_t)((s * 8 + 4 + j) & 63) * 262144);
            }
        }
        const float *A = As + buf * 128 * LDK, *B = Bs + buf * 128 * LDK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (V >= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(A + ((wm * 2 + i) * 32 + lr) * LDK + kk * 8 + lh * 4);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(B + ((wn * 2 + j) * 32 + lr) * LDK + kk * 8 + lh * 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        if (V >= 2) {
            float *A2 = As + (buf ^ 1) * 128 * LDK, *B2 = Bs + (buf ^ 1) * 128 * LDK;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<f32x4 *>(A2 + ((t >> 3) + 32 * j) * LDK + (t & 7) * 4) = ra[j];
                *reinterpret_cast<f32x4 *>(B2 + ((t >> 3) + 32 * j) * LDK + (t & 7) * 4) = rb[j];
            }
            __syncthreads();
        }
    }
    float r = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
    out[(size_t)blockIdx.x * 256 + t] = r;
}

template <int V>
void run(const float *src, float *out, int blocks, int steps)
{
    const size_t lds = 4 * 128 * LDK * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), lds, 0, src, out, steps);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), lds, 0, src, out, steps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double flops = (double)blocks * steps * 128.0 * 128.0 * 32.0 * 2.0;
    printf("variant %d: blocks %d steps %d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", V, blocks, steps, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 8192, steps = argc > 2 ? atoi(argv[2]) : 72;
    float *src, *out;
    hipMalloc(&src, (size_t)64 * 262144 * 4 + (size_t)blocks * 1024 * 4 + 4096);
    hipMemset(src, 0, (size_t)64 * 262144 * 4 + (size_t)blocks * 1024 * 4);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    run<0>(src, out, blocks, steps);
    run<1>(src, out, blocks, steps);
    run<2>(src, out, blocks, steps);
    run<3>(src, out, blocks, steps);
    return 0;
}
