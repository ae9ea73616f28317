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

// variant 4: the pipelined schedule of conv_igemm_kernel<PIPE>: fragments one group ahead, ds_writes /
// global loads dealt out between MFMA sub-groups, one raw barrier between groups 2 and 3
__global__ __launch_bounds__(256) void probe_pipe(const float *src, float *out, int steps)
{
    extern __shared__ __align__(16) float lds[];
    float *As = lds, *Bs = lds + 2 * 128 * LDK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    for (int i = t; i < 4 * 128 * LDK; i += 256) lds[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 fa0[2], fb0[2], fa1[2], fb1[2], ra[4], rb[4];
    const float *g = src + ((size_t)blockIdx.x * 256 + t) * 4;
    for (int j = 0; j < 4; ++j) { ra[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; rb[j] = ra[j]; }
    auto rd = [&](int buf, int kk, f32x4 (&fa)[2], f32x4 (&fb)[2]) {
        const float *A = As + buf * 128 * LDK, *B = Bs + buf * 128 * LDK;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(A + ((wm * 2 + i) * 32 + lr) * LDK + kk * 8 + lh * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(B + ((wn * 2 + j) * 32 + lr) * LDK + kk * 8 + lh * 4);
    };
    auto sub = [&](const f32x4 (&fa)[2], const f32x4 (&fb)[2], int e) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto st = [&](int j, int buf, bool b) {
        float *P = (b ? Bs : As) + buf * 128 * LDK;
        *reinterpret_cast<f32x4 *>(P + ((t >> 3) + 32 * j) * LDK + (t & 7) * 4) = b ? rb[j] : ra[j];
    };
    auto ld = [&](int j, int s, bool b) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(g + (size_t)((s * 8 + (b ? 4 : 0) + j) & 63) * 262144);
        if (b) rb[j] = v; else ra[j] = v;
    };
    rd(0, 0, fa0, fb0);
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        rd(buf, 1, fa1, fb1);
        sub(fa0, fb0, 0); sub(fa0, fb0, 1); sub(fa0, fb0, 2); sub(fa0, fb0, 3);
        rd(buf, 2, fa0, fb0);
        st(0, buf ^ 1, false); st(1, buf ^ 1, false); sub(fa1, fb1, 0);
        st(2, buf ^ 1, false); st(3, buf ^ 1, false); sub(fa1, fb1, 1);
        st(0, buf ^ 1, true); st(1, buf ^ 1, true); sub(fa1, fb1, 2);
        st(2, buf ^ 1, true); st(3, buf ^ 1, true); sub(fa1, fb1, 3);
        rd(buf, 3, fa1, fb1);
        ld(0, s, false); ld(1, s, false); sub(fa0, fb0, 0);
        ld(2, s, false); ld(3, s, false); sub(fa0, fb0, 1);
        ld(0, s, true); ld(1, s, true); sub(fa0, fb0, 2);
        ld(2, s, true); ld(3, s, true); sub(fa0, fb0, 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rd(buf ^ 1, 0, fa0, fb0);
        sub(fa1, fb1, 0); sub(fa1, fb1, 1); sub(fa1, fb1, 2); sub(fa1, fb1, 3);
    }
    float r = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
    out[(size_t)blockIdx.x * 256 + t] = r;
}

void run_pipe(const float *src, float *out, int blocks, int steps)
{
    const size_t lds = 4 * 128 * LDK * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe_pipe, dim3(blocks), dim3(256), lds, 0, src, out, steps);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(probe_pipe, dim3(blocks), dim3(256), lds, 0, src, out, steps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double flops = (double)blocks * steps * 128.0 * 128.0 * 32.0 * 2.0;
    printf("variant 4 (pipelined): blocks %d steps %d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", blocks, steps, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
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
    run_pipe(src, out, blocks, steps);
    return 0;
}
