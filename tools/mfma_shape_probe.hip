// Standalone probe (round 5): which fp32 matrix instruction sustains more FLOP/s on gfx950 under load — v_mfma_f32_32x32x2_f32
// (what csrc/conv.hip issues) or v_mfma_f32_16x16x4_f32?  Same FLOPs per cycle on paper (64 / clk / SIMD); the question is the clock
// the chip holds (MI355X_MICROARCH.md, DVFS give-back (7): for the bf16 shapes the 16x16 form delivered 1.12-1.15x on random data).
// Bare loops, operands in registers (random normal values, eight rotating fragments per operand), a 64x64 accumulator tile per wave
// in both shapes (the conv kernels' wave tile), W waves per SIMD, every CU busy, ~1 s per measurement after a 1 s warm-up.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe && tools/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const float *src, float *out, int iters)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(size_t)t * 16 + i]; b[i] = src[(size_t)t * 16 + 8 + i]; }
    float r = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)          // 4 k-steps of 2: 16 MFMAs = 16 x 4096 FLOP x 2
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k * 2 + i) & 7], b[(k * 2 + j + (it & 1)) & 7], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)          // 2 k-steps of 4: 32 MFMAs of 2048 x 2 FLOP = the same 8 k of a 64x64 tile
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k * 4 + i) & 7], b[(k * 4 + j + (it & 1)) & 7], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) r += acc[i][j][e];
    }
    out[t] = r;
}

int main()
{
    const int blocks_per_cu[2] = {1, 2};
    float *src, *out;
    const size_t n = (size_t)256 * 8 * 256 * 16;
    hipMalloc(&src, n * 4); hipMalloc(&out, n / 4);
    float *h = (float *)malloc(n * 4);
    srand(1);
    for (size_t i = 0; i < n; ++i) {       // Box-Muller
        float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f);
        h[i] = sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char *data_name[3] = {"dense N(0,1) operands", "A = relu(N(0,1)) (half zeros: post-ReLU activations), B = 0.03 N(0,1) (filters)", "all-zero operands"};
    for (int data = 0; data < 3; ++data) {
    printf("--- %s\n", data_name[data]);
    {
        float *h2 = (float *)malloc(n * 4);
        for (size_t i = 0; i < n; ++i) {
            const bool is_a = (i % 16) < 8;
            h2[i] = data == 0 ? h[i] : data == 1 ? (is_a ? fmaxf(h[i], 0.f) : 0.03f * h[i]) : 0.f;
        }
        hipMemcpy(src, h2, n * 4, hipMemcpyHostToDevice);
        free(h2);
    }
    for (int w = 0; w < 2; ++w) {
        const int blocks = 256 * blocks_per_cu[w];
        for (int rep = 0; rep < 1; ++rep)
            for (int shape = 0; shape < 2; ++shape) {
                auto launch = [&]() {
                    if (shape == 0) hipLaunchKernelGGL(probe<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
                    else hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
                };
                for (int i = 0; i < 30; ++i) launch();       // warm-up under load
                hipDeviceSynchronize();
                hipEventRecord(e0);
                const int L = 30;
                for (int i = 0; i < L; ++i) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)blocks * 4 * iters * 16 * 4096.0 * L;       // 16 x (32x32x2 MACs = 4096 FLOP) per wave and iteration
                printf("%s  %d wave(s) per SIMD: %.1f TFLOP/s (%.1f ms per launch)\n", shape == 0 ? "32x32x2_f32" : "16x16x4_f32",
                       blocks_per_cu[w], flop / (ms * 1e-3) / 1e12, ms / L);
            }
    }
    }
    return 0;
}
