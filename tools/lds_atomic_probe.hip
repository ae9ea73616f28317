// LDS integer-atomic throughput on gfx950 with the DCN data gradient's scatter pattern (csrc/dcn.hip, dcn_dgrad_win_kernel):
// 512 threads; 8 lanes share a window pixel (4 channels each), a wave's 8 pixels are consecutive, 4 bilinear corners.
//   u32 : 4 corners x 4 ds_add_u32 per item, window pixel stride 33 words (the round-2..5 kernel)
//   u64 : 4 corners x 2 ds_add_u64 per item (two channels per word pair), window pixel stride 34 words
//   u32r/u64r : the same with returning atomics (never used; for scale)
// Prints cycles per wave-instruction at one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_probe.hip -o tools/lds_atomic_probe && tools/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int NPX = 15 * 23, WW = 23;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const int *base_px, int iters, int *out)
{
    extern __shared__ __align__(16) int win[];
    constexpr int STR = (MODE & 1) ? 34 : 33;
    const int t = threadIdx.x;
    for (int i = t; i < NPX * STR; i += 512) win[i] = 0;
    __syncthreads();
    const int a_col = (t & 7) * 4, row = t >> 3;
    int px = base_px[(blockIdx.x * 64 + row) & 4095];
    int v = t + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int po = (e >> 1) * WW + (e & 1);
            int *db = win + (px + po) * STR + a_col;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicAdd(db + c, v + c);
            } else if constexpr (MODE == 1) {
                unsigned long long *d2 = reinterpret_cast<unsigned long long *>(db);
#pragma unroll
                for (int c = 0; c < 2; ++c) atomicAdd(d2 + c, ((unsigned long long)(unsigned)(v + c) << 32) + (long long)(v - c));
            }
        }
        px = (px + 7) % (NPX - WW - 2);
        v = v * 3 + 1;
    }
    __syncthreads();
    int s = 0;
    for (int i = t; i < NPX * STR; i += 512) s += win[i];
    if (s == 0x7fffffff) out[blockIdx.x] = s;
}

int main()
{
    int *base, *out;
    int h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = ((i % 64) / 16) * WW + (i % 16) + (i / 64) % 3;     // 8 x 16 block, small offsets
    hipMalloc(&base, sizeof(h));
    hipMalloc(&out, 4096 * 4);
    hipMemcpy(base, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 2000, blocks = 256;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), NPX * 34 * 4, 0, base, iters, out);
            else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), NPX * 34 * 4, 0, base, iters, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double instr_per_wave = (double)iters * 4 * (mode == 0 ? 4 : 2);
        const double cyc = best * 1e-3 * 2.4e9;                       // one workgroup per CU: 8 waves share the CU's LDS
        printf("%s: %.3f ms, %.1f cycles per wave-instruction per CU (8 waves issuing), %.1f cycles per (item, corner)\n",
               mode == 0 ? "ds_add_u32 x4" : "ds_add_u64 x2", best, cyc / (instr_per_wave * 8), cyc / (iters * 4.0 * 8));
    }
    return 0;
}
