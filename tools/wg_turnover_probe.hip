// What a workgroup costs before it computes anything, at the DCN kernels' launch shape on gfx950: 4096 workgroups of 512 threads
// with 144 KB of dynamic LDS each (one per CU at a time, 16 rounds).  Variants: empty body; N workgroup barriers; and the same with
// 256 threads / 64 KB (two per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/wg_turnover_probe.hip -o tools/wg_turnover_probe && tools/wg_turnover_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NB>
__global__ void probe(int *out, int lds_words)
{
    extern __shared__ int smem[];
    int v = threadIdx.x;
    if (lds_words > 0) smem[threadIdx.x] = v;
#pragma unroll 1
    for (int i = 0; i < NB; ++i) {
        __syncthreads();
        v += smem[(threadIdx.x + i) & 511];
    }
    if (v == 0x7fffffff) out[blockIdx.x] = v;
}

template <int NB>
static float run(int blocks, int threads, int lds)
{
    int *out;
    hipMalloc(&out, 1 << 20);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<NB>, dim3(blocks), dim3(threads), lds, 0, out, lds / 4);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipFree(out);
    return best;
}

int main()
{
    printf("4096 x 512 threads, 144 KB LDS, no barrier : %.3f ms\n", run<0>(4096, 512, 144 * 1024));
    printf("4096 x 512 threads, 144 KB LDS, 88 barriers: %.3f ms\n", run<88>(4096, 512, 144 * 1024));
    printf("4096 x 512 threads,  64 KB LDS, 88 barriers: %.3f ms\n", run<88>(4096, 512, 64 * 1024));
    printf("4096 x 512 threads,   4 KB LDS, 88 barriers: %.3f ms\n", run<88>(4096, 512, 4 * 1024));
    printf(" 256 x 512 threads, 144 KB LDS, 88 barriers: %.3f ms (one round)\n", run<88>(256, 512, 144 * 1024));
    printf(" 256 x 512 threads, 144 KB LDS, 1408 barriers (16 tiles' worth in one workgroup): %.3f ms\n", run<1408>(256, 512, 144 * 1024));
    return 0;
}
