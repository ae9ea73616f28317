// Does a SIMD overlap one wave's matrix instructions with another wave's vector instructions?  (gfx950)
// 512 threads per block, one block per CU: waves 0-3 issue v_mfma_f32_32x32x16_bf16 on 4 independent accumulators, waves 4-7
// issue a vector instruction mix on 8 independent chains.  Times: matrix waves alone, vector waves alone, both.
//   hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe tools/coissue_probe.hip && tools/coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void probe(float *out, int n_mfma, int n_valu, int prio)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (n_mfma == 0) return;
        if (prio) __builtin_amdgcn_s_setprio(1);
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(e - lane); }
        for (int s = 0; s < n_mfma; ++s) {
#pragma unroll
            for (int r = 0; r < 12; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float v = 0.f;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) v += acc[i][e];
        out[blockIdx.x * 512 + threadIdx.x] = v;
    } else {
        if (n_valu == 0) return;
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = 1.0f + lane * 0.001f + i;
        for (int s = 0; s < n_valu; ++s) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == 0) x[i] = x[i] * 1.0001f - 0.5f;                 // v_fma / v_mul + v_sub
                    else if (KIND == 1) {                                            // cvt_pk + shift + sub (the split's inner step)
                        f32x2 p = {x[i], x[i]};
                        bf16x2 h = __builtin_convertvector(p, bf16x2);
                        unsigned u = __builtin_bit_cast(unsigned, h);
                        x[i] = x[i] - __builtin_bit_cast(float, u << 16) + 1.0f;
                    } else {                                                         // and + sub
                        unsigned u = __builtin_bit_cast(unsigned, x[i]) & 0xffff0000u;
                        x[i] = x[i] - __builtin_bit_cast(float, u) + 1.0f;
                    }
                }
        }
        float v = 0.f;
        for (int i = 0; i < 8; ++i) v += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = v;
    }
}

template <int KIND>
static float run(float *out, int nm, int nv, int prio)
{
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, nm, nv, prio);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, nm, nv, prio);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms;
}

int main()
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int nm = 2000, nv = 2000;      // 96000 MFMAs (32 cycles each) per wave; 96000 chain steps of the vector mix
    printf("matrix waves alone: %.3f ms (%d MFMAs per wave -> %.1f cycles each at 2.4 GHz)\n", run<0>(out, nm, 0, 0), nm * 48, run<0>(out, nm, 0, 0) * 2.4e6 / (nm * 48));
    const char *names[3] = {"mul+sub", "cvt_pk+shl+sub+add", "and+sub+add"};
    float va[3] = {run<0>(out, 0, nv, 0), run<1>(out, 0, nv, 0), run<2>(out, 0, nv, 0)};
    float vb[3] = {run<0>(out, nm, nv, 0), run<1>(out, nm, nv, 0), run<2>(out, nm, nv, 0)};
    float vc[3] = {run<0>(out, nm, nv, 1), run<1>(out, nm, nv, 1), run<2>(out, nm, nv, 1)};
    for (int k = 0; k < 3; ++k)
        printf("%-20s vector alone %.3f ms (%.1f cycles per chain step)   both %.3f ms   both, matrix waves at priority 1: %.3f ms\n", names[k], va[k],
               va[k] * 2.4e6 / (nv * 48), vb[k], vc[k]);
    return 0;
}
