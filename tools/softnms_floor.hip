// Latency floor of ONE outer step of a wavefront-parallel Soft-NMS on gfx950 (tools/bench_softnms.py).
//
// The outer loop of cpu_soft_nms (/root/reference/ext/nms/nms/cpu_nms.pyx:36-118) is a serial dependency chain: step i+1
// cannot pick its box before step i has decayed every score.  Whatever the decay costs, every step of a workgroup of W
// waves must at least (a) reduce a (score, index) pair over the 64 lanes of a wave and (b) exchange the W results
// through LDS behind one workgroup barrier.  This probe runs exactly that chain, nothing else, `steps` times: the
// per-lane candidate of the next step depends on the result of this one, so nothing overlaps.  One workgroup per
// `blocks`; time / steps = the floor a Soft-NMS step is quoted against ("µs per outer step vs the 64-lane
// reduction latency floor").
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int T>
__global__ __launch_bounds__(T) void floor_kernel(const float *seed, int steps, float *out)
{
    __shared__ float red_s[2][16];
    __shared__ int red_p[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = seed[(blockIdx.x * T + tid) & 4095];
    int p = tid;
    float acc = 0.f;
    for (int i = 0; i < steps; ++i) {
        float bs = s;
        int bp = p;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {       // 64-lane arg-max, lowest index among equal maxima
            const float os = __shfl_xor(bs, o, 64);
            const int op = __shfl_xor(bp, o, 64);
            const bool take = bs < os || (bs == os && op < bp);
            bs = take ? os : bs;
            bp = take ? op : bp;
        }
        if (T > 64) {                            // one LDS round trip behind one barrier (double-buffered slots)
            const int b = i & 1;
            if (lane == 0) { red_s[b][wave] = bs; red_p[b][wave] = bp; }
            __syncthreads();
            bs = red_s[b][0]; bp = red_p[b][0];
#pragma unroll
            for (int w = 1; w < T / 64; ++w) {
                const float os = red_s[b][w];
                const int op = red_p[b][w];
                const bool take = bs < os || (bs == os && op < bp);
                bs = take ? os : bs;
                bp = take ? op : bp;
            }
        }
        acc += bs;
        // the next candidate depends on this step's winner (a stand-in for the decay): no overlap between steps
        s = (tid == (bp & (T - 1))) ? s * 0.5f : s + bs * 1e-9f;
    }
    if (tid == 0) out[blockIdx.x] = acc;
}

extern "C" int softnms_floor_run(int threads, int blocks, int steps, const float *seed, float *out, hipStream_t stream)
{
    if (threads == 64) hipLaunchKernelGGL(floor_kernel<64>, dim3(blocks), dim3(64), 0, stream, seed, steps, out);
    else if (threads == 256) hipLaunchKernelGGL(floor_kernel<256>, dim3(blocks), dim3(256), 0, stream, seed, steps, out);
    else if (threads == 512) hipLaunchKernelGGL(floor_kernel<512>, dim3(blocks), dim3(512), 0, stream, seed, steps, out);
    else if (threads == 1024) hipLaunchKernelGGL(floor_kernel<1024>, dim3(blocks), dim3(1024), 0, stream, seed, steps, out);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
