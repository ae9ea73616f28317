#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
// LDS image: rows = k (32), cols = n (32), value = 100*k + n.  Each lane follows the T10 recipe for the B operand.
__global__ void k(short *out) {
    __shared__ short lds[32 * 32];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)(100 * (i / 32) + (i % 32));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const int n0 = 16 * (g & 1), k0 = 8 * (g >> 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + (k0 + q) * 32 + n0 + 4 * p));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + (k0 + 4 + q) * 32 + n0 + 4 * p));
    for (int e = 0; e < 4; ++e) { out[l * 8 + e] = lo[e]; out[l * 8 + 4 + e] = hi[e]; }
}
int main() {
    short *d; hipMalloc(&d, 64 * 8 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
        const int want = 100 * (8 * (l / 32) + e) + (l % 32);     // B[k = 8*(l/32)+e][n = l%32]
        if (h[l * 8 + e] != want) { if (bad < 8) printf("lane %d e %d got %d want %d\n", l, e, h[l * 8 + e], want); ++bad; }
    }
    printf("bad=%d\n", bad);
    return 0;
}
