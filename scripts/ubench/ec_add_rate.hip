// Pure-compute rate of the fused projective add (12 Fr muls) vs the same kernel shape with memory traffic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../gkr_msm_amd/csrc/fr.hip.h"
using namespace gm;
struct P3 { Fr x, y, z; };
__device__ __forceinline__ P3 padd(const P3& p, const P3& g) {
    const Fr A = fr_mul(p.x, g.x), B = fr_mul(p.y, g.y), zz = fr_mul(p.z, g.z);
    const Fr s = fr_sub(fr_sub(fr_mul(fr_add(p.x, p.y), fr_add(g.x, g.y)), A), B);
    const Fr t = fr_sub(B, fr_mul_by_a(A));
    const Fr X = fr_mul(s, zz), Y = fr_mul(t, zz), z2 = fr_sqr(zz);
    const Fr dxy = fr_mul_by_d(fr_mul(A, B));
    const Fr m = fr_sub(z2, dxy), q = fr_add(z2, dxy);
    P3 r; r.x = fr_mul(m, X); r.y = fr_mul(q, Y); r.z = fr_mul(m, q);
    return r;
}
__global__ void k_loop(const Fr* a, Fr* o, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    P3 p, g;
    p.x = fr_load(a + i); p.y = fr_load(a + i + 1); p.z = fr_load(a + i + 2);
    g.x = fr_load(a + i + 3); g.y = fr_load(a + i + 4); g.z = fr_load(a + i + 5);
    for (int it = 0; it < iters; it++) { p = padd(p, g); g = padd(g, p); }
    fr_store(o + i, fr_add(fr_add(p.x, p.y), fr_add(g.x, g.z)));
}
// streaming shape of k_add_level: out[j] = in[2j] + in[2j+1]
__global__ void k_stream(const Fr* ix, const Fr* iy, const Fr* iz, Fr* ox, Fr* oy, Fr* oz, uint32_t n) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    P3 p, g;
    p.x = fr_load(ix + 2 * j); p.y = fr_load(iy + 2 * j); p.z = fr_load(iz + 2 * j);
    g.x = fr_load(ix + 2 * j + 1); g.y = fr_load(iy + 2 * j + 1); g.z = fr_load(iz + 2 * j + 1);
    P3 r = padd(p, g);
    fr_store(ox + j, r.x); fr_store(oy + j, r.y); fr_store(oz + j, r.z);
}
int main() {
    const uint32_t n = 1 << 23;
    Fr *a, *o, *ix, *iy, *iz, *ox, *oy, *oz;
    hipMalloc(&a, (size_t)(1 << 20) * 32 + 256); hipMalloc(&o, (size_t)(1 << 20) * 32);
    hipMemset(a, 0x11, (size_t)(1 << 20) * 32 + 256);
    hipMalloc(&ix, (size_t)2 * n * 32); hipMalloc(&iy, (size_t)2 * n * 32); hipMalloc(&iz, (size_t)2 * n * 32);
    hipMalloc(&ox, (size_t)n * 32); hipMalloc(&oy, (size_t)n * 32); hipMalloc(&oz, (size_t)n * 32);
    hipMemset(ix, 0x12, (size_t)2 * n * 32); hipMemset(iy, 0x13, (size_t)2 * n * 32); hipMemset(iz, 0x14, (size_t)2 * n * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        const int blocks = 256 * 8, threads = 128, iters = 50;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_loop, dim3(blocks), dim3(threads), 0, 0, a, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double adds = (double)blocks * threads * iters * 2;
        printf("register loop : %.3f ms  %.2f G add/s = %.1f G mul/s\n", ms, adds / ms / 1e6, adds * 12 / ms / 1e6);
        for (int tb : {64, 128, 256}) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_stream, dim3((n + tb - 1) / tb), dim3(tb), 0, 0, ix, iy, iz, ox, oy, oz, n);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("streaming tb=%3d: %.3f ms  %.2f G add/s = %.1f G mul/s, %.0f GB/s\n", tb, ms, n / ms / 1e6, n * 12.0 / ms / 1e6,
                   n * 288.0 / ms / 1e6);
        }
    }
    return 0;
}
