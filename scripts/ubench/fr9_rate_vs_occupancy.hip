// Product rate of the 9 x 29-bit Fr form against waves per SIMD (capped with dynamic LDS): one chain of dependent products per wave and
// two interleaved ones.  MI355X: 153 G products/s at 8 waves per SIMD with one chain, 173-177 G at <= 6 waves or with two chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../gkr_msm_amd/csrc/fr9.hip.h"
using namespace gm;
template <int KIND>
__global__ void __launch_bounds__(256) k_mulchain(const Fr* in, Fr* o, int iters) {
    extern __shared__ uint32_t lds[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (iters < 0) lds[threadIdx.x] = i;
    Fr9 x = fr9_load(in + i), y = fr9_load(in + i + 1);
    if (KIND == 0) { for (int it = 0; it < iters; it++) { x = fr9_mul(x, y); y = fr9_mul(y, x); } }
    else {
        Fr9 z = fr9_load(in + i + 2), w = fr9_load(in + i + 3);
        for (int it = 0; it < iters / 2; it++) { Fr9 t, u; fr9_mul2(x, y, z, w, t, u); x = t; z = u; fr9_mul2(y, x, w, z, t, u); y = t; w = u; }
        x = fr9_norm(fr9_add(x, z)); y = fr9_norm(fr9_add(y, w));
    }
    fr9_store(o + i, fr9_mul(x, y));
}
int main() {
    const int n = 256 * 256 * 8, iters = 1000;
    Fr* h = (Fr*)malloc(32 * (n + 4));
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n + 4; i++) { for (int k = 0; k < 8; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].l[k] = (uint32_t)(s >> 16); } h[i].l[7] &= 0x3fffffff; }
    Fr *d, *o; hipMalloc(&d, 32 * (n + 4)); hipMalloc(&o, 32 * n);
    hipMemcpy(d, h, 32 * (n + 4), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_mulchain<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_mulchain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ldsz[6] = {0, 20 * 1024, 26 * 1024, 40 * 1024, 53 * 1024, 80 * 1024};
    const char* occ[6] = {"unrestricted", "<= 8", "<= 6", "<= 4", "<= 3", "<= 2"};
    for (int kind = 0; kind < 2; kind++)
        for (int l = 0; l < 6; l++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_mulchain<0>, dim3(n / 256), dim3(256), ldsz[l], 0, d, o, iters);
                else hipLaunchKernelGGL(k_mulchain<1>, dim3(n / 256), dim3(256), ldsz[l], 0, d, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-6s waves/SIMD %-14s %.3f ms  %.1f G Fr-mul/s\n", kind ? "mul2" : "mul", occ[l], ms, (double)n * 2 * iters / ms / 1e6);
        }
    return 0;
}
