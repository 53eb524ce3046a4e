// Product rate of the 14 x 28-bit Fq form against waves per SIMD (capped with dynamic LDS).  MI355X: 68-74 G products/s unrestricted,
// 76-78 G at <= 4 or <= 2 waves per SIMD, 45 G with a single wave per SIMD (two interleaved chains do not help there).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../gkr_msm_amd/csrc/fq14.hip.h"
using namespace gm;
template <int KIND>
__global__ void __launch_bounds__(256) k_mulchain(const Fq* in, Fq* o, int iters) {
    extern __shared__ uint32_t lds[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (iters < 0) lds[threadIdx.x] = i;   // keep the allocation
    Fq14 x = fq14_from(fq_load(in + i)), y = fq14_from(fq_load(in + i + 1));
    if (KIND == 0) { for (int it = 0; it < iters; it++) { x = fq14_mul(x, y); y = fq14_mul(y, x); } }
    else {
        Fq14 z = fq14_from(fq_load(in + i + 2)), w = fq14_from(fq_load(in + i + 3));
        for (int it = 0; it < iters / 2; it++) { Fq14 t, u; fq14_mul2(x, y, z, w, t, u); x = t; z = u; fq14_mul2(y, x, w, z, t, u); y = t; w = u; }
        x = fq14_add(x, z); y = fq14_add(y, w);
    }
    fq_store(o + i, fq14_to(fq14_norm(fq14_add(x, y))));
}
int main() {
    const int n = 256 * 256 * 8, iters = 400;
    Fq* h = (Fq*)malloc(48 * (n + 4));
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n + 4; i++) { for (int k = 0; k < 12; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].l[k] = (uint32_t)(s >> 16); } h[i].l[11] &= 0x0fffffff; }
    Fq *d, *o; hipMalloc(&d, 48 * (n + 4)); hipMalloc(&o, 48 * n);
    hipMemcpy(d, h, 48 * (n + 4), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_mulchain<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_mulchain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ldsz[4] = {0, 40 * 1024, 80 * 1024, 160 * 1024};
    const char* occ[4] = {"unrestricted", "<= 4 waves/SIMD", "<= 2 waves/SIMD", "1 wave/SIMD"};
    for (int kind = 0; kind < 2; kind++)
        for (int l = 0; l < 4; l++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_mulchain<0>, dim3(n / 256), dim3(256), ldsz[l], 0, d, o, iters);
                else hipLaunchKernelGGL(k_mulchain<1>, dim3(n / 256), dim3(256), ldsz[l], 0, d, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-10s %-18s %.3f ms  %.1f G Fq-mul/s  (%s)\n", kind ? "mul2" : "mul", occ[l], ms, (double)n * 2 * iters / ms / 1e6, hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
