// Correctness + throughput of the fixed-register asm Fr multiplication vs the C formulation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../gkr_msm_amd/csrc/fr.hip.h"
using namespace gm;

__device__ __forceinline__ Fr fr_mul_asm(const Fr& a, const Fr& b) {
    Fr r;
    asm volatile(
#include "../../gkr_msm_amd/csrc/fr_mul_asm.inc"
        : "={v16}"(r.l[0]), "={v18}"(r.l[1]), "={v20}"(r.l[2]), "={v22}"(r.l[3]), "={v24}"(r.l[4]), "={v26}"(r.l[5]),
          "={v28}"(r.l[6]), "={v30}"(r.l[7])
        : "{v0}"(a.l[0]), "{v1}"(a.l[1]), "{v2}"(a.l[2]), "{v3}"(a.l[3]), "{v4}"(a.l[4]), "{v5}"(a.l[5]), "{v6}"(a.l[6]),
          "{v7}"(a.l[7]), "{v8}"(b.l[0]), "{v9}"(b.l[1]), "{v10}"(b.l[2]), "{v11}"(b.l[3]), "{v12}"(b.l[4]),
          "{v13}"(b.l[5]), "{v14}"(b.l[6]), "{v15}"(b.l[7])
        : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "v17", "v19", "v21", "v23", "v25", "v27", "v29", "v31", "v32",
          "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
          "v49", "v50");
    return r;
}

__global__ void k_check(const Fr* a, const Fr* b, Fr* o1, Fr* o2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = fr_load(a + i), y = fr_load(b + i);
    fr_store(o1 + i, fr_mul(x, y));
    fr_store(o2 + i, fr_mul_asm(x, y));
}
template <bool ASM>
__global__ void k_chain(const Fr* a, Fr* o, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x = fr_load(a + i), y = fr_load(a + i + 1);
    for (int it = 0; it < iters; it++) {
        x = ASM ? fr_mul_asm(x, y) : fr_mul(x, y);
        y = ASM ? fr_mul_asm(y, x) : fr_mul(y, x);
    }
    fr_store(o + i, fr_add(x, y));
}

int main() {
    const int n = 1 << 20;
    Fr *ha = (Fr*)malloc(n * 32 + 32), *hb = (Fr*)malloc(n * 32);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < n + 1; i++) for (int j = 0; j < 8; j++) ha[i].l[j] = rnd();
    for (int i = 0; i < n; i++) for (int j = 0; j < 8; j++) hb[i].l[j] = rnd();
    // inputs must be < p: clear the top bits (p > 2^254) and add edge cases
    for (int i = 0; i < n + 1; i++) ha[i].l[7] &= 0x3fffffff;
    for (int i = 0; i < n; i++) hb[i].l[7] &= 0x3fffffff;
    for (int j = 0; j < 8; j++) { ha[0].l[j] = 0; hb[1].l[j] = 0; ha[2].l[j] = fr_p(j); hb[2].l[j] = fr_p(j); }
    ha[2].l[0] = 0; hb[2].l[0] = 0;  // p - 1
    Fr *da, *db, *o1, *o2;
    hipMalloc(&da, n * 32 + 32); hipMalloc(&db, n * 32); hipMalloc(&o1, n * 32); hipMalloc(&o2, n * 32);
    hipMemcpy(da, ha, n * 32 + 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, da, db, o1, o2, n);
    Fr *h1 = (Fr*)malloc(n * 32), *h2 = (Fr*)malloc(n * 32);
    hipMemcpy(h1, o1, n * 32, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, n * 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) if (memcmp(&h1[i], &h2[i], 32)) { if (bad < 3) printf("mismatch at %d\n", i); bad++; }
    // host cross-check of a few
    int badh = 0;
    for (int i = 0; i < 1000; i++) { Fr r = fr_mul(ha[i], hb[i]); if (memcmp(&r, &h1[i], 32)) badh++; }
    printf("asm vs C mismatches: %d / %d ; C-device vs host mismatches: %d / 1000\n", bad, n, badh);
    const int blocks = 256 * 8, threads = 256, iters = 200;
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<false>), dim3(blocks), dim3(threads), 0, 0, da, o1, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double muls = (double)blocks * threads * iters * 2;
        printf("C   : %.3f ms  %.1f G mul/s\n", ms, muls / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<true>), dim3(blocks), dim3(threads), 0, 0, da, o2, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("asm : %.3f ms  %.1f G mul/s\n", ms, muls / ms / 1e6);
    }
    // occupancy sweep: dynamic LDS limits resident 256-thread blocks per CU (= waves per SIMD)
    for (int occ : {1, 2, 3, 4, 6, 8}) {
        const size_t lds = (160 * 1024) / occ - 512;
        hipFuncSetAttribute((const void*)k_chain<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_chain<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        float msC, msA;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<false>), dim3(blocks), dim3(threads), lds, 0, da, o1, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msC, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<true>), dim3(blocks), dim3(threads), lds, 0, da, o2, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msA, e0, e1);
        double muls = (double)blocks * threads * iters * 2;
        printf("waves/SIMD %d : C %.1f G mul/s   asm %.1f G mul/s\n", occ, muls / msC / 1e6, muls / msA / 1e6);
    }
    hipMemcpy(h1, o1, 4096 * 32, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, 4096 * 32, hipMemcpyDeviceToHost);
    printf("chain results equal: %s\n", memcmp(h1, h2, 4096 * 32) ? "NO" : "yes");
    return bad != 0;
}
