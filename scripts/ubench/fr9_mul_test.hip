// Correctness + throughput of the 9 x 29-bit Montgomery product (csrc/fr9.hip.h) against the 8 x 32 one (csrc/fr.hip.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../gkr_msm_amd/csrc/fr9.hip.h"
using namespace gm;

__global__ void k_check(const Fr* a, const Fr* b, Fr* o1, Fr* o2, Fr* o3, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = fr_load(a + i), y = fr_load(b + i);
    fr_store(o1 + i, fr_mul(x, y));
    const Fr9 x9 = fr9_from(x), y9 = fr9_from(y);
    fr_store(o2 + i, fr9_to(fr9_mul(x9, y9)));
    // ((x + y) (x y - y^2) + 5 x) d through the lazy forms vs the canonical ones
    const Fr ref = fr_mul_by_d(fr_add(fr_mul(fr_add(x, y), fr_sub(fr_mul(x, y), fr_sqr(y))), fr_neg(fr_mul_by_a(x))));
    const Fr9 D = fr9_norm(fr9_sub2_32(fr9_mul(x9, y9), fr9_sqr(y9), fr9_zero()));     // L 2^29, S 47.5
    const Fr9 E = fr9_mul(fr9_add(x9, y9), D);                                           // L 2^30 x 2^29; S 44
    const Fr9 t = fr9_mul(fr9_norm(fr9_add(E, fr9_mul5(x9))), fr9_coeff_d());            // S (44 + 160) / 70.66 + 1
    fr_store(o3 + i, fr_eq(fr9_to(t), ref) && fr_eq(fr9_to(x9), x) ? fr_one() : fr_zero());
}
template <int KIND>
__global__ void k_chain(const Fr* a, Fr* o, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (KIND == 0) {
        Fr x = fr_load(a + i), y = fr_load(a + i + 1);
        for (int it = 0; it < iters; it++) { x = fr_mul(x, y); y = fr_mul(y, x); }
        fr_store(o + i, fr_add(x, y));
    } else if (KIND == 1) {
        Fr9 x = fr9_load(a + i), y = fr9_load(a + i + 1);
        for (int it = 0; it < iters; it++) { x = fr9_mul(x, y); y = fr9_mul(y, x); }
        fr9_store(o + i, fr9_add(x, y));
    } else {
        // the same number of products, two at a time (independent pairs)
        Fr9 x = fr9_load(a + i), y = fr9_load(a + i + 1), z = x, w = y;
        for (int it = 0; it < iters / 2; it++) { Fr9 t, u; fr9_mul2(x, y, z, w, t, u); x = t; z = u; fr9_mul2(y, x, w, z, t, u); y = t; w = u; }
        fr9_store(o + i, fr9_add(fr9_add(x, y), fr9_add(z, w)));
    }
}

int main() {
    const int n = 1 << 20;
    Fr *ha = (Fr*)malloc(n * 32 + 32), *hb = (Fr*)malloc(n * 32);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < n + 1; i++) for (int j = 0; j < 8; j++) ha[i].l[j] = rnd();
    for (int i = 0; i < n; i++) for (int j = 0; j < 8; j++) hb[i].l[j] = rnd();
    for (int i = 0; i < n + 1; i++) ha[i].l[7] &= 0x3fffffff;
    for (int i = 0; i < n; i++) hb[i].l[7] &= 0x3fffffff;
    for (int j = 0; j < 8; j++) { ha[0].l[j] = 0; hb[1].l[j] = 0; ha[2].l[j] = fr_p(j); hb[2].l[j] = fr_p(j); ha[3].l[j] = fr_p(j); hb[4].l[j] = fr_p(j); }
    ha[2].l[0] = 0; hb[2].l[0] = 0; ha[3].l[0] = 0; hb[4].l[0] = 0;  // p - 1
    for (int j = 0; j < 8; j++) { ha[5].l[j] = j == 0; hb[5].l[j] = j == 0; ha[6].l[j] = 0xffffffffu; hb[6].l[j] = 0xffffffffu; }
    ha[6].l[7] = 0x3fffffff; hb[6].l[7] = 0x3fffffff;
    Fr *da, *db, *o1, *o2, *o3;
    hipMalloc(&da, n * 32 + 32); hipMalloc(&db, n * 32); hipMalloc(&o1, n * 32); hipMalloc(&o2, n * 32); hipMalloc(&o3, n * 32);
    hipMemcpy(da, ha, n * 32 + 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, da, db, o1, o2, o3, n);
    Fr *h1 = (Fr*)malloc(n * 32), *h2 = (Fr*)malloc(n * 32), *h3 = (Fr*)malloc(n * 32);
    hipMemcpy(h1, o1, n * 32, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, n * 32, hipMemcpyDeviceToHost); hipMemcpy(h3, o3, n * 32, hipMemcpyDeviceToHost);
    int bad = 0, bad3 = 0;
    for (int i = 0; i < n; i++) if (memcmp(&h1[i], &h2[i], 32)) { if (bad < 3) printf("mismatch at %d\n", i); bad++; }
    for (int i = 0; i < n; i++) if (h3[i].l[0] == 0 && h3[i].l[1] == 0) { if (bad3 < 3) printf("formula mismatch at %d\n", i); bad3++; }
    printf("fr9 vs fr mismatches: %d / %d ; lazy formula mismatches: %d\n", bad, n, bad3);
    const int blocks = 256 * 8, threads = 256, iters = 200;
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms;
        double muls = (double)blocks * threads * iters * 2;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<0>), dim3(blocks), dim3(threads), 0, 0, da, o1, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("8x32 asm : %.3f ms  %.1f G mul/s\n", ms, muls / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<1>), dim3(blocks), dim3(threads), 0, 0, da, o2, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("9x29     : %.3f ms  %.1f G mul/s\n", ms, muls / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<2>), dim3(blocks), dim3(threads), 0, 0, da, o3, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("9x29 x2  : %.3f ms  %.1f G mul/s\n", ms, muls / ms / 1e6);
    }
    hipMemcpy(h1, o1, 4096 * 32, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, 4096 * 32, hipMemcpyDeviceToHost);
    const int chain_bad = memcmp(h1, h2, 4096 * 32) != 0;
    printf("chain results equal: %s\n", chain_bad ? "NO" : "yes");
    return bad != 0 || bad3 != 0 || chain_bad;
}
