// Correctness + throughput of the 14 x 28-bit Fq form (csrc/fq14.hip.h) and of the three G1 additions evaluated in it
// (csrc/g1.hip.h: g1_add14 / g1_add_mixed14 / g1_add_aff14) against the 12 x 32 path, on the device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../gkr_msm_amd/csrc/g1.hip.h"
using namespace gm;

struct Six {
    Fq v[6];
};

__device__ __forceinline__ bool jac_eq(const G1Jac& a, const G1Jac& b) { return fq_eq(a.x, b.x) && fq_eq(a.y, b.y) && fq_eq(a.z, b.z); }

// out[i] bit 0: product / square / load-store round trip; bit 1: jacobian add; bit 2: mixed add; bit 3: affine add
__global__ void __launch_bounds__(256) k_check(const Six* in, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Six s;
    for (int j = 0; j < 6; j++) s.v[j] = fq_load(&in[i].v[j]);
    uint32_t ok = 0;
    const Fq14 a = fq14_from(s.v[0]), b = fq14_from(s.v[1]);
    if (fq_eq(fq14_to(fq14_mul(a, b)), fq_mul_c(s.v[0], s.v[1])) && fq_eq(fq14_to(fq14_sqr(a)), fq_mul_c(s.v[0], s.v[0])) &&
        fq_eq(fq14_to(a), s.v[0]) && fq_eq(fq14_to(fq14_norm(fq14_sub4(fq14_shl<2>(a), fq14_norm(fq14_add(b, fq14_shl<1>(b)))))),
                                           fq_sub(fq_dbl(fq_dbl(s.v[0])), fq_add(fq_dbl(s.v[1]), s.v[1]))))
        ok |= 1;
    G1Jac p, q;
    p.x = s.v[0]; p.y = s.v[1]; p.z = s.v[2]; q.x = s.v[3]; q.y = s.v[4]; q.z = s.v[5];
    G1Aff pa, qa;
    pa.x = s.v[0]; pa.y = s.v[1]; qa.x = s.v[3]; qa.y = s.v[4];
    if (jac_eq(g1_add14(p, q), g1_add_c(p, q))) ok |= 2;
    if (jac_eq(g1_add_mixed14(p, qa), g1_add_mixed_c(p, qa))) ok |= 4;
    if (jac_eq(g1_add_aff14(pa, qa), g1_add_aff_c(pa, qa))) ok |= 8;
    out[i] = ok;
}

template <int KIND>
__global__ void __launch_bounds__(256) k_chain(const Six* in, G1Jac* o, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    G1Jac p, q;
    p.x = fq_load(&in[i].v[0]); p.y = fq_load(&in[i].v[1]); p.z = fq_load(&in[i].v[2]);
    q.x = fq_load(&in[i].v[3]); q.y = fq_load(&in[i].v[4]); q.z = fq_load(&in[i].v[5]);
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { p = g1_add_c(p, q); q = g1_add_c(q, p); }
        else { p = g1_add14(p, q); q = g1_add14(q, p); }
    }
    g1_store(o + i, KIND == 0 ? g1_add_c(p, q) : g1_add14(p, q));
}

int main() {
    const int n = 1 << 18;
    Six* h = (Six*)malloc(sizeof(Six) * n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    auto canon = [&](Fq& x) {   // below q: clear the top bits, then subtract q while needed
        x.l[11] &= 0x1fffffffu;
        for (int r = 0; r < 2; r++) x = fq_reduce_once(x);
    };
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 6; j++) { for (int k = 0; k < 12; k++) h[i].v[j].l[k] = rnd(); canon(h[i].v[j]); }
    // edge cases: zeros, q - 1, ones, all 28-bit limbs full, the same point twice (H = 0), opposite points, infinities
    Fq qm1; for (int k = 0; k < 12; k++) qm1.l[k] = fq_p(k); qm1.l[0] -= 1;
    Fq full; for (int k = 0; k < 12; k++) full.l[k] = 0xffffffffu; full.l[11] = 0x0fffffffu; canon(full);
    for (int j = 0; j < 6; j++) { h[0].v[j] = fq_zero(); h[1].v[j] = qm1; h[2].v[j] = fq_one(); h[3].v[j] = full; }
    h[3].v[3] = qm1;
    for (int j = 0; j < 3; j++) h[4].v[3 + j] = h[4].v[j];                     // P + P
    for (int j = 0; j < 3; j++) h[5].v[3 + j] = h[5].v[j]; h[5].v[4] = fq_neg(h[5].v[1]);   // P + (-P)
    h[6].v[2] = fq_zero(); h[7].v[5] = fq_zero();                              // infinity on either side
    h[8].v[2] = fq_one(); h[8].v[5] = fq_one(); h[8].v[3] = h[8].v[0];         // Z = 1, same x: mixed / affine H = 0
    h[9].v[0] = fq_zero(); h[9].v[1] = fq_zero();                              // affine infinity
    Six* d; uint32_t* dout; G1Jac *o0, *o1;
    hipMalloc(&d, sizeof(Six) * n); hipMalloc(&dout, 4 * n); hipMalloc(&o0, sizeof(G1Jac) * n); hipMalloc(&o1, sizeof(G1Jac) * n);
    hipMemcpy(d, h, sizeof(Six) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, d, dout, n);
    uint32_t* ho = (uint32_t*)malloc(4 * n);
    hipMemcpy(ho, dout, 4 * n, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) for (int b = 0; b < 4; b++) if (!((ho[i] >> b) & 1)) { if (bad[b] < 3) printf("  case %d fails check %d\n", i, b); bad[b]++; }
    printf("fq14 field mismatches: %d / %d\n", bad[0], n);
    printf("g1 add mismatches: jacobian %d mixed %d affine %d / %d\n", bad[1], bad[2], bad[3], n);
    // throughput: chains of dependent jacobian additions, every lane of the chip busy
    const int iters = 50, nt = 256 * 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int kind = 0; kind < 2; kind++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k_chain<0>, dim3(nt / 256 > n / 256 ? n / 256 : nt / 256), dim3(256), 0, 0, d, o0, iters);
            else hipLaunchKernelGGL(k_chain<1>, dim3(nt / 256 > n / 256 ? n / 256 : nt / 256), dim3(256), 0, 0, d, o1, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[kind], e0, e1);
        }
        const double adds = (double)(n) * (2 * iters + 1);
        printf("%s: %.3f ms, %.2f G jacobian additions/s\n", kind == 0 ? "12 x 32" : "14 x 28", ms[kind], adds / ms[kind] / 1e6);
    }
    G1Jac* r0 = (G1Jac*)malloc(sizeof(G1Jac) * n); G1Jac* r1 = (G1Jac*)malloc(sizeof(G1Jac) * n);
    hipMemcpy(r0, o0, sizeof(G1Jac) * n, hipMemcpyDeviceToHost); hipMemcpy(r1, o1, sizeof(G1Jac) * n, hipMemcpyDeviceToHost);
    printf("chain results equal: %s\n", memcmp(r0, r1, sizeof(G1Jac) * n) == 0 ? "yes" : "NO");
    printf("speedup %.2fx\n", ms[0] / ms[1]);
    return (bad[0] | bad[1] | bad[2] | bad[3]) != 0 || memcmp(r0, r1, sizeof(G1Jac) * n) != 0;
}
