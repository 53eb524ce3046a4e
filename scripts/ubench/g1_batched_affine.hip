// Batched-affine G1 additions against the Jacobian ones, measured (verdict r02 item 9: "measure, do not argue").
//
// N independent additions P_i + Q_i of affine points, three ways, all in the 14 x 28 field form (csrc/fq14.hip.h):
//   aff+aff -> Jacobian   (g1_add_aff14p: what level 0 of the sum-by-key tree runs; 4 M + 2 S)
//   Jac+Jac -> Jacobian   (g1_add14p: what the levels above it run; 11 M + 5 S)
//   batched affine -> affine: ONE field inversion per workgroup shared by 256 x KPT additions (Montgomery's trick), three launches:
//     K1  per thread KPT chained products of the denominators x2 - x1 (stored), then an inclusive prefix and suffix scan of the
//         threads' totals through LDS; per thread the product of "everybody else" (exclusive prefix x exclusive suffix) is stored
//     K2  one inversion per workgroup total (Fermat, fq_inv), all workgroups' side by side
//     K3  per thread: the inverse of its own total = inv(total of the workgroup) x everybody else's product, walked backwards
//         through its chain; lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1
//   ~8 products per addition at KPT = 8 (1 + 17/KPT in K1, 2 + 3 + 1/KPT in K3) against 16 for Jac + Jac.
// The exceptional cases (P = +-Q, infinity) are NOT handled: this is the speed of the common path, which is what the decision needs.
// Correctness of that path: 64 sampled results against the 12 x 32 Jacobian formula brought to affine form on the host.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../gkr_msm_amd/csrc/g1.hip.h"
using namespace gm;

#define KPT 8
struct Raw14 { uint32_t l[14]; };
__device__ __forceinline__ void st14(Raw14* p, const Fq14& v) {
#pragma unroll
    for (int i = 0; i < 14; i++) p->l[i] = v.l[i];
}
__device__ __forceinline__ Fq14 ld14(const Raw14* p) {
    Fq14 v;
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = p->l[i];
    return v;
}

__global__ void __launch_bounds__(256) k_affaff(const G1Aff* P, const G1Aff* Q, G1Jac* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const G1P14 r = g1_add_aff14p(g1_aff_load(P + i), g1_aff_load(Q + i));
    g1_store(out + i, g1p14_to(r));
}
__global__ void __launch_bounds__(256) k_jacjac(const G1Jac* P, const G1Jac* Q, G1Jac* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const G1P14 r = g1_add14p(g1p14_from(g1_load(P + i)), g1p14_from(g1_load(Q + i)));
    g1_store(out + i, g1p14_to(r));
}

// K1
__global__ void __launch_bounds__(256) k_ba_prefix(const G1Aff* P, const G1Aff* Q, Raw14* pre, Raw14* others, Raw14* totals, int n) {
    __shared__ Raw14 sp[256], ss[256];
    const int t = threadIdx.x;
    const long base = ((long)blockIdx.x * 256 + t) * KPT;
    Fq14 run;
#pragma unroll 1
    for (int j = 0; j < KPT; j++) {
        const Fq14 d = fq14_norm(fq14_sub4(fq14_from(fq_load(&Q[base + j].x)), fq14_from(fq_load(&P[base + j].x))));
        run = j ? fq14_mul(run, d) : d;
        st14(pre + base + j, run);
    }
    st14(&sp[t], run);
    st14(&ss[t], run);
    __syncthreads();
    // inclusive prefix (sp) and inclusive suffix (ss) products over the 256 threads' totals (Hillis-Steele, 8 steps)
    for (int d = 1; d < 256; d <<= 1) {
        Fq14 a = ld14(&sp[t]), b = ld14(&ss[t]);
        const bool hp = t >= d, hs = t + d < 256;
        Fq14 ap, bs;
        if (hp) ap = ld14(&sp[t - d]);
        if (hs) bs = ld14(&ss[t + d]);
        __syncthreads();
        if (hp) st14(&sp[t], fq14_mul(a, ap));
        if (hs) st14(&ss[t], fq14_mul(b, bs));
        __syncthreads();
    }
    // everybody else's product: exclusive prefix x exclusive suffix
    Fq14 e;
    if (t == 0) e = ld14(&ss[1]);
    else if (t == 255) e = ld14(&sp[254]);
    else e = fq14_mul(ld14(&sp[t - 1]), ld14(&ss[t + 1]));
    st14(others + (long)blockIdx.x * 256 + t, e);
    if (t == 255) st14(totals + blockIdx.x, ld14(&sp[255]));
}
// K2
__global__ void __launch_bounds__(64) k_ba_invert(Raw14* totals, int nblocks) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nblocks) return;
    const Fq v = fq14_to(fq14_norm(ld14(totals + b)));
    st14(totals + b, fq14_from(fq_inv(v)));
}
// K3
__global__ void __launch_bounds__(256) k_ba_add(const G1Aff* P, const G1Aff* Q, const Raw14* pre, const Raw14* others, const Raw14* totals,
                                                G1Aff* out, int n) {
    const int t = threadIdx.x;
    const long base = ((long)blockIdx.x * 256 + t) * KPT;
    Fq14 inv_run = fq14_mul(ld14(totals + blockIdx.x), ld14(others + (long)blockIdx.x * 256 + t));   // 1 / (this thread's chain total)
#pragma unroll 1
    for (int j = KPT - 1; j >= 0; j--) {
        const Fq14 x1 = fq14_from(fq_load(&P[base + j].x)), y1 = fq14_from(fq_load(&P[base + j].y));
        const Fq14 x2 = fq14_from(fq_load(&Q[base + j].x)), y2 = fq14_from(fq_load(&Q[base + j].y));
        Fq14 inv_d;
        if (j) {
            inv_d = fq14_mul(inv_run, ld14(pre + base + j - 1));                       // 1 / d_j
            inv_run = fq14_mul(inv_run, fq14_norm(fq14_sub4(x2, x1)));                 // 1 / (d_0 .. d_{j-1})
        } else {
            inv_d = inv_run;
        }
        const Fq14 lam = fq14_mul(fq14_norm(fq14_sub4(y2, y1)), inv_d);               // S 1
        const Fq14 x3 = fq14_norm(fq14_sub4(fq14_norm(fq14_sub4(fq14_sqr(lam), x1)), x2));   // S <= 9.1
        const Fq14 y3 = fq14_norm(fq14_sub4(fq14_mul(lam, fq14_sub16(x1, x3)), y1));          // S <= 5.1
        G1Aff r;
        r.x = fq14_to(x3);
        r.y = fq14_to(y3);
        g1_aff_store(out + base + j, r);
    }
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 22;
    const int n = 1 << lg, nblocks = n / (256 * KPT);
    std::vector<G1Aff> hp(n), hq(n);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&](Fq& f) { for (int k = 0; k < 12; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; f.l[k] = (uint32_t)(s >> 16); } f.l[11] &= 0x0fffffff; };
    for (int i = 0; i < n; i++) { rnd(hp[i].x); rnd(hp[i].y); rnd(hq[i].x); rnd(hq[i].y); }
    G1Aff *dp, *dq, *dout_a;
    G1Jac *djp, *djq, *dout_j;
    Raw14 *pre, *others, *totals;
    hipMalloc(&dp, sizeof(G1Aff) * n); hipMalloc(&dq, sizeof(G1Aff) * n); hipMalloc(&dout_a, sizeof(G1Aff) * n);
    hipMalloc(&djp, sizeof(G1Jac) * n); hipMalloc(&djq, sizeof(G1Jac) * n); hipMalloc(&dout_j, sizeof(G1Jac) * n);
    hipMalloc(&pre, sizeof(Raw14) * (size_t)n); hipMalloc(&others, sizeof(Raw14) * (size_t)nblocks * 256); hipMalloc(&totals, sizeof(Raw14) * nblocks);
    hipMemcpy(dp, hp.data(), sizeof(G1Aff) * n, hipMemcpyHostToDevice);
    hipMemcpy(dq, hq.data(), sizeof(G1Aff) * n, hipMemcpyHostToDevice);
    {   // Jacobian operands: the same points with z = some non-trivial value (the coordinates need not be consistent for a rate)
        std::vector<G1Jac> jp(n), jq(n);
        for (int i = 0; i < n; i++) { jp[i].x = hp[i].x; jp[i].y = hp[i].y; jp[i].z = hq[i].y; jq[i].x = hq[i].x; jq[i].y = hq[i].y; jq[i].z = hp[i].y; }
        hipMemcpy(djp, jp.data(), sizeof(G1Jac) * n, hipMemcpyHostToDevice);
        hipMemcpy(djq, jq.data(), sizeof(G1Jac) * n, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_affaff, dim3(n / 256), dim3(256), 0, 0, dp, dq, dout_j, n);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("aff + aff -> Jacobian : %.3f ms  %.2f G additions/s\n", ms, n / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_jacjac, dim3(n / 256), dim3(256), 0, 0, djp, djq, dout_j, n);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("Jac + Jac -> Jacobian : %.3f ms  %.2f G additions/s\n", ms, n / ms / 1e6);
        float m1, m2, m3;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ba_prefix, dim3(nblocks), dim3(256), 0, 0, dp, dq, pre, others, totals, n);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&m1, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ba_invert, dim3((nblocks + 63) / 64), dim3(64), 0, 0, totals, nblocks);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&m2, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ba_add, dim3(nblocks), dim3(256), 0, 0, dp, dq, pre, others, totals, dout_a, n);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&m3, e0, e1);
        if (rep) printf("batched affine (KPT %d): prefix %.3f + invert %.3f (%d inversions) + add %.3f = %.3f ms  %.2f G additions/s (%.2f without the inversion's latency)\n",
                        KPT, m1, m2, nblocks, m3, m1 + m2 + m3, n / (m1 + m2 + m3) / 1e6, n / (m1 + m3) / 1e6);
    }
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    // 64 samples against the Jacobian formula on the host
    std::vector<G1Aff> got(n);
    hipMemcpy(got.data(), dout_a, sizeof(G1Aff) * n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k = 0; k < 64; k++) {
        const int i = (int)(((long)k * 2654435761u) % n);
        const G1Aff want = g1_to_aff(g1_add_aff_c(hp[i], hq[i]));
        if (!fq_eq(want.x, got[i].x) || !fq_eq(want.y, got[i].y)) bad++;
    }
    printf("batched affine vs Jacobian formula (64 samples): %d mismatches\n", bad);
    return bad != 0;
}
