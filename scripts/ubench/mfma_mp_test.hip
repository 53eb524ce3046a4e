// "Measure, do not argue": the m * p half of the 9 x 29-bit Montgomery product (csrc/fr9.hip.h) as an int8 MFMA contraction against
// the constant Toeplitz matrix of p, versus the 81 v_mad_u64_u32 (+ 9 quotient-digit steps) it would replace.
//
//   A  k_mp_valu       the reduction half of fr9_mul as it is: the 17 column sums of a*b are given, the 9 quotient digits m_k and the
//                      81 products m_j * p_(k-j) run product-scanning in one 64-bit accumulator (p's limbs in SGPRs)
//   B  k_mp_mfma_pure  ONLY the matrix instructions of the MFMA form: u = m * p for the 64 elements of a wave as
//                      [64 x 38 seven-bit digits] x [38 x 76 Toeplitz(p)] -> 2 row tiles x 3 column tiles x 2 K tiles
//                      = 12 x v_mfma_i32_32x32x32_i8 (operands assumed to be in the MFMA register layout already)
//   C  k_mp_mfma_lower_bound  B + the cheapest conceivable VALU work around it: cutting the 9 limbs of m into 38 digits and packing
//                      them 4 per dword (no cross-lane transpose to the MFMA operand layout), and carry-propagating the 96 int32
//                      column sums a lane ends up holding (shift, add, mask per sum; again no transpose back to "one element per
//                      lane").  A LOWER bound for the MFMA route: the quotient m = T_low * (-p^-1) mod R, itself a constant
//                      product of the same shape, and both layout transposes are not even counted.
// Prints wave-cycles per 64 reductions (2.4 GHz) at four waves per SIMD on all CUs; A is what the library does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITER 2000
#define M29 0x1fffffffu
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// limbs of p = BLS12-381 r in 29-bit words (p_0 = 1)
__device__ __forceinline__ constexpr uint32_t P9(int i) {
    constexpr uint32_t p[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
    return p[i];
}
__device__ __forceinline__ void mad(uint64_t& acc, uint32_t a, uint32_t b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void mad_k(uint64_t& acc, uint32_t a, uint32_t k) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc"); }

__global__ void __launch_bounds__(256) k_mp_valu(uint32_t* out, uint32_t seed) {
    uint32_t t[17], r[9], m[9];
    for (int i = 0; i < 17; i++) t[i] = (seed * (i + 3) + threadIdx.x * 7919u) & M29;
    for (int it = 0; it < ITER; it++) {
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            acc += t[k];
#pragma unroll
            for (int j = 0; j < k; j++) mad_k(acc, m[j], P9(k - j));
            m[k] = (0u - (uint32_t)acc) & M29;
            mad(acc, m[k], 1u);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
            acc += t[k];
#pragma unroll
            for (int j = k - 8; j < 9; j++) mad_k(acc, m[j], P9(k - j));
            r[k - 9] = (uint32_t)acc & M29;
            acc >>= 29;
        }
        r[8] = (uint32_t)acc;
#pragma unroll
        for (int i = 0; i < 9; i++) { t[i] ^= r[i]; t[i + 8] += r[i] & 0xff; }
    }
    uint32_t s = 0;
    for (int i = 0; i < 17; i++) s += t[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mp_mfma_pure(uint32_t* out, uint32_t seed) {
    v4i a[4], b[6];   // A: 2 row tiles x 2 K tiles of m's digits; B: 3 column tiles x 2 K tiles of Toeplitz(p)
    for (int i = 0; i < 4; i++) a[i] = (v4i){(int)(seed + i), (int)threadIdx.x, (int)(seed * 3), 7};
    for (int i = 0; i < 6; i++) b[i] = (v4i){(int)(seed * 5 + i), 11, (int)threadIdx.x, 13};
    v16i c[6];
    for (int i = 0; i < 6; i++) c[i] = (v16i){0};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++) {
                c[rt * 3 + ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt * 2 + 0], b[ct * 2 + 0], c[rt * 3 + ct], 0, 0, 0);
                c[rt * 3 + ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt * 2 + 1], b[ct * 2 + 1], c[rt * 3 + ct], 0, 0, 0);
            }
        a[0].x ^= c[0][0] & 0x7f;   // keep the chain alive
    }
    uint32_t s = 0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 16; j++) s += (uint32_t)c[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mp_mfma_lower_bound(uint32_t* out, uint32_t seed) {
    uint32_t m[9];
    for (int i = 0; i < 9; i++) m[i] = (seed * (i + 3) + threadIdx.x * 7919u) & M29;
    v4i b[6];
    for (int i = 0; i < 6; i++) b[i] = (v4i){(int)(seed * 5 + i), 11, (int)threadIdx.x, 13};
    uint32_t keep = 0;
    for (int it = 0; it < ITER; it++) {
        // (1) 9 x 29 bits -> 38 digits of 7 bits, 4 per dword (10 dwords = the K = 64 bytes of one A row)
        uint32_t dg[10];
#pragma unroll
        for (int w = 0; w < 10; w++) {
            uint32_t v = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int d = 4 * w + q, bit = 7 * d;
                if (d < 38) {
                    const int li = bit / 29, sh = bit % 29;
                    uint32_t x = m[li] >> sh;
                    if (sh + 7 > 29 && li + 1 < 9) x |= m[li + 1] << (29 - sh);
                    v |= (x & 0x7fu) << (8 * q);
                }
            }
            dg[w] = v;
        }
        v4i a[4];
        a[0] = (v4i){(int)dg[0], (int)dg[1], (int)dg[2], (int)dg[3]};
        a[1] = (v4i){(int)dg[4], (int)dg[5], (int)dg[6], (int)dg[7]};
        a[2] = (v4i){(int)dg[8], (int)dg[9], 0, 0};
        a[3] = a[0];
        v16i c[6];
#pragma unroll
        for (int i = 0; i < 6; i++) c[i] = (v16i){0};
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++) {
                c[rt * 3 + ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt * 2 + 0], b[ct * 2 + 0], c[rt * 3 + ct], 0, 0, 0);
                c[rt * 3 + ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rt * 2 + 1], b[ct * 2 + 1], c[rt * 3 + ct], 0, 0, 0);
            }
        // (2) carry-propagate the 96 column sums this lane holds (7-bit digits): shift, add, mask
        uint32_t carry = 0, folded = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t v = (uint32_t)c[i][j] + carry;
                carry = v >> 7;
                folded += (v & 0x7fu) << (j & 15);
            }
        keep += folded + carry;
#pragma unroll
        for (int i = 0; i < 9; i++) m[i] = (m[i] + (keep >> i)) & M29;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep + m[0];
}

template <typename K>
static double run(K kern, const char* name, int waves_per_simd) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
    uint32_t* out = nullptr;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 999u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // every SIMD runs waves_per_simd waves of ITER iterations: cycles per iteration of ONE wave's worth of work on a SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * waves_per_simd);
    printf("%-26s %d waves/SIMD: %8.3f ms -> %7.0f SIMD cycles per 64 reductions (%.1f G reductions/s on the chip)\n", name, waves_per_simd, ms, cyc,
           (double)blocks * 256 * ITER / (ms * 1e-3) / 1e9);
    hipFree(out);
    return cyc;
}

int main() {
    double a = 0, b = 0, c = 0;
    for (int w : {2, 4}) {
        a = run(k_mp_valu, "A valu (81 mad + m digits)", w);
        b = run(k_mp_mfma_pure, "B mfma only (12 x 32x32x32)", w);
        c = run(k_mp_mfma_lower_bound, "C mfma + minimal VALU", w);
    }
    printf("verdict at 4 waves/SIMD: MFMA lower bound / VALU = %.2f (%s)\n", c / a, c > a ? "the MFMA route loses before its transposes and the quotient product are counted"
                                                                                       : "the MFMA route deserves a full implementation");
    return 0;
}
