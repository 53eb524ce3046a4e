// Does the "s_nop 0" the compiler puts after every inline-asm v_mad_u64_u32 cost issue slots?  Two kernels run the same chain of
// multiply-adds (two independent accumulators, alternating): A = one asm statement per instruction (the compiler pads each with
// s_nop 0), B = the same instructions inside ONE asm statement (no padding).  Waves per SIMD as an argument (LDS padding).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHAIN 64
__device__ __forceinline__ void mad(uint64_t& acc, uint32_t a, uint32_t b) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t x, uint32_t y) {
    extern __shared__ char pad[];
    uint64_t a = threadIdx.x, b = blockIdx.x;
    uint32_t u = x + threadIdx.x, v = y ^ blockIdx.x;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < CHAIN; i++) { mad(a, u, v); mad(b, v, u); }
        } else if (MODE == 2) {   // ONE dependent chain, no padding
#pragma unroll
            for (int i = 0; i < CHAIN / 8; i++)
                asm volatile(
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0"
                    : "+v"(a), "+v"(b) : "v"(u), "v"(v) : "vcc");
        } else {
#pragma unroll
            for (int i = 0; i < CHAIN / 8; i++)
                asm volatile(
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1"
                    : "+v"(a), "+v"(b) : "v"(u), "v"(v) : "vcc");
        }
        u += (uint32_t)a; v ^= (uint32_t)b;
    }
    if (pad[0] == 77) out[0] = 1;
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a ^ b) + u + v;
}
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 5;   // waves per SIMD wanted: a block of 4 waves = 1 per SIMD
    const size_t lds = waves >= 8 ? 0 : (size_t)(160 * 1024 / waves) - 1024;
    uint32_t* d;
    const int blocks = 256 * 8 * 4;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), lds, 0, d, 200, 3u, 5u);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), lds, 0, d, 200, 3u, 5u);
            else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), lds, 0, d, 200, 3u, 5u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double mads = (double)blocks * 256 * 200 * 2 * CHAIN;
            if (rep) printf("waves/SIMD %d mode %s: %.3f ms, %.1f G mad/s (lane), %.2f cycles per wave-mad per SIMD at 2.4 GHz\n", waves, mode == 2 ? "one asm block, ONE dependent chain" : mode ? "one asm block (no s_nop)" : "asm per instruction (s_nop 0 each)",
                            ms, mads / ms / 1e6, 2.4e9 * 1024.0 * (ms * 1e-3) / (mads / 64));
        }
    }
    return 0;
}
