// Instruction-rate microbenchmark for the integer / fp64 ops a 256-bit modular multiplier can be built from.
// Each kernel runs ITER iterations of UNROLL independent chains per lane; all 256 CUs, 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ITER 4096
#define CHAINS 8

#define KERNEL(name, decl, body, fin)                                                                  \
    __global__ void name(uint64_t* out, uint32_t seed) {                                               \
        decl;                                                                                          \
        for (int it = 0; it < ITER; it++) { body; }                                                    \
        fin;                                                                                           \
    }

__global__ void k_mad64(uint64_t* out, uint32_t seed) {
    uint64_t acc[CHAINS];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(uint64_t* out, uint32_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = seed * 3 + 1;
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(uint64_t* out, uint32_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = seed * 3 + 0x80000001u;
    for (int c = 0; c < CHAINS; c++) acc[c] = (c + threadIdx.x + seed) | 0x80000000u;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(uint64_t* out, uint32_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = (seed * 3 + 1) & 0xffffff;
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[c]) : "v"(b));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_addco(uint64_t* out, uint32_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = seed * 3 + 1;
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[c]) : "v"(b) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add64(uint64_t* out, uint32_t seed) {
    uint64_t acc[CHAINS];
    uint64_t b = seed * 3 + 1;
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(b));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(uint64_t* out, uint32_t seed) {
    double acc[CHAINS];
    double b = 1.0 + seed * 1e-9, d = 1e-3;
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(b), "v"(d));
    }
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_fma32(uint64_t* out, uint32_t seed) {
    float acc[CHAINS];
    float b = 1.0f + seed * 1e-9f, d = 1e-3f;
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(b), "v"(d));
    }
    float s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_mov(uint64_t* out, uint32_t seed) {
    uint32_t acc[CHAINS], t[CHAINS];
    for (int c = 0; c < CHAINS; c++) { acc[c] = c + threadIdx.x + seed; t[c] = 0; }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_mov_b32 %0, %1" : "=v"(t[c]) : "v"(acc[c]));
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[c]) : "v"(t[c]));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_shr64(uint64_t* out, uint32_t seed) {
    uint64_t acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = ((uint64_t)(c + threadIdx.x + seed) << 40) | 12345;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[c]));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_alignbit(uint64_t* out, uint32_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = seed * 3 + 1;
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(acc[c]) : "v"(b));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_and(uint64_t* out, uint32_t seed) {
    uint32_t acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(acc[c]));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// dependent chain of multiply-adds (one accumulator): what the compiler pads with s_nop
__global__ void k_mad64_dep(uint64_t* out, uint32_t seed) {
    uint64_t acc = threadIdx.x;
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename K>
static void run(const char* name, K k, int ops_per_iter, uint64_t* d_out) {
    const int blocks = 256 * 8, threads = 256;  // 8 blocks of 4 waves per CU = 8 waves / SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * (threads / 64) * ITER * ops_per_iter;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
    printf("%-10s %8.3f ms  %7.2f G wave-instr/s  => %6.2f cycles/wave-instr/SIMD @2.4GHz (%.1f Tlane-op/s)\n", name, ms,
           wave_instr / (ms * 1e-3) / 1e9, 2.4e9 / per_simd_per_s, wave_instr * 64 / (ms * 1e-3) / 1e12);
}

int main() {
    uint64_t* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * 8);
    run("mad_u64_u32", k_mad64, CHAINS, d_out);
    run("mul_lo_u32", k_mullo, CHAINS, d_out);
    run("mul_hi_u32", k_mulhi, CHAINS, d_out);
    run("mad_u32_u24", k_mad24, CHAINS, d_out);
    run("addc_co_u32", k_addco, CHAINS, d_out);
    run("lshl_add_u64", k_add64, CHAINS, d_out);
    run("fma_f64", k_fma64, CHAINS, d_out);
    run("fma_f32", k_fma32, CHAINS, d_out);
    run("mov_b32", k_mov, 2 * CHAINS, d_out);
    run("lshrrev_b64", k_shr64, CHAINS, d_out);
    run("alignbit_b32", k_alignbit, CHAINS, d_out);
    run("and_b32 lit", k_and, CHAINS, d_out);
    run("mad64 dep+nop", k_mad64_dep, CHAINS, d_out);
    return 0;
}
