// Can the host store straight into device memory (large BAR), and how fast does a polling kernel see it compared with polling
// pinned host memory over PCIe?  Development probe for the challenge hand-off of the persistent round kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <x86intrin.h>

__global__ void k_pingpong(volatile uint32_t* flag_in, volatile uint32_t* flag_out, int rounds, uint64_t* ticks) {
    uint64_t t0 = wall_clock64();
    for (int r = 1; r <= rounds; r++) {
        // tell the host
        __hip_atomic_store((uint32_t*)flag_out, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // wait for the answer
        while (__hip_atomic_load((uint32_t*)flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (uint32_t)r) __builtin_amdgcn_s_sleep(1);
    }
    *ticks = wall_clock64() - t0;
}

static double run(volatile uint32_t* d_in_devptr, volatile uint32_t* h_in_hostptr, volatile uint32_t* d_out, volatile uint32_t* h_out, int rounds) {
    uint64_t* d_ticks;
    hipMalloc(&d_ticks, 8);
    *h_in_hostptr = 0; *h_out = 0;
    hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(1), 0, 0, d_in_devptr, d_out, rounds, d_ticks);
    for (int r = 1; r <= rounds; r++) {
        while (*h_out != (uint32_t)r) _mm_pause();
        *h_in_hostptr = (uint32_t)r;
        _mm_sfence();
    }
    hipDeviceSynchronize();
    uint64_t t;
    hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost);
    return t / 100.0 / rounds;   // us per round trip (100 MHz clock)
}

int main() {
    int large_bar = 0;
    hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    uint32_t *h_out, *h_in;
    hipHostMalloc(&h_out, 64, hipHostMallocMapped);
    hipHostMalloc(&h_in, 64, hipHostMallocMapped);
    printf("pinned host memory both ways : %.2f us per round trip\n", run(h_in, h_in, h_out, h_out, 2000));
    if (large_bar) {
        uint32_t* d_in = nullptr;
        hipError_t e = hipExtMallocWithFlags((void**)&d_in, 4096, hipDeviceMallocFinegrained);
        printf("hipExtMallocWithFlags(finegrained): %s ptr %p\n", hipGetErrorString(e), (void*)d_in);
        if (e == hipSuccess) {
            hipMemset(d_in, 0, 64);
            hipDeviceSynchronize();
            fflush(stdout);
            // host store straight into device memory
            printf("device memory written by the host: %.2f us per round trip\n", run(d_in, d_in, h_out, h_out, 2000));
        }
    }
    return 0;
}
