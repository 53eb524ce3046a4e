// Does a peer process see, through a HIP IPC mapping, what the owner of an allocation has written and synchronised?
// W processes share ONE device.  Every iteration: each process fills its buffer with a pattern of (rank, iteration) by a kernel,
// synchronises its stream, meets the others at a barrier in shared host memory, copies the NEXT rank's buffer through its IPC
// mapping (a kernel copy), checks the copy against the pattern on the device, meets the others again.  Printed: mismatching words.
// With the runtime's default of four hardware queues per process the count stays 0; the question is what happens when every
// process also holds many streams under GPU_MAX_HW_QUEUES=16 ("An oversubscribed device").
//   hipcc -O2 --offload-arch=gfx950 -o ipc_visibility ipc_visibility.hip
//   GPU_MAX_HW_QUEUES=16 ./ipc_visibility [processes 4] [extra streams 12] [iterations 200] [MiB 64]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#define CHECK(c)                                                                                         \
    do {                                                                                                 \
        hipError_t e_ = (c);                                                                             \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #c, hipGetErrorString(e_)); _exit(2); } \
    } while (0)

struct Shared {
    volatile uint32_t arrive, generation;
    hipIpcMemHandle_t handle[16];
    volatile uint64_t bad[16];
};

static void barrier(Shared* sh, uint32_t world) {
    const uint32_t gen = __atomic_load_n(&sh->generation, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&sh->arrive, 1, __ATOMIC_ACQ_REL) == world) {
        __atomic_store_n(&sh->arrive, 0, __ATOMIC_RELAXED);
        __atomic_add_fetch(&sh->generation, 1, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&sh->generation, __ATOMIC_ACQUIRE) == gen) __builtin_ia32_pause();
    }
}

__device__ __forceinline__ uint64_t pattern(uint32_t rank, uint32_t it, uint64_t i) {
    uint64_t z = i * 0x9e3779b97f4a7c15ull + ((uint64_t)rank << 48) + ((uint64_t)it << 24) + 1;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    return z ^ (z >> 27);
}
__global__ void k_fill(uint64_t* p, uint64_t n, uint32_t rank, uint32_t it) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = pattern(rank, it, i);
}
__global__ void k_copy(const uint64_t* s, uint64_t* d, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void k_check(const uint64_t* p, uint64_t n, uint32_t rank, uint32_t it, unsigned long long* bad) {
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += p[i] != pattern(rank, it, i);
    if (c) atomicAdd(bad, c);
}
__global__ void k_touch(uint32_t* p) { if (threadIdx.x == 0) p[0] += 1; }

int main(int argc, char** argv) {
    const uint32_t world = argc > 1 ? atoi(argv[1]) : 4, extra = argc > 2 ? atoi(argv[2]) : 12, iters = argc > 3 ? atoi(argv[3]) : 200;
    const uint64_t n = (uint64_t)(argc > 4 ? atoi(argv[4]) : 64) << 17;   // 64-bit words
    Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    memset((void*)sh, 0, sizeof(Shared));
    for (uint32_t rank = 0; rank < world; rank++) {
        if (fork() != 0) continue;
        // ---- a rank (the HIP runtime starts here, after the fork)
        CHECK(hipSetDevice(0));
        hipStream_t s;
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        uint32_t* d_t;
        CHECK(hipMalloc((void**)&d_t, 4096));
        CHECK(hipMemset(d_t, 0, 4096));
        hipStream_t* more = (hipStream_t*)calloc(extra ? extra : 1, sizeof(hipStream_t));
        for (uint32_t k = 0; k < extra; k++) {   // streams with work on them: the hardware queues a bench process holds after its other legs
            CHECK(hipStreamCreateWithFlags(&more[k], hipStreamNonBlocking));
            hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, more[k], d_t + 16 * k);
        }
        CHECK(hipDeviceSynchronize());
        uint64_t *x, *y;
        unsigned long long* d_bad;
        CHECK(hipMalloc((void**)&x, n * 8));
        CHECK(hipMalloc((void**)&y, n * 8));
        CHECK(hipMalloc((void**)&d_bad, 8));
        CHECK(hipMemset(d_bad, 0, 8));
        CHECK(hipIpcGetMemHandle(&sh->handle[rank], x));
        barrier(sh, world);
        const uint32_t peer = (rank + 1) % world;
        void* px = nullptr;
        CHECK(hipIpcOpenMemHandle(&px, sh->handle[peer], hipIpcMemLazyEnablePeerAccess));
        unsigned long long total = 0, bad_iters = 0;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (uint32_t it = 0; it < iters; it++) {
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, s, x, n, rank, it);
            CHECK(hipStreamSynchronize(s));
            barrier(sh, world);                              // every source is complete
            hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s, (const uint64_t*)px, y, n);
            hipLaunchKernelGGL(k_check, dim3(2048), dim3(256), 0, s, y, n, peer, it, d_bad);
            unsigned long long b = 0;
            CHECK(hipMemcpyAsync(&b, d_bad, 8, hipMemcpyDeviceToHost, s));
            CHECK(hipStreamSynchronize(s));
            if (b != total) { bad_iters++; total = b; }
            barrier(sh, world);                              // everybody is done reading
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("rank %u: %llu mismatching words in %llu of %u iterations (reading rank %u's %llu MiB through its IPC mapping), %.2f ms per iteration\n", rank,
               total, bad_iters, iters, peer, (unsigned long long)(n >> 17), ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / iters);
        sh->bad[rank] = total;
        fflush(stdout);
        _exit(total ? 1 : 0);
    }
    int rc = 0;
    for (uint32_t k = 0; k < world; k++) {
        int st = 0;
        wait(&st);
        if (!WIFEXITED(st) || WEXITSTATUS(st) > 1) rc = 2;
        else if (WEXITSTATUS(st) == 1 && rc == 0) rc = 1;
    }
    printf("%s\n", rc == 0 ? "every copy matched" : rc == 1 ? "MISMATCHES (see above)" : "a rank failed");
    return rc;
}
