// What do FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for GATHERS?  (verdict r03, weak 7)
// The guide calibrates FETCH_SIZE for 16-byte-per-lane coalesced streams only (it reads exactly half the bytes on gfx950) and says
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The MSM's level-0 kernel
// gathers two 64-byte affine points per lane from a 64 MB table that lives in the Infinity Cache; the sumcheck kernels read 32-byte
// field elements.  This program runs kernels with KNOWN byte counts -- a coalesced 16 B / lane stream (the guide's case, as the
// control), and random 32-byte and 64-byte gathers from a 64 MB and a 1 GiB table -- each under its own kernel name; run it once
// under `rocprofv3 --pmc FETCH_SIZE` and once under `--pmc WRITE_SIZE`, and scripts/pmc_gather_factor.py divides the known bytes by
// what the counters say.  It prints the known byte counts as one JSON line.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// control: every lane reads 16 contiguous bytes, writes 16
__global__ void __launch_bounds__(256) k_stream16(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 v = in[i];
    v.x ^= v.w;
    out[i] = v;
}

// every lane reads ENTRY bytes (32 or 64) at a pseudo-random entry of the table, writes 16 bytes coalesced
template <int ENTRY, int TABLE_LOG>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, uint4* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t entries = (1ull << TABLE_LOG) / ENTRY;
    const uint64_t e = mix(i) & (entries - 1);
    const uint4* p = table + e * (ENTRY / 16);
    uint4 acc = p[0];
#pragma unroll
    for (int k = 1; k < ENTRY / 16; k++) {
        const uint4 v = p[k];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    out[i] = acc;
}

// the MSM level-0 shape: indices come from MEMORY (4 bytes per lane, coalesced), two 64-byte entries per lane, 96 bytes written
template <int TABLE_LOG>
__global__ void __launch_bounds__(256) k_gather_pairs64(const uint4* __restrict__ table, const uint32_t* __restrict__ idx,
                                                         uint4* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = idx[2 * i], b = idx[2 * i + 1];
    const uint4* p = table + (uint64_t)a * 4;
    const uint4* q = table + (uint64_t)b * 4;
    uint4 r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 u = p[k], v = q[k];
        r[k].x = u.x ^ v.x; r[k].y = u.y + v.y; r[k].z = u.z ^ v.w; r[k].w = u.w + v.z;
    }
    // 96 bytes per lane, as three column stores of 32 bytes
    for (int c = 0; c < 3; c++) {
        out[(uint64_t)c * 2 * n + 2 * i] = r[c];
        out[(uint64_t)c * 2 * n + 2 * i + 1] = r[(c + 1) & 3];
    }
}
// memory-only model of a level-0 launch with a table of 2^20 points of ENTRY bytes each (64: canonical x, y; 72: x, y in the 36-byte
// 9 x 29 cell form; 144: x, y, x y, d x y in that form): two gathered points per lane, 108 bytes written (3 x 36)
template <int ENTRY>
__global__ void __launch_bounds__(256) k_level0_model(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx,
                                                       uint32_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = idx[2 * i] & 0xfffffu, b = idx[2 * i + 1] & 0xfffffu;
    const uint32_t* p = table + (uint64_t)a * (ENTRY / 4);
    const uint32_t* q = table + (uint64_t)b * (ENTRY / 4);
    uint32_t acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < ENTRY / 4; k += 4) {
        uint32_t u[4], v[4];
        __builtin_memcpy(u, p + k, 16);
        __builtin_memcpy(v, q + k, 16);
#pragma unroll
        for (int e = 0; e < 4; e++) acc[(k + e) % 9] += u[e] ^ (v[e] * 3u);
    }
    if (ENTRY % 16) {   // 72 = 4 x 16 + 8
        acc[7] += p[ENTRY / 4 - 2] ^ q[ENTRY / 4 - 2];
        acc[8] += p[ENTRY / 4 - 1] ^ q[ENTRY / 4 - 1];
    }
    for (int c = 0; c < 3; c++) {
        uint32_t* o = out + ((uint64_t)c * n + i) * 9;
#pragma unroll
        for (int e = 0; e < 9; e++) o[e] = acc[e] + c;
    }
}

__global__ void __launch_bounds__(256) k_fill_idx(uint32_t* idx, uint64_t n, uint32_t entries) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)(mix(i * 7 + 1) % entries);
}
__global__ void __launch_bounds__(256) k_fill(uint4* p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((uint32_t)i, (uint32_t)(i >> 3), 7u, (uint32_t)i * 3u);
}

int main() {
    const uint64_t N = 1ull << 24;           // lanes per launch
    uint4 *table, *out;
    uint32_t* idx;
    CK(hipMalloc(&table, 1ull << 30));
    CK(hipMalloc(&out, N * 96));
    CK(hipMalloc(&idx, 2 * N * 4));
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, table, (1ull << 30) / 16);
    hipLaunchKernelGGL(k_fill_idx, dim3((2 * N + 255) / 256), dim3(256), 0, 0, idx, 2 * N, (1u << 26) / 64);
    CK(hipDeviceSynchronize());
    const dim3 g((unsigned)(N / 256)), b(256);
    for (int rep = 0; rep < 3; rep++) {      // first repetition warms the caches; the summary takes the last
        hipLaunchKernelGGL(k_stream16, g, b, 0, 0, table, out, N);
        hipLaunchKernelGGL((k_gather<32, 26>), g, b, 0, 0, table, out, N);
        hipLaunchKernelGGL((k_gather<64, 26>), g, b, 0, 0, table, out, N);
        hipLaunchKernelGGL((k_gather<32, 30>), g, b, 0, 0, table, out, N);
        hipLaunchKernelGGL((k_gather<64, 30>), g, b, 0, 0, table, out, N);
        hipLaunchKernelGGL((k_gather_pairs64<26>), g, b, 0, 0, table, idx, out, N);
        CK(hipDeviceSynchronize());
    }
    // memory-only timing of the level-0 access pattern for three point-table formats (HIP events; not part of the counter runs' summary)
    float ms_model[3] = {0, 0, 0};
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        uint32_t* out9;
        CK(hipMalloc(&out9, N * 108));
        for (int v = 0; v < 3; v++) {
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, 0));
                if (v == 0) hipLaunchKernelGGL((k_level0_model<64>), g, b, 0, 0, (const uint32_t*)table, idx, out9, N);
                if (v == 1) hipLaunchKernelGGL((k_level0_model<72>), g, b, 0, 0, (const uint32_t*)table, idx, out9, N);
                if (v == 2) hipLaunchKernelGGL((k_level0_model<144>), g, b, 0, 0, (const uint32_t*)table, idx, out9, N);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_model[v], e0, e1));
            }
        }
        fprintf(stderr, "level-0 memory model, 2^24 lanes, two gathered points + 108 B written per lane: 64 B points %.3f ms, 72 B %.3f ms, 144 B %.3f ms\n",
                ms_model[0], ms_model[1], ms_model[2]);
    }
    printf("{\"lanes\": %llu, \"level0_memory_model_ms\": {\"64\": %.4f, \"72\": %.4f, \"144\": %.4f}, \"kernels\": {"
           "\"k_stream16\": {\"read\": %llu, \"written\": %llu}, "
           "\"k_gather<32, 26>\": {\"read\": %llu, \"written\": %llu, \"table_MB\": 64}, "
           "\"k_gather<64, 26>\": {\"read\": %llu, \"written\": %llu, \"table_MB\": 64}, "
           "\"k_gather<32, 30>\": {\"read\": %llu, \"written\": %llu, \"table_MB\": 1024}, "
           "\"k_gather<64, 30>\": {\"read\": %llu, \"written\": %llu, \"table_MB\": 1024}, "
           "\"k_gather_pairs64<26>\": {\"read\": %llu, \"written\": %llu, \"table_MB\": 64, \"note\": \"the MSM level-0 shape: 2 x 64 B gathered + 8 B of indices read, 96 B written per lane\"}}}\n",
           (unsigned long long)N, ms_model[0], ms_model[1], ms_model[2], (unsigned long long)(N * 16), (unsigned long long)(N * 16), (unsigned long long)(N * 32),
           (unsigned long long)(N * 16), (unsigned long long)(N * 64), (unsigned long long)(N * 16), (unsigned long long)(N * 32),
           (unsigned long long)(N * 16), (unsigned long long)(N * 64), (unsigned long long)(N * 16), (unsigned long long)(N * 136),
           (unsigned long long)(N * 96));
    return 0;
}
