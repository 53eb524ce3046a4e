// Stable radix sort of (key, value) u32 pairs and a plain exclusive scan, for the G1 "sum by key" engine (g1.hip): the
// histogram / chunk-scan / ballot-ranked scatter scheme of the bucket sort in msm.hip, run once per RS_BITS-bit digit (LSD).
// Digits are narrow on purpose: a wave then appends to 2^RS_BITS output runs only, and the lines it is filling stay in L2 until
// they are full (with 8-bit digits every resident wave keeps 256 partial lines open and the scatter writes 4 bytes per line).
// One wave walks one chunk of RS_CHUNK items in order, so equal keys keep their input order: the association order of the
// point sums below -- and with it every intermediate Jacobian representative -- is a function of the input alone.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace gm {

static constexpr uint32_t RS_CHUNK = 4096;  // items per wave
static constexpr uint32_t RS_WAVES = 4;     // waves per workgroup
#ifndef GM_RS_BITS
#define GM_RS_BITS 5
#endif
static constexpr uint32_t RS_BITS = GM_RS_BITS, RS_BINS = 1u << RS_BITS;

// hist[d * nchunks + c] = number of items of chunk c whose digit is d
__global__ void __launch_bounds__(64 * RS_WAVES) k_rs_hist(const uint32_t* __restrict__ keys, uint64_t n, uint32_t shift, uint32_t nchunks,
                                                          uint32_t* __restrict__ hist) {
    __shared__ uint32_t cnt_all[RS_WAVES][RS_BINS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* cnt = cnt_all[wave];
    const uint64_t c = (uint64_t)blockIdx.x * RS_WAVES + wave;
    if (c >= nchunks) return;  // whole wave; only wave-level barriers below
    for (uint32_t i = lane; i < RS_BINS; i += 64) cnt[i] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint64_t x0 = c * RS_CHUNK;
    for (uint32_t i = lane; i < RS_CHUNK; i += 64) {
        const uint64_t x = x0 + i;
        if (x < n) atomicAdd(&cnt[(keys[x] >> shift) & (RS_BINS - 1)], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < RS_BINS; i += 64) hist[(uint64_t)i * nchunks + c] = cnt[i];
}

// exclusive scan of `n` counters in place (+ the total at v[n] when with_total); one 1024-thread workgroup
__global__ void __launch_bounds__(1024) k_exclusive_scan_u32(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n,
                                                             int with_total) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t per = (n + 1023) / 1024;
    const uint64_t r0 = (uint64_t)tid * per;
    const uint64_t r1 = (r0 + per < n) ? r0 + per : n;
    uint32_t sum = 0;
    for (uint64_t r = r0; r < r1; r++) sum += src[r];
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += wave_tot[w];
    uint32_t run = base + inc - sum;
    for (uint64_t r = r0; r < r1; r++) {
        const uint32_t v = src[r];   // read before the write: src may be dst
        dst[r] = run;
        run += v;
    }
    if (with_total && tid == 1023) {
        uint32_t tot = 0;
        for (int w = 0; w < 16; w++) tot += wave_tot[w];
        dst[n] = tot;
    }
}

// The same scan for long inputs, two launches: every 1024-thread workgroup scans its tile of SCAN_TILE counters and publishes
// the tile total; the second launch adds, to every tile, the sum of the totals before it (each workgroup adds them up itself:
// there are at most n / SCAN_TILE of them).
static constexpr uint32_t SCAN_ITEMS = 16, SCAN_TILE = 1024 * SCAN_ITEMS;
__global__ void __launch_bounds__(1024) k_scan_tiles(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n,
                                                     uint32_t* __restrict__ tile_tot) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)tid * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (r0 + k < n) ? src[r0 + k] : 0u;
        sum += v[k];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < 16; w++) { const uint32_t t = wave_tot[w]; tot += t; if (w < wave) base += t; }
    uint32_t run = base + inc - sum;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
        if (r0 + k < n) dst[r0 + k] = run;
        run += v[k];
    }
    if (tid == 0) tile_tot[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) k_scan_fix(uint32_t* __restrict__ dst, uint64_t n, const uint32_t* __restrict__ tile_tot,
                                                   uint32_t ntiles, int with_total) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    uint32_t part = 0;
    const uint32_t upto = (with_total && b + 1 == ntiles) ? ntiles : b;   // the last tile also needs the grand total
    uint32_t mine_all = 0;
    for (uint32_t i = tid; i < upto; i += 1024) { const uint32_t t = tile_tot[i]; mine_all += t; if (i < b) part += t; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { part += __shfl_down(part, d, 64); mine_all += __shfl_down(mine_all, d, 64); }
    __shared__ uint32_t wave_all[16];
    if (lane == 0) { wave_tot[wave] = part; wave_all[wave] = mine_all; }
    __syncthreads();
    uint32_t off = 0, all = 0;
    for (uint32_t w = 0; w < 16; w++) { off += wave_tot[w]; all += wave_all[w]; }
    const uint64_t r0 = (uint64_t)b * SCAN_TILE + (uint64_t)tid * SCAN_ITEMS;
    if (off) {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_ITEMS; k++)
            if (r0 + k < n) dst[r0 + k] += off;
    }
    if (with_total && b + 1 == ntiles && tid == 0) dst[n] = all;
}

// exclusive scan of n counters (src may be dst); tile_tot: scratch of ceil(n / SCAN_TILE) words
static inline void exclusive_scan_u32(const uint32_t* src, uint32_t* dst, uint64_t n, int with_total, uint32_t* tile_tot, hipStream_t s) {
    if (n <= SCAN_TILE) {
        hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(1), dim3(1024), 0, s, src, dst, n, with_total);
        return;
    }
    const uint32_t ntiles = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(1024), 0, s, src, dst, n, tile_tot);
    hipLaunchKernelGGL(k_scan_fix, dim3(ntiles), dim3(1024), 0, s, dst, n, tile_tot, ntiles, with_total);
}
static inline size_t scan_tmp_bytes(uint64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(uint32_t); }

// scatter of one digit pass; base[d * nchunks + c] = first output slot of (digit d, chunk c)
__global__ void __launch_bounds__(64 * RS_WAVES) k_rs_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint64_t n,
                                                             uint32_t shift, uint32_t nchunks, const uint32_t* __restrict__ base) {
    __shared__ uint32_t cnt_all[RS_WAVES][RS_BINS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* cnt = cnt_all[wave];
    const uint64_t c = (uint64_t)blockIdx.x * RS_WAVES + wave;
    if (c >= nchunks) return;
    for (uint32_t i = lane; i < RS_BINS; i += 64) cnt[i] = base[(uint64_t)i * nchunks + c];
    __builtin_amdgcn_wave_barrier();
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t x0 = c * RS_CHUNK;
    for (uint32_t i = 0; i < RS_CHUNK; i += 64) {
        const uint64_t x = x0 + i + lane;
        const bool valid = x < n;
        const uint32_t k = valid ? keys[x] : 0u;
        const uint32_t dg = (k >> shift) & (RS_BINS - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < RS_BITS; b++) {
            const uint64_t m = __ballot((dg >> b) & 1u);
            peers &= ((dg >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = __popcll(peers & lane_lt);
        const uint32_t pos = cnt[dg] + before;
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers >> lane) == 1ull) cnt[dg] = pos + 1;   // the highest peer lane publishes the new count
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            keys_out[pos] = k;
            vals_out[pos] = vals[x];
        }
    }
}

// bytes of histogram scratch for n items
static inline size_t radix_sort_tmp_bytes(uint64_t n) {
    const uint64_t nchunks = (n + RS_CHUNK - 1) / RS_CHUNK;
    const uint64_t entries = (uint64_t)RS_BINS * (nchunks ? nchunks : 1) + 1;
    return (size_t)entries * sizeof(uint32_t) + scan_tmp_bytes(entries);
}

// Sorts n pairs by the low `bits` bits of the key, stable.  (keys_a, vals_a) hold the input and are clobbered; the result is
// in (*keys_res, *vals_res), which point at the a or the b buffers depending on the number of passes.
static inline hipError_t radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint64_t n,
                                              uint32_t bits, uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t s) {
    uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    if (n) {
        const uint32_t nchunks = (uint32_t)((n + RS_CHUNK - 1) / RS_CHUNK);
        const unsigned blocks = (nchunks + RS_WAVES - 1) / RS_WAVES;
        for (uint32_t shift = 0; shift < (bits ? bits : 1); shift += RS_BITS) {
            hipLaunchKernelGGL(k_rs_hist, dim3(blocks), dim3(64 * RS_WAVES), 0, s, ki, n, shift, nchunks, tmp);
            exclusive_scan_u32(tmp, tmp, (uint64_t)RS_BINS * nchunks, 0, tmp + (uint64_t)RS_BINS * nchunks + 1, s);
            hipLaunchKernelGGL(k_rs_scatter, dim3(blocks), dim3(64 * RS_WAVES), 0, s, ki, vi, ko, vo, n, shift, nchunks, tmp);
            uint32_t* t;
            t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
    }
    *keys_res = ki;
    *vals_res = vi;
    return hipGetLastError();
}

}  // namespace gm
