// Stable radix sort of (key, value) u32 pairs and a plain exclusive scan, for the G1 "sum by key" engine (g1.hip): the
// histogram / scan / ballot-ranked scatter scheme of the bucket sort in msm.hip, run once per 8-bit digit (LSD).
// A workgroup owns a tile of RS_TILE consecutive items.  Its four waves rank their quarter of the tile in input order (equal
// digits found with 8 ballots per 64 items), the tile is regrouped by digit in LDS, and written out digit by digit: items of one
// digit leave as one run of consecutive words (16 items on average), so the scatter writes whole cache lines instead of 4 bytes
// per line.  Equal keys keep their input order: the association order of the point sums that follow -- and with it every
// intermediate Jacobian representative -- is a function of the input alone.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace gm {

static constexpr uint32_t RS_WAVES = 4;                    // waves per workgroup
static constexpr uint32_t RS_PER_WAVE = 1024;              // items one wave ranks (16 steps of 64)
static constexpr uint32_t RS_TILE = RS_WAVES * RS_PER_WAVE;
static constexpr uint32_t RS_BITS = 8, RS_BINS = 1u << RS_BITS;

// hist[d * ntiles + t] = number of items of tile t whose digit is d
__global__ void __launch_bounds__(64 * RS_WAVES) k_rs_hist(const uint32_t* __restrict__ keys, uint64_t n, uint32_t shift, uint32_t ntiles,
                                                          uint32_t* __restrict__ hist) {
    __shared__ uint32_t cnt[RS_BINS];
    const uint32_t tid = threadIdx.x;
    cnt[tid] = 0;
    __syncthreads();
    const uint64_t x0 = (uint64_t)blockIdx.x * RS_TILE;
    for (uint32_t i = tid; i < RS_TILE; i += 64 * RS_WAVES) {
        const uint64_t x = x0 + i;
        if (x < n) atomicAdd(&cnt[(keys[x] >> shift) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    hist[(uint64_t)tid * ntiles + blockIdx.x] = cnt[tid];
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

// scatter of one digit pass; base[d * ntiles + t] = first output slot of (digit d, tile t)
__global__ void __launch_bounds__(64 * RS_WAVES) k_rs_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint64_t n,
                                                             uint32_t shift, uint32_t ntiles, const uint32_t* __restrict__ base) {
    __shared__ uint32_t cnt[RS_WAVES][RS_BINS];   // per wave: running count of every digit, then its first slot in the tile
    __shared__ uint32_t dig_start[RS_BINS + 1];   // first slot of every digit in the regrouped tile
    __shared__ uint32_t gbase[RS_BINS];
    __shared__ uint32_t st_key[RS_TILE], st_val[RS_TILE];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr uint32_t STEPS = RS_PER_WAVE / 64;
    for (uint32_t i = lane; i < RS_BINS; i += 64) cnt[wave][i] = 0;
    if (base) gbase[tid] = base[(uint64_t)tid * ntiles + blockIdx.x];
    __builtin_amdgcn_wave_barrier();
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t x0 = (uint64_t)blockIdx.x * RS_TILE + (uint64_t)wave * RS_PER_WAVE;
    uint32_t k[STEPS], v[STEPS], rank[STEPS];
#pragma unroll
    for (uint32_t st = 0; st < STEPS; st++) {
        const uint64_t x = x0 + st * 64 + lane;
        const bool valid = x < n;
        k[st] = valid ? keys[x] : 0xffffffffu;
        v[st] = valid ? vals[x] : 0u;
    }
#pragma unroll
    for (uint32_t st = 0; st < STEPS; st++) {
        const bool valid = x0 + st * 64 + lane < n;
        const uint32_t dg = (k[st] >> shift) & (RS_BINS - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < RS_BITS; b++) {
            const uint64_t m = __ballot((dg >> b) & 1u);
            peers &= ((dg >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = __popcll(peers & lane_lt);
        const uint32_t pos = cnt[wave][dg] + before;
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers >> lane) == 1ull) cnt[wave][dg] = pos + 1;   // the highest peer lane publishes the new count
        __builtin_amdgcn_wave_barrier();
        rank[st] = valid ? pos : 0xffffffffu;
    }
    __syncthreads();
    // thread d: slots of digit d in the regrouped tile, wave after wave (stable), after all smaller digits
    {
        uint32_t c[RS_WAVES], tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < RS_WAVES; w++) { c[w] = cnt[w][tid]; tot += c[w]; }
        // exclusive scan of tot over the 256 digits: wave scan + the four wave totals
        uint32_t inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if ((int)lane >= d) inc += t;
        }
        __shared__ uint32_t wtot[RS_WAVES];
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t run = inc - tot;
        for (uint32_t w = 0; w < wave; w++) run += wtot[w];
        dig_start[tid] = run;
        if (!base) gbase[tid] = run;   // a single tile is its own histogram: slot of digit d = its slot in the regrouped tile
        if (tid == RS_BINS - 1) dig_start[RS_BINS] = run + tot;
#pragma unroll
        for (uint32_t w = 0; w < RS_WAVES; w++) { cnt[w][tid] = run; run += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t st = 0; st < STEPS; st++) {
        if (rank[st] != 0xffffffffu) {
            const uint32_t dg = (k[st] >> shift) & (RS_BINS - 1);
            const uint32_t p = cnt[wave][dg] + rank[st];
            st_key[p] = k[st];
            st_val[p] = v[st];
        }
    }
    __syncthreads();
    const uint32_t total = dig_start[RS_BINS];
    for (uint32_t p = tid; p < total; p += 64 * RS_WAVES) {
        const uint32_t key = st_key[p];
        const uint32_t dg = (key >> shift) & (RS_BINS - 1);
        const uint32_t pos = gbase[dg] + (p - dig_start[dg]);
        keys_out[pos] = key;
        vals_out[pos] = st_val[p];
    }
}

// bytes of histogram scratch for n items
static inline size_t radix_sort_tmp_bytes(uint64_t n) {
    const uint64_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    const uint64_t entries = (uint64_t)RS_BINS * (ntiles ? ntiles : 1) + 1;
    return (size_t)entries * sizeof(uint32_t) + scan_tmp_bytes(entries);
}

// Sorts n pairs by the low `bits` bits of the key, stable.  (keys_a, vals_a) hold the input and are clobbered; the result is
// in (*keys_res, *vals_res), which point at the a or the b buffers depending on the number of passes.
static inline hipError_t radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint64_t n,
                                              uint32_t bits, uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t s) {
    uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    if (n) {
        const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
        for (uint32_t shift = 0; shift < (bits ? bits : 1); shift += RS_BITS) {
            if (ntiles > 1) {
                hipLaunchKernelGGL(k_rs_hist, dim3(ntiles), dim3(64 * RS_WAVES), 0, s, ki, n, shift, ntiles, tmp);
                exclusive_scan_u32(tmp, tmp, (uint64_t)RS_BINS * ntiles, 0, tmp + (uint64_t)RS_BINS * ntiles + 1, s);
            }
            hipLaunchKernelGGL(k_rs_scatter, dim3(ntiles), dim3(64 * RS_WAVES), 0, s, ki, vi, ko, vo, n, shift, ntiles,
                               ntiles > 1 ? (const uint32_t*)tmp : (const uint32_t*)nullptr);
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
