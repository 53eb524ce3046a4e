// BLS12-381 G1 side of the hot path on the device.
//
// Everything the reference does with G1 on this path is "add up base points grouped by a small integer key", then a
// weighted sum of the groups:
//   PushForwardState::new       d_outer[y][digit] += SRS[..], c_outer[y][counter] += SRS[..]   pushforward.rs:395-456
//                               d_comm / c_comm = sum_i i * bucket_i                            pushforward.rs:504-524
//   msm_bigint_(wnaf_)nonaff    window buckets, running-sum reduce, window recombination        msm_nonaffine.rs:89-272
//   KzgProvingKey::commit       <G1 as VariableBaseMSM>::msm                                     commitments/kzg.rs:123-126
//   binary_msm                  sum of table entries selected by gamma-bit chunks               binary_msm.rs:19-29
//   Pullback::bucketed_msm      buckets[mapping[i]] += bases[i], then msm_nonaff                pullback.rs:27-59
// One engine serves all of them: sort (key, point index) pairs by key (rocPRIM radix sort), find the row boundaries,
// and reduce every row with a flat pairwise tree -- level l adds cells (2i, 2i+1) of every row, the thread -> (row, cell)
// mapping goes through the per-level row offsets, so the work is perfectly balanced whatever the bucket skew (d_outer
// has 256 rows of 4096 points per window, an MSM window has 8192 rows of ~128).  Weighted sums sum_i i*B_i use the
// same engine on the bit decomposition of i:  sum_i i B_i = sum_b 2^b (sum_{i : bit b set} B_i).
// G1 results are group elements (g1.hip.h): the tree order is free, unlike the Bandersnatch bucket sums of msm.hip whose
// projective coordinates are part of the proof.
//
// Bound: integer VALU (a Jacobian addition is 16 Fq multiplications of ~290 v_mad_u64_u32 each); HBM traffic is
// ~0.4 KB per input point.
#include <string.h>

#include <cstring>
#include <mutex>
#include <thread>
#include <vector>


#include "common.hpp"
#include "internal.hpp"
#include "g1.hip.h"
#include "sort.hip.h"
#include "msm_plan.hpp"

namespace gm {

static constexpr uint32_t G1_NEG_BIT = 0x80000000u;  // task index flag: subtract the point instead of adding it

// ------------------------------------------------------------------------------------------ engine kernels
// off[k] = first position of the sorted key array holding a key >= k, k = 0..nkeys (off[nkeys] = number of valid tasks)
__global__ void __launch_bounds__(256) k_g1_lower_bound(const uint32_t* __restrict__ keys, uint64_t ntasks, uint32_t nkeys,
                                                         uint32_t* __restrict__ off, uint32_t* __restrict__ max_len) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nkeys) return;
    auto lb = [&](uint32_t key) {
        uint64_t lo = 0, hi = ntasks;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        return (uint32_t)lo;
    };
    const uint32_t o = lb(k);
    off[k] = o;
    if (k < nkeys) {
        const uint32_t len = lb(k + 1) - o;
        if (len) atomicMax(max_len, len);
    }
}

// Row layouts of all the levels of the tree in two launches: level l (blockIdx.y = l - 1) has rows of ceil(len0 / 2^l) cells;
// its offsets are the exclusive scan of those lengths over the nkeys rows (+ the total at row nkeys).  Same tile scheme as
// exclusive_scan_u32 (sort.hip.h), the lengths computed on the fly from the level-0 offsets.
__global__ void __launch_bounds__(1024) k_g1_levels_scan_tiles(const uint32_t* __restrict__ off0, uint32_t nkeys, uint32_t* __restrict__ off_all,
                                                                size_t ostride, uint32_t* __restrict__ tile_tot) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, shift = blockIdx.y + 1;
    const uint32_t n = nkeys + 1;
    uint32_t* dst = off_all + (size_t)shift * ostride;
    const uint32_t r0 = blockIdx.x * SCAN_TILE + tid * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t sum = 0;
    uint32_t prev = (r0 < n) ? off0[r0] : 0u;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
        uint32_t len = 0;
        if (r0 + k < nkeys) { const uint32_t nx = off0[r0 + k + 1]; len = nx - prev; prev = nx; }
        v[k] = (len + (1u << shift) - 1) >> shift;
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
    if (tid == 0) tile_tot[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) k_g1_levels_scan_fix(uint32_t nkeys, uint32_t* __restrict__ off_all, size_t ostride,
                                                              const uint32_t* __restrict__ tile_tot) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    if (b == 0) return;
    const uint32_t* tt = tile_tot + (size_t)blockIdx.y * gridDim.x;
    uint32_t part = 0;
    for (uint32_t i = tid; i < b; i += 1024) part += tt[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
    if (lane == 0) wave_tot[wave] = part;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < 16; w++) off += wave_tot[w];
    uint32_t* dst = off_all + (size_t)(blockIdx.y + 1) * ostride;
    const uint32_t r0 = b * SCAN_TILE + tid * SCAN_ITEMS;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_ITEMS; k++)
        if (r0 + k <= nkeys) dst[r0 + k] += off;
}

__device__ __forceinline__ uint32_t g1_find_row(const uint32_t* __restrict__ off, uint32_t nrows, uint32_t j) {
    uint32_t lo = 0, hi = nrows;  // off[lo] <= j < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// First row of every 128-cell block of every level's output layout, all levels in one launch (as msm.hip's k_block_rows): a thread
// of k_g1_level then brackets its row with two loads instead of an 18-step search of dependent loads -- these kernels run one or
// two waves per SIMD, nothing hides that latency.
struct G1BlockRowsArgs {
    uint32_t nlev;
    uint32_t first[34];   // first entry of level l + 1 in blk_row; first[nlev] = total entries
};
__global__ void __launch_bounds__(256) k_g1_block_rows(G1BlockRowsArgs a, const uint32_t* __restrict__ off_all, size_t ostride,
                                                       uint32_t nrows, uint32_t* __restrict__ blk_row) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a.first[a.nlev]) return;
    uint32_t l = 0;
    while (l + 1 < a.nlev && e >= a.first[l + 1]) l++;
    const uint32_t* off = off_all + (size_t)(l + 1) * ostride;
    const uint32_t j = (e - a.first[l]) * 128, total = off[nrows];
    blk_row[e] = j < total ? g1_find_row(off, nrows, j) : nrows - 1;
}
__device__ __forceinline__ uint32_t g1_find_row_tab(const uint32_t* __restrict__ off, uint32_t nrows, const uint32_t* __restrict__ br, uint32_t j) {
    uint32_t lo = br[blockIdx.x], hi = br[blockIdx.x + 1] + 1;  // off[lo] <= j < off[hi]
    if (hi > nrows) hi = nrows;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// sources of level 0: points addressed through the sorted task indices
struct G1SrcAff {
    const G1Aff* pts;
    const uint32_t* idx;
    __device__ __forceinline__ G1Aff get(uint32_t cell) const {
        const uint32_t t = idx[cell];
        G1Aff p = g1_aff_load(pts + (t & ~G1_NEG_BIT));
        if (t & G1_NEG_BIT) p = g1_aff_neg(p);
        return p;
    }
};
struct G1SrcJac {
    const G1Jac* pts;
    const uint32_t* idx;  // nullptr: cells are stored in place (levels >= 1)
    __device__ __forceinline__ G1Jac get(uint32_t cell) const {
        if (!idx) return g1_load(pts + cell);
        const uint32_t t = idx[cell];
        G1Jac p = g1_load(pts + (t & ~G1_NEG_BIT));
        if (t & G1_NEG_BIT) p = g1_neg(p);
        return p;
    }
};

// intermediate cells of the tree: the 14 x 28 form as an addition leaves it (G1P14: Jacobian, 42 words = 168 bytes) -- neither the
// conversion to the wire form on the way out (3 x ~140 instructions) nor the one back on the way in (6 x ~100).
// -DGM_G1_CELLS_XYZZ=1: XYZZ cells (g1.hip.h, G1X14: 56 words = 224 bytes; a cell + cell addition is 12 products + 2 squares instead
// of the Jacobian 12 + 4 = 10 % fewer multiply-adds).  Measured in round 4, same box, alternating runs: 2^21-point MSM 11.41-11.47 ms
// against 11.27-11.30 with Jacobian cells, outer buckets 22.1 against 21.8 ms -- the 33 % wider cells (247 VGPRs instead of 209, 28 KB
// of LDS staging per workgroup instead of 21) take back what the two squares save.  Correct (the whole G1 / opening / proof suite
// passes with it), not faster: off.
#ifndef GM_G1_CELLS_XYZZ
#define GM_G1_CELLS_XYZZ 0
#endif
#if GM_G1_CELLS_XYZZ
typedef G1X14 G1Cell;
static constexpr uint32_t G1_CELL_WORDS = 56;
__device__ __forceinline__ G1Cell g1c_from_jac(const G1Jac& p) { return g1x_from_jac(p); }
__device__ __forceinline__ G1Jac g1c_to_jac(const G1Cell& p) { return g1x_to_jac(p); }
__device__ __forceinline__ G1Cell g1c_add(const G1Cell& p, const G1Cell& q) { return g1x_add(p, q); }
__device__ __forceinline__ G1Cell g1c_add_aff(const G1Aff& p, const G1Aff& q) { return g1x_add_aff(p, q); }
__device__ __forceinline__ const Fq14& g1c_coord(const G1Cell& p, int c) { return c == 0 ? p.x : c == 1 ? p.y : c == 2 ? p.zz : p.zzz; }
__device__ __forceinline__ Fq14& g1c_coord(G1Cell& p, int c) { return c == 0 ? p.x : c == 1 ? p.y : c == 2 ? p.zz : p.zzz; }
#else
typedef G1P14 G1Cell;
static constexpr uint32_t G1_CELL_WORDS = 42;
__device__ __forceinline__ G1Cell g1c_from_jac(const G1Jac& p) { return g1p14_from(p); }
__device__ __forceinline__ G1Jac g1c_to_jac(const G1Cell& p) { return g1p14_to(p); }
__device__ __forceinline__ G1Cell g1c_add(const G1Cell& p, const G1Cell& q) { return g1_add14p(p, q); }
__device__ __forceinline__ G1Cell g1c_add_aff(const G1Aff& p, const G1Aff& q) { return g1_add_aff14p(p, q); }
__device__ __forceinline__ const Fq14& g1c_coord(const G1Cell& p, int c) { return c == 0 ? p.x : c == 1 ? p.y : p.z; }
__device__ __forceinline__ Fq14& g1c_coord(G1Cell& p, int c) { return c == 0 ? p.x : c == 1 ? p.y : p.z; }
#endif
static constexpr int G1_CELL_COORDS = G1_CELL_WORDS / 14;
struct __attribute__((packed, aligned(8))) G1Quad {
    uint32_t a, b, c, d;
};
struct G1SrcCells {
    const uint32_t* cells;
    __device__ __forceinline__ G1Cell get(uint32_t cell) const {
        const uint32_t* p = cells + (size_t)cell * G1_CELL_WORDS;
        uint32_t w[G1_CELL_WORDS];
#pragma unroll
        for (int k = 0; k < (int)G1_CELL_WORDS / 4; k++) {
            const G1Quad q = *reinterpret_cast<const G1Quad*>(p + 4 * k);
            w[4 * k] = q.a; w[4 * k + 1] = q.b; w[4 * k + 2] = q.c; w[4 * k + 3] = q.d;
        }
#pragma unroll
        for (int k = 4 * ((int)G1_CELL_WORDS / 4); k < (int)G1_CELL_WORDS; k++) w[k] = p[k];
        G1Cell r;
#pragma unroll
        for (int c = 0; c < G1_CELL_COORDS; c++)
#pragma unroll
            for (int i = 0; i < 14; i++) g1c_coord(r, c).l[i] = w[14 * c + i];
        return r;
    }
};
// one output cell of a level, in the cell form
__device__ __forceinline__ G1Cell g1_pair14(const G1SrcAff& s, uint32_t c0, bool two) {
    const G1Aff a = s.get(c0);
    return two ? g1c_add_aff(a, s.get(c0 + 1)) : g1c_from_jac(g1_from_aff(a));
}
__device__ __forceinline__ G1Cell g1_pair14(const G1SrcJac& s, uint32_t c0, bool two) {
    const G1Cell a = g1c_from_jac(s.get(c0));
    return two ? g1c_add(a, g1c_from_jac(s.get(c0 + 1))) : a;
}
__device__ __forceinline__ G1Cell g1_pair14(const G1SrcCells& s, uint32_t c0, bool two) {
    const G1Cell a = s.get(c0);
    return two ? g1c_add(a, s.get(c0 + 1)) : a;
}
__device__ __forceinline__ G1Jac g1_pair(const G1SrcCells& s, uint32_t c0, bool) { return g1c_to_jac(s.get(c0)); }
__device__ __forceinline__ G1Jac g1_pair(const G1SrcAff& s, uint32_t c0, bool two) {
    const G1Aff a = s.get(c0);
    return two ? g1_add_aff(a, s.get(c0 + 1)) : g1_from_aff(a);
}
__device__ __forceinline__ G1Jac g1_pair(const G1SrcJac& s, uint32_t c0, bool two) {
    const G1Jac a = s.get(c0);
    return two ? g1_add(a, s.get(c0 + 1)) : a;
}

// one tree level: out cell j of row r = in cell 2p (+ in cell 2p+1 when the row has it), p = j - off_out[r]
static constexpr uint32_t G1_ROWS_LDS = 512;
// (two waves per SIMD asked for: the Jacobian-source instance otherwise takes 277 VGPRs + 21 AGPRs = ONE wave per SIMD; at 256 with
// 92 bytes of spills it runs the weighted sums of every MSM and the phase-2 pull commitments 14 % faster where they are large)
template <class Src>
__global__ void __launch_bounds__(128, 2) k_g1_level(Src src, const uint32_t* __restrict__ off_in,
                                                   const uint32_t* __restrict__ off_out, uint32_t nrows,
                                                   uint32_t* __restrict__ out, const uint32_t* __restrict__ br) {
    // the 128 results of a workgroup are 21 KB (28 KB with XYZZ cells) of contiguous output: staged in LDS and written as whole lines (a lane storing
    // its own cell leaves every 128-byte line partly written per store instruction: 1.7x the bytes in HBM writes measured)
    __shared__ uint2 stage[128 * (G1_CELL_WORDS / 2)];
    // the rows this workgroup's cells lie in (k_g1_block_rows) are few: their offsets go to LDS in one round trip and the row search
    // runs there instead of as a chain of up to seven dependent global loads per lane
    __shared__ uint32_t s_off[G1_ROWS_LDS];
    const uint32_t total = off_out[nrows];
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rlo = br[blockIdx.x], rhi = br[blockIdx.x + 1] + 1;   // off_out[rlo] <= j < off_out[rhi]
    if (rhi > nrows) rhi = nrows;
    const bool in_lds = rhi - rlo <= G1_ROWS_LDS;
    if (in_lds)
        for (uint32_t k = threadIdx.x; k < rhi - rlo; k += 128) s_off[k] = off_out[rlo + k];
    __syncthreads();
    if (j < total) {
        uint32_t r, o_r;
        if (in_lds) {
            uint32_t lo = 0, hi = rhi - rlo;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_off[mid] <= j) lo = mid; else hi = mid;
            }
            r = rlo + lo;
            o_r = s_off[lo];
        } else {
            r = g1_find_row_tab(off_out, nrows, br, j);
            o_r = off_out[r];
        }
        const uint32_t p = j - o_r;
        const uint32_t in0 = off_in[r], len = off_in[r + 1] - in0;
        const G1Cell v = g1_pair14(src, in0 + 2 * p, 2 * p + 1 < len);
        uint2* st = stage + threadIdx.x * (G1_CELL_WORDS / 2);
#pragma unroll
        for (int c = 0; c < G1_CELL_COORDS; c++)
#pragma unroll
            for (int k = 0; k < 7; k++) st[7 * c + k] = make_uint2(g1c_coord(v, c).l[2 * k], g1c_coord(v, c).l[2 * k + 1]);
    }
    __syncthreads();
    const uint32_t b0 = blockIdx.x * blockDim.x;
    if (b0 >= total) return;
    const uint32_t nv = (total - b0 < 128u ? total - b0 : 128u) * (G1_CELL_WORDS / 2);
    uint2* o2 = reinterpret_cast<uint2*>(out + (size_t)b0 * G1_CELL_WORDS);
    for (uint32_t k = threadIdx.x; k < nv; k += 128) o2[k] = stage[k];
}

// rows of at most one cell -> dense output (empty rows = infinity)
template <class Src>
__global__ void __launch_bounds__(128) k_g1_rows_out(Src src, const uint32_t* __restrict__ off, uint32_t nrows,
                                                      G1Jac* __restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t o = off[r];
    g1_store(out + r, (off[r + 1] > o) ? g1_pair(src, o, false) : g1_inf());
}

// ------------------------------------------------------------------------------------------ task builders
// MSM windows: task t = w * n + i -> key (w << c | digit) (digit 0: no task), point i
__global__ void __launch_bounds__(256) k_g1_msm_tasks(const uint32_t* __restrict__ scalars, uint64_t n, uint32_t c,
                                                       uint32_t nwin, int mont, uint32_t sentinel,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = fr_load(reinterpret_cast<const Fr*>(scalars) + i);
    if (mont) s = fr_from_mont(s);  // `into_bigint()` of msm_nonaffine.rs:21-23
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t bit = w * c;
        uint32_t d = 0;
        if (bit < 256) {
            const uint32_t li = bit >> 5, sh = bit & 31;
            uint64_t v = s.l[li];
            if (li + 1 < 8) v |= (uint64_t)s.l[li + 1] << 32;
            d = (uint32_t)(v >> sh) & ((1u << c) - 1);
        }
        const uint64_t t = (uint64_t)w * n + i;
        keys[t] = d ? ((w << c) | d) : sentinel;
        idx[t] = (uint32_t)i;
    }
}

// grouped MSM (several base arrays against one scalar array: the pull commitments of second_phase, pushforward.rs:596-605):
// flat base j of group g (pre[g] <= j < pre[g+1], i = j - pre[g]) in window w -> key ((g * nwin + w) << c | digit), point g * stride + i
__global__ void __launch_bounds__(256) k_g1_msm_tasks_grouped(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ pre,
                                                               uint32_t ngroups, uint64_t stride, uint64_t total, uint32_t c,
                                                               uint32_t nwin, int mont, uint32_t sentinel,
                                                               uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    uint32_t lo = 0, hi = ngroups;  // pre[lo] <= j < pre[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pre[mid] <= j) lo = mid; else hi = mid;
    }
    const uint32_t g = lo;
    const uint64_t i = j - pre[g];
    Fr s = fr_load(reinterpret_cast<const Fr*>(scalars) + i);
    if (mont) s = fr_from_mont(s);
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t bit = w * c;
        uint32_t d = 0;
        if (bit < 256) {
            const uint32_t li = bit >> 5, sh = bit & 31;
            uint64_t v = s.l[li];
            if (li + 1 < 8) v |= (uint64_t)s.l[li + 1] << 32;
            d = (uint32_t)(v >> sh) & ((1u << c) - 1);
        }
        const uint64_t t = (uint64_t)w * total + j;
        keys[t] = d ? (((g * nwin + w) << c) | d) : sentinel;
        idx[t] = (uint32_t)(g * stride + i);
    }
}

// ---- fixed-base MSM (a registered proving key): tab[w * n_reg + i] = 2^(c w) * base_i, affine.  With the window multiples
// precomputed, every window of every scalar lands in ONE set of 2^c buckets (key = digit), so a call costs n * nwin mixed
// additions into 2^16 long rows plus a single bucket reduction -- ~28 % fewer field multiplications than per-window buckets
// at 2^21 points, for 96 * nwin bytes of table per base (3.2 GB at 2^21: HBM is what this machine has plenty of).
__global__ void __launch_bounds__(128) k_g1_fixed_tables(const G1Aff* __restrict__ bases, uint64_t n, uint32_t c, uint32_t nwin,
                                                          G1Aff* __restrict__ tab) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Aff a = g1_aff_load(bases + i);
    g1_aff_store(tab + i, a);
    for (uint32_t w = 1; w < nwin; w++) {
        G1Jac p = g1_from_aff(a);
        for (uint32_t k = 0; k < c; k++) p = g1_dbl(p);
        a = g1_to_aff(p);
        g1_aff_store(tab + (uint64_t)w * n + i, a);
    }
}

__global__ void __launch_bounds__(256) k_g1_msm_tasks_fixed(const uint32_t* __restrict__ scalars, uint64_t n, uint64_t n_reg, uint32_t c,
                                                             uint32_t nwin, int mont, uint32_t sentinel,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = fr_load(reinterpret_cast<const Fr*>(scalars) + i);
    if (mont) s = fr_from_mont(s);
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t bit = w * c;
        uint32_t d = 0;
        if (bit < 256) {
            const uint32_t li = bit >> 5, sh = bit & 31;
            uint64_t v = s.l[li];
            if (li + 1 < 8) v |= (uint64_t)s.l[li + 1] << 32;
            d = (uint32_t)(v >> sh) & ((1u << c) - 1);
        }
        const uint64_t t = (uint64_t)w * n + i;
        keys[t] = d ? d : sentinel;
        idx[t] = (uint32_t)((uint64_t)w * n_reg + i);
    }
}

// weighted sum sum_i i * B[g][i] over groups g of `glen` buckets by bit decomposition: task (g, i, b) -> key g * nbits + b
__global__ void __launch_bounds__(256) k_g1_bit_tasks(uint32_t ngroups, uint32_t glen, uint32_t nbits, uint32_t sentinel,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = (uint64_t)ngroups * glen * nbits;
    if (t >= total) return;
    const uint32_t b = (uint32_t)(t % nbits);
    const uint64_t gi = t / nbits;
    const uint32_t i = (uint32_t)(gi % glen), g = (uint32_t)(gi / glen);
    keys[t] = ((i >> b) & 1u) ? g * nbits + b : sentinel;
    idx[t] = (uint32_t)gi;
}

// binary_msm: chunk t with coefficient byte c != 0 selects table entry t * (2^gamma - 1) + c - 1   (binary_msm.rs:22)
__global__ void __launch_bounds__(256) k_g1_binary_tasks(const uint8_t* __restrict__ coefs, uint64_t nchunks, uint32_t tab_len,
                                                          uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    const uint32_t c = coefs[t];
    keys[t] = c ? 0u : 1u;
    idx[t] = (uint32_t)(t * tab_len + (c ? c - 1 : 0));
}

// prepare_chunk (binary_msm.rs:32-42): entry i-1 of chunk k = sum over set bits idx of i of chunk[len-1-idx]
__global__ void __launch_bounds__(128) k_g1_prepare_tables(const G1Aff* __restrict__ bases, uint64_t n, uint32_t gamma,
                                                            G1Aff* __restrict__ tables) {
    const uint32_t tab_len = (1u << gamma) - 1;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nchunks = (n + gamma - 1) / gamma;
    if (t >= nchunks * tab_len) return;
    const uint64_t k = t / tab_len;
    const uint32_t i = (uint32_t)(t % tab_len) + 1;
    const uint64_t first = k * gamma;
    const uint32_t clen = (uint32_t)((n - first < gamma) ? n - first : gamma);
    G1Jac acc = g1_inf();
    for (uint32_t b = 0; b < clen; b++)
        if ((i >> b) & 1u) acc = g1_add_mixed(acc, g1_aff_load(bases + first + (clen - 1 - b)));
    g1_aff_store(tables + t, g1_to_aff(acc));  // G::normalize_batch (binary_msm.rs:41)
}

// pushforward outer buckets: task (y, x) adds basis[x + N * slot(y mod comm_mul)] to row (y / comm_mul - m0) * stride + v[y][x].
// slot: where slice s = y mod comm_mul of the KZG basis sits in the basis array the caller handed over -- the identity for a
// process that holds the whole key, a compact index for a rank of a window-sharded run that holds only the slices of its windows
// (gm_msm_g1_outer_part); m0 = the first commitment matrix the plan's windows touch.
struct G1Slots {
    uint16_t slot[256];   // clm <= 8
};
template <typename V>
__global__ void __launch_bounds__(256) k_g1_outer_tasks(const V* __restrict__ v, uint64_t N, uint32_t y0, uint32_t ny, uint32_t clm,
                                                         uint32_t m0, G1Slots sl, uint32_t stride, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ idx) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * ny) return;
    const uint32_t yl = (uint32_t)(t / N), y = y0 + yl;
    const uint64_t x = t % N;
    keys[t] = ((y >> clm) - m0) * stride + (uint32_t)v[t];
    idx[t] = (uint32_t)(x + N * sl.slot[y & ((1u << clm) - 1)]);
}

// plain keyed tasks: key = mapping[i] (Pullback::bucketed_msm, pullback.rs:44-46), point i
__global__ void __launch_bounds__(256) k_g1_map_tasks(const uint32_t* __restrict__ mapping, uint64_t n, uint32_t nkeys,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                       uint32_t* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = mapping[i];
    if (m >= nkeys) atomicAdd(bad, 1u);
    keys[i] = m < nkeys ? m : nkeys;
    idx[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(128) k_g1_to_affine(const G1Jac* __restrict__ in, uint64_t n, G1Aff* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_aff_store(out + i, g1_to_aff(g1_load(in + i)));
}

__global__ void __launch_bounds__(128) k_g1_from_affine(const G1Aff* __restrict__ in, uint64_t n, G1Jac* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_store(out + i, g1_from_aff(g1_aff_load(in + i)));
}

// k_i * G for the synthetic SRS of bench / tests: double-and-add over a 64-bit seeded scalar per point
__global__ void __launch_bounds__(128) k_g1_gen_points(G1Aff gen, uint64_t n, uint64_t seed, G1Aff* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (i + 1);  // SplitMix64 of the index
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    z |= 1;
    G1Jac acc = g1_inf();
    for (int b = 63; b >= 0; b--) {
        acc = g1_dbl(acc);
        if ((z >> b) & 1) acc = g1_add_mixed(acc, gen);
    }
    g1_aff_store(out + i, g1_to_aff(acc));
}

// mock KZG setup (KzgProvingKey::mock_setup, commitments/kzg.rs:84-98): ptau_1[i] = tau^i * g0, for tests and the bench's
// end-to-end check (a proof against this SRS satisfies the pairing equation, which the known tau lets one check in G1)
__global__ void __launch_bounds__(128) k_g1_mock_srs(G1Aff g0, Fr tau, uint64_t n, G1Aff* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr acc = fr_one(), b = tau;
    for (uint64_t e = i; e; e >>= 1) {
        if (e & 1) acc = fr_mul(acc, b);
        b = fr_sqr(b);
    }
    const Fr k = fr_from_mont(acc);
    G1Jac r = g1_inf();
    for (int li = 7; li >= 0; li--)
        for (int bit = 31; bit >= 0; bit--) {
            r = g1_dbl(r);
            if ((k.l[li] >> bit) & 1) r = g1_add_mixed(r, g0);
        }
    g1_aff_store(out + i, g1_to_aff(r));
}

// ------------------------------------------------------------------------------------------ engine (host)
// grow-only device scratch shared by the G1 calls of this process on one device (one call at a time per device: guarded by a mutex)
struct G1Scratch {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    std::mutex mu;
    int32_t reserve(size_t bytes) {
        if (bytes <= cap) return GM_OK;
        if (base) dev_free(base);
        base = nullptr;
        cap = 0;
        hipError_t e = dev_alloc((void**)&base, bytes);
        if (e != hipSuccess) return set_err(GM_ERR_HIP, "hipMalloc(G1 scratch %zu): %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return GM_OK;
    }
    void* carve(size_t b) {
        const size_t a = (used + 255) & ~(size_t)255;
        used = a + b;
        return base + a;
    }
    void release() {
        if (base) dev_free(base);
        base = nullptr;
        cap = used = 0;
    }
};
// one per DEVICE: rank threads of one process that drive different GPUs must neither share scratch memory (it lives on the device that
// was current when it grew) nor queue behind each other's G1 calls
static G1Scratch& g1_scratch() {
    static G1Scratch s[GM_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return s[(dev >= 0 && dev < GM_MAX_DEVICES) ? dev : 0];
}

static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

struct G1Layout {
    size_t sort_tmp = 0, scan_tmp = 0, total = 0;
};

static uint32_t bit_len(uint32_t v) {
    uint32_t b = 0;
    while (v) { b++; v >>= 1; }
    return b;
}

// bytes of scratch the engine needs for ntasks tasks over nkeys rows (besides the caller's task arrays)
static int32_t g1_engine_layout(uint64_t ntasks, uint32_t nkeys, G1Layout* L) {
    const size_t st = radix_sort_tmp_bytes(ntasks), sc = 33 * scan_tmp_bytes((size_t)nkeys + 1);   // tile totals of every level
    L->sort_tmp = st;
    L->scan_tmp = sc;
    L->total = al(st) + al(sc) + 2 * al(ntasks * 4) + 35 * al(((size_t)nkeys + 1) * 4) + al(64) + al((ntasks / 64 + (size_t)34 * (nkeys / 32 + 3) + 64) * 4) +
               al((ntasks / 2 + nkeys + 1) * (size_t)G1_CELL_WORDS * 4) + al((ntasks / 4 + nkeys + 1) * (size_t)G1_CELL_WORDS * 4) + 4096;
    return GM_OK;
}

// Sums the points of every key.  keys/idx: ntasks unsorted tasks (device, clobbered); tasks with key == nkeys are
// dropped.  src_aff / src_jac: exactly one non-null.  out: nkeys Jacobian points.  Synchronises the stream once.
static int32_t g1_sum_by_key(G1Scratch& ws, const G1Aff* src_aff, const G1Jac* src_jac, uint32_t* keys, uint32_t* idx,
                             uint64_t ntasks, uint32_t nkeys, G1Jac* out, hipStream_t s) {
    GM_REQUIRE(ntasks < (1ull << 31), "too many G1 tasks (%llu)", (unsigned long long)ntasks);
    GM_REQUIRE(nkeys >= 1 && nkeys < (1u << 30), "bad key count %u", nkeys);
    G1Layout L;
    int32_t rc = g1_engine_layout(ntasks ? ntasks : 1, nkeys, &L);
    if (rc) return rc;
    uint32_t* sort_tmp = (uint32_t*)ws.carve(L.sort_tmp);
    uint32_t* scan_tmp = (uint32_t*)ws.carve(L.scan_tmp);
    uint32_t* keys_s = (uint32_t*)ws.carve((ntasks + 1) * 4);
    uint32_t* idx_s = (uint32_t*)ws.carve((ntasks + 1) * 4);
    const size_t orow = (size_t)nkeys + 1;
    uint32_t* off_all = (uint32_t*)ws.carve(33 * al(orow * 4));  // level l at off_all + l * ostride
    const size_t ostride = al(orow * 4) / 4;
    (void)ws.carve(orow * 4);   // (reserved)
    uint32_t* d_max = (uint32_t*)ws.carve(64);
    uint32_t* bufA = (uint32_t*)ws.carve((ntasks / 2 + nkeys + 1) * (size_t)G1_CELL_WORDS * 4);
    uint32_t* bufB = (uint32_t*)ws.carve((ntasks / 4 + nkeys + 1) * (size_t)G1_CELL_WORDS * 4);
    uint32_t* blk_row = (uint32_t*)ws.carve((ntasks / 64 + (size_t)34 * (nkeys / 32 + 3) + 64) * 4);
    if (ws.used > ws.cap) return set_err(GM_ERR_STATE, "G1 scratch under-reserved (%zu > %zu)", ws.used, ws.cap);

    GM_HIP(hipMemsetAsync(d_max, 0, 4, s));
    // stable sort of the tasks by key (sort.hip.h): the sorted arrays end up in the caller's or in the scratch buffers
    GM_HIP(radix_sort_pairs_u32(keys, idx, keys_s, idx_s, ntasks, bit_len(nkeys), sort_tmp, &keys_s, &idx_s, s));
    hipLaunchKernelGGL(k_g1_lower_bound, dim3(ceil_div(orow, 256)), dim3(256), 0, s, keys_s, ntasks, nkeys, off_all, d_max);
    GM_LAUNCH_CHECK();
    uint32_t max_len = 0;
    GM_HIP(hipMemcpyAsync(&max_len, d_max, 4, hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    uint32_t nlev = 0;
    while ((1u << nlev) < max_len) nlev++;
    GM_REQUIRE(nlev <= 32, "row too long");
    if (nlev) {
        const uint32_t ntiles = ceil_div(orow, SCAN_TILE);
        hipLaunchKernelGGL(k_g1_levels_scan_tiles, dim3(ntiles, nlev), dim3(1024), 0, s, off_all, nkeys, off_all, ostride, scan_tmp);
        GM_LAUNCH_CHECK();
        if (ntiles > 1) {
            hipLaunchKernelGGL(k_g1_levels_scan_fix, dim3(ntiles, nlev), dim3(1024), 0, s, nkeys, off_all, ostride, scan_tmp);
            GM_LAUNCH_CHECK();
        }
    }
    const G1SrcAff sa{src_aff, idx_s};
    const G1SrcJac sj{src_jac, idx_s};
    if (nlev == 0) {
        if (src_aff) hipLaunchKernelGGL((k_g1_rows_out<G1SrcAff>), dim3(ceil_div(nkeys, 128)), dim3(128), 0, s, sa, off_all, nkeys, out);
        else hipLaunchKernelGGL((k_g1_rows_out<G1SrcJac>), dim3(ceil_div(nkeys, 128)), dim3(128), 0, s, sj, off_all, nkeys, out);
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    // level 1 from the sources; cells of level l never exceed ntasks / 2^l + nkeys
    uint64_t bound = ntasks / 2 + nkeys;
    G1BlockRowsArgs ba;
    ba.nlev = nlev;
    {
        uint64_t b = bound;
        uint32_t tot = 0;
        for (uint32_t l = 0; l < nlev; l++) {
            ba.first[l] = tot;
            tot += (uint32_t)((b + 127) / 128) + 1;
            b = b / 2 + nkeys;
        }
        ba.first[nlev] = tot;
        hipLaunchKernelGGL(k_g1_block_rows, dim3(ceil_div(tot, 256)), dim3(256), 0, s, ba, off_all, ostride, nkeys, blk_row);
        GM_LAUNCH_CHECK();
    }
    if (src_aff) hipLaunchKernelGGL((k_g1_level<G1SrcAff>), dim3(ceil_div(bound, 128)), dim3(128), 0, s, sa, off_all, off_all + ostride, nkeys, bufA, blk_row + ba.first[0]);
    else hipLaunchKernelGGL((k_g1_level<G1SrcJac>), dim3(ceil_div(bound, 128)), dim3(128), 0, s, sj, off_all, off_all + ostride, nkeys, bufA, blk_row + ba.first[0]);
    GM_LAUNCH_CHECK();
    uint32_t *cur = bufA, *nxt = bufB;
    for (uint32_t l = 2; l <= nlev; l++) {
        bound = bound / 2 + nkeys;
        const G1SrcCells sl{cur};
        hipLaunchKernelGGL((k_g1_level<G1SrcCells>), dim3(ceil_div(bound, 128)), dim3(128), 0, s, sl, off_all + (l - 1) * ostride,
                           off_all + l * ostride, nkeys, nxt, blk_row + ba.first[l - 1]);
        GM_LAUNCH_CHECK();
        uint32_t* t = cur; cur = nxt; nxt = t;
    }
    const G1SrcCells sl{cur};
    hipLaunchKernelGGL((k_g1_rows_out<G1SrcCells>), dim3(ceil_div(nkeys, 128)), dim3(128), 0, s, sl, off_all + nlev * ostride, nkeys, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// bufB must hold level-2 cells: bound_2 = (ntasks/2 + nkeys)/2 + nkeys <= ntasks/4 + 1.5 nkeys: widen the reservation
static size_t g1_engine_bytes(uint64_t ntasks, uint32_t nkeys) {
    G1Layout L;
    if (g1_engine_layout(ntasks ? ntasks : 1, nkeys, &L)) return 0;
    return L.total + al((size_t)nkeys * sizeof(G1Jac));
}

// acc = sum_pos 2^pos S[pos] on the host (Horner from the top bit)
static G1Jac g1_horner_bits(const std::vector<G1Jac>& S) {
    G1Jac acc = g1_inf();
    for (size_t pos = S.size(); pos-- > 0;) {
        acc = g1_dbl(acc);
        acc = g1_add(acc, S[pos]);
    }
    return acc;
}

// sum_i i * B[g][i] for ngroups groups of glen buckets (device, Jacobian) -> host points
static int32_t g1_weighted_sums(G1Scratch& ws, const G1Jac* d_buckets, uint32_t ngroups, uint32_t glen, std::vector<G1Jac>* out,
                                hipStream_t s) {
    const uint32_t nbits = bit_len(glen - 1) ? bit_len(glen - 1) : 1;
    const uint64_t ntasks = (uint64_t)ngroups * glen * nbits;
    const uint32_t nkeys = ngroups * nbits;
    uint32_t* keys = (uint32_t*)ws.carve(ntasks * 4);
    uint32_t* idx = (uint32_t*)ws.carve(ntasks * 4);
    G1Jac* d_S = (G1Jac*)ws.carve((size_t)nkeys * sizeof(G1Jac));
    if (ws.used > ws.cap) return set_err(GM_ERR_STATE, "G1 scratch under-reserved");
    hipLaunchKernelGGL(k_g1_bit_tasks, dim3(ceil_div(ntasks, 256)), dim3(256), 0, s, ngroups, glen, nbits, nkeys, keys, idx);
    GM_LAUNCH_CHECK();
    int32_t rc = g1_sum_by_key(ws, nullptr, d_buckets, keys, idx, ntasks, nkeys, d_S, s);
    if (rc) return rc;
    std::vector<G1Jac> S(nkeys);
    GM_HIP(hipMemcpyAsync(S.data(), d_S, (size_t)nkeys * sizeof(G1Jac), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    out->resize(ngroups);
    host_parallel_for(ngroups, [&](uint32_t g) {
        (*out)[g] = g1_horner_bits(std::vector<G1Jac>(S.begin() + (size_t)g * nbits, S.begin() + (size_t)(g + 1) * nbits));
    });
    return GM_OK;
}
static size_t g1_weighted_bytes(uint32_t ngroups, uint32_t glen) {
    const uint32_t nbits = bit_len(glen - 1) ? bit_len(glen - 1) : 1;
    const uint64_t ntasks = (uint64_t)ngroups * glen * nbits;
    return 2 * al(ntasks * 4) + al((size_t)ngroups * nbits * sizeof(G1Jac)) + g1_engine_bytes(ntasks, ngroups * nbits) + 4096;
}

// Window width of the per-window MSM: the cost is windows x (a task per point + the buckets, whose reduction is Jacobian work worth
// about 15 tasks each); measured at 2^19 .. 2^22 points x 255 bits (c = 11 .. 16) the minimum of this model is within 2 % of the
// best width everywhere, where "log2 n - 7, at most 14" lost 4-6 % at 2^21 and 2^22 (17 windows of 15 bits cover 255 bits exactly).
static uint32_t g1_msm_window(uint64_t n, uint32_t nbits) {
    static const int forced = [] { const char* e = getenv("GM_G1_WINDOW"); return e ? atoi(e) : 0; }();   // development switch
    if (forced >= 2 && forced <= 16) return (uint32_t)forced;
    uint32_t best = 2;
    double best_cost = 0;
    for (uint32_t c = 2; c <= 16; c++) {
        const double cost = (double)((nbits + c - 1) / c) * ((double)n + 15.0 * (double)(1u << c));
        if (c == 2 || cost < best_cost) { best = c; best_cost = cost; }
    }
    return best;
}

// sum_i scalar_i * base_i; exactly one of aff / jac.  Result on the host (Jacobian).
static int32_t g1_msm_core(const G1Aff* aff, const G1Jac* jac, const uint64_t* d_scalars, uint64_t n, int scalars_mont,
                           uint32_t nbits, G1Jac* h_out, hipStream_t s) {
    if (n == 0) { *h_out = g1_inf(); return GM_OK; }
    GM_REQUIRE(nbits >= 1 && nbits <= 256, "bad scalar width %u", nbits);
    const uint32_t c = g1_msm_window(n, nbits);
    const uint32_t nwin = (nbits + c - 1) / c;
    const uint64_t ntasks = (uint64_t)nwin * n;
    const uint32_t nkeys = nwin << c;
    GM_REQUIRE(ntasks < (1ull << 31), "MSM too large for one call: %llu tasks", (unsigned long long)ntasks);
    G1Scratch& ws = g1_scratch();
    std::lock_guard<std::mutex> lock(ws.mu);
    const size_t need = 2 * al(ntasks * 4) + al((size_t)nkeys * sizeof(G1Jac)) + g1_engine_bytes(ntasks, nkeys) +
                        g1_weighted_bytes(nwin, 1u << c) + 8192;
    int32_t rc = ws.reserve(need);
    if (rc) return rc;
    ws.used = 0;
    uint32_t* keys = (uint32_t*)ws.carve(ntasks * 4);
    uint32_t* idx = (uint32_t*)ws.carve(ntasks * 4);
    G1Jac* buckets = (G1Jac*)ws.carve((size_t)nkeys * sizeof(G1Jac));
    hipLaunchKernelGGL(k_g1_msm_tasks, dim3(ceil_div(n, 256)), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(d_scalars), n, c,
                       nwin, scalars_mont, nkeys, keys, idx);
    GM_LAUNCH_CHECK();
    const size_t mark = ws.used;
    rc = g1_sum_by_key(ws, aff, jac, keys, idx, ntasks, nkeys, buckets, s);
    if (rc) return rc;
    ws.used = mark;  // the engine's scratch is free again; the bucket array stays
    // window w contributes 2^(c w) * sum_d d * B[w][d]: all windows at once as one weighted sum over bit positions
    std::vector<G1Jac> wsum;
    rc = g1_weighted_sums(ws, buckets, nwin, 1u << c, &wsum, s);
    if (rc) return rc;
    G1Jac acc = g1_inf();
    for (uint32_t w = nwin; w-- > 0;) {
        for (uint32_t k = 0; k < c; k++) acc = g1_dbl(acc);
        acc = g1_add(acc, wsum[w]);
    }
    *h_out = acc;
    return GM_OK;
}

// ngroups MSMs sharing one scalar array: group g = bases[g * stride .. g * stride + n_g[g]) against scalars[0 .. n_g[g]).
// One engine call for all groups and windows (the 64 separate calls of the full prover were launch-bound).
static int32_t g1_msm_grouped_core(const G1Jac* jac, uint64_t stride, const uint32_t* h_n, uint32_t ngroups, const uint64_t* d_scalars,
                                   int scalars_mont, uint32_t nbits, G1Jac* h_out, hipStream_t s) {
    GM_REQUIRE(ngroups >= 1 && ngroups <= 65536 && nbits >= 1 && nbits <= 256, "bad argument");
    std::vector<uint32_t> pre(ngroups + 1, 0);
    uint32_t nmax = 0;
    for (uint32_t g = 0; g < ngroups; g++) {
        GM_REQUIRE(h_n[g] <= stride, "group longer than the stride");
        pre[g + 1] = pre[g] + h_n[g];
        nmax = h_n[g] > nmax ? h_n[g] : nmax;
    }
    const uint64_t total = pre[ngroups];
    for (uint32_t g = 0; g < ngroups; g++) h_out[g] = g1_inf();
    if (total == 0) return GM_OK;
    // window from the typical group size (buckets are per (group, window)): c = log2(median-ish n) - 4, within [2, 12]
    uint32_t lg = 0;
    while ((1ull << lg) < (total / ngroups + 1)) lg++;
    int ci = (int)lg - 4;
    if (ci < 2) ci = 2;
    if (ci > 12) ci = 12;
    const uint32_t c = (uint32_t)ci, nwin = (nbits + c - 1) / c;
    const uint64_t ntasks = (uint64_t)nwin * total;
    const uint64_t nkeys64 = ((uint64_t)ngroups * nwin) << c;
    GM_REQUIRE(ntasks < (1ull << 31) && nkeys64 < (1ull << 30) && stride * ngroups < (1ull << 31), "grouped MSM too large for one call");
    const uint32_t nkeys = (uint32_t)nkeys64;
    G1Scratch& ws = g1_scratch();
    std::lock_guard<std::mutex> lock(ws.mu);
    const size_t need = 2 * al(ntasks * 4) + al((size_t)nkeys * sizeof(G1Jac)) + al((ngroups + 1) * 4) + g1_engine_bytes(ntasks, nkeys) +
                        g1_weighted_bytes(ngroups * nwin, 1u << c) + 8192;
    int32_t rc = ws.reserve(need);
    if (rc) return rc;
    ws.used = 0;
    uint32_t* keys = (uint32_t*)ws.carve(ntasks * 4);
    uint32_t* idx = (uint32_t*)ws.carve(ntasks * 4);
    G1Jac* buckets = (G1Jac*)ws.carve((size_t)nkeys * sizeof(G1Jac));
    uint32_t* d_pre = (uint32_t*)ws.carve((ngroups + 1) * 4);
    GM_HIP(hipMemcpyAsync(d_pre, pre.data(), (ngroups + 1) * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_g1_msm_tasks_grouped, dim3(ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(d_scalars),
                       d_pre, ngroups, stride, total, c, nwin, scalars_mont, nkeys, keys, idx);
    GM_LAUNCH_CHECK();
    const size_t mark = ws.used;
    rc = g1_sum_by_key(ws, nullptr, jac, keys, idx, ntasks, nkeys, buckets, s);  // synchronises: `pre` may go
    if (rc) return rc;
    ws.used = mark;
    std::vector<G1Jac> wsum;
    rc = g1_weighted_sums(ws, buckets, ngroups * nwin, 1u << c, &wsum, s);
    if (rc) return rc;
    host_parallel_for(ngroups, [&](uint32_t g) {
        G1Jac acc = g1_inf();
        for (uint32_t w = nwin; w-- > 0;) {
            for (uint32_t k = 0; k < c; k++) acc = g1_dbl(acc);
            acc = g1_add(acc, wsum[(size_t)g * nwin + w]);
        }
        h_out[g] = acc;
    }, 2);
    return GM_OK;
}

// ---- registry of fixed bases: gm_g1_msm looks its base pointer up here
struct G1FixedBase {
    const G1Aff* bases = nullptr;
    uint64_t n = 0;
    uint32_t c = 16, nwin = 16;
    G1Aff* tab = nullptr;
    int dev = 0;                 // device the bases live on
    G1Aff first{}, last{};       // fingerprint: bases[0] and bases[n - 1] at registration
};
static std::mutex& g1_fixed_mu() {
    static std::mutex m;
    return m;
}
static std::vector<G1FixedBase>& g1_fixed_registry() {
    static std::vector<G1FixedBase> r;
    return r;
}
// A registered key is recognised by its pointer, its device AND its content fingerprint: the caching allocator makes address
// reuse likely, so a buffer that was freed or overwritten without gm_g1_fixed_base_release must not silently get the old
// tables (two 96-byte reads, ~20 us, against an MSM of milliseconds).
static bool g1_fixed_lookup(const G1Aff* bases, uint64_t n, G1FixedBase* out, hipStream_t s) {
    G1FixedBase cand;
    {
        std::lock_guard<std::mutex> lock(g1_fixed_mu());
        bool found = false;
        int dev = 0;
        (void)hipGetDevice(&dev);
        for (const G1FixedBase& f : g1_fixed_registry())
            if (f.bases == bases && n <= f.n && f.dev == dev) { cand = f; found = true; break; }
        if (!found) return false;
    }
    G1Aff now[2];
    if (hipMemcpyAsync(&now[0], cand.bases, sizeof(G1Aff), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&now[1], cand.bases + (cand.n - 1), sizeof(G1Aff), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (memcmp(&now[0], &cand.first, sizeof(G1Aff)) != 0 || memcmp(&now[1], &cand.last, sizeof(G1Aff)) != 0) return false;
    *out = cand;
    return true;
}

static int32_t g1_msm_fixed_core(const G1FixedBase& fb, const uint64_t* d_scalars, uint64_t n, int scalars_mont, uint32_t nbits,
                                 G1Jac* h_out, hipStream_t s) {
    if (n == 0) { *h_out = g1_inf(); return GM_OK; }
    GM_REQUIRE(nbits >= 1 && nbits <= 256, "bad scalar width %u", nbits);
    uint32_t nwin = (nbits + fb.c - 1) / fb.c;
    if (nwin > fb.nwin) nwin = fb.nwin;
    const uint64_t ntasks = (uint64_t)nwin * n;
    const uint32_t nkeys = 1u << fb.c;
    GM_REQUIRE(ntasks < (1ull << 31) && fb.n * fb.nwin < (1ull << 31), "fixed-base MSM too large for one call");
    G1Scratch& ws = g1_scratch();
    std::lock_guard<std::mutex> lock(ws.mu);
    const size_t need = 2 * al(ntasks * 4) + al((size_t)nkeys * sizeof(G1Jac)) + g1_engine_bytes(ntasks, nkeys) +
                        g1_weighted_bytes(1, nkeys) + 8192;
    int32_t rc = ws.reserve(need);
    if (rc) return rc;
    ws.used = 0;
    uint32_t* keys = (uint32_t*)ws.carve(ntasks * 4);
    uint32_t* idx = (uint32_t*)ws.carve(ntasks * 4);
    G1Jac* buckets = (G1Jac*)ws.carve((size_t)nkeys * sizeof(G1Jac));
    hipLaunchKernelGGL(k_g1_msm_tasks_fixed, dim3(ceil_div(n, 256)), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(d_scalars), n,
                       fb.n, fb.c, nwin, scalars_mont, nkeys, keys, idx);
    GM_LAUNCH_CHECK();
    const size_t mark = ws.used;
    rc = g1_sum_by_key(ws, fb.tab, nullptr, keys, idx, ntasks, nkeys, buckets, s);
    if (rc) return rc;
    ws.used = mark;
    std::vector<G1Jac> wsum;
    rc = g1_weighted_sums(ws, buckets, 1, nkeys, &wsum, s);   // sum_d d * B[d]: the whole MSM
    if (rc) return rc;
    *h_out = wsum[0];
    return GM_OK;
}

}  // namespace gm

using namespace gm;

// ============================================================================================ C ABI
static void put_aff(uint64_t* h, const G1Jac& p) {
    const G1Aff a = g1_to_aff(p);
    memcpy(h, &a, sizeof(G1Aff));
}

extern "C" int32_t gm_g1_release_scratch(void) {
    G1Scratch& ws = g1_scratch();
    std::lock_guard<std::mutex> lock(ws.mu);
    ws.release();
    return GM_OK;
}

// host-side point arithmetic (same source as the device code): op 0 add (jac, jac) -> jac, 1 double, 2 mixed add
// (jac, aff) -> jac, 3 into_affine (jac -> aff), 4 affine + affine -> jac, 5 on-curve check of an affine point
// (out: one u64 per point), 6 Fq mul (a, b: 6 x u64), 7 Fq mul through the 32-bit-limb device formulation
extern "C" int32_t gm_g1_host(int32_t op, const uint64_t* h_a, const uint64_t* h_b, uint64_t* h_out, uint64_t n) {
    GM_REQUIRE(op >= 0 && op <= 7 && h_a && h_out, "bad argument");
    GM_REQUIRE(!(op == 0 || op == 2 || op == 4 || op == 6 || op == 7) || h_b, "second operand missing");
    for (uint64_t i = 0; i < n; i++) {
        G1Jac ja, jb, jr;
        G1Aff aa, ab, ar;
        switch (op) {
            case 0: memcpy(&ja, h_a + 18 * i, 144); memcpy(&jb, h_b + 18 * i, 144); jr = g1_add(ja, jb); memcpy(h_out + 18 * i, &jr, 144); break;
            case 1: memcpy(&ja, h_a + 18 * i, 144); jr = g1_dbl(ja); memcpy(h_out + 18 * i, &jr, 144); break;
            case 2: memcpy(&ja, h_a + 18 * i, 144); memcpy(&ab, h_b + 12 * i, 96); jr = g1_add_mixed(ja, ab); memcpy(h_out + 18 * i, &jr, 144); break;
            case 3: memcpy(&ja, h_a + 18 * i, 144); ar = g1_to_aff(ja); memcpy(h_out + 12 * i, &ar, 96); break;
            case 4: memcpy(&aa, h_a + 12 * i, 96); memcpy(&ab, h_b + 12 * i, 96); jr = g1_add_aff(aa, ab); memcpy(h_out + 18 * i, &jr, 144); break;
            case 5: memcpy(&aa, h_a + 12 * i, 96); h_out[i] = g1_aff_on_curve(aa) ? 1 : 0; break;
            default: {
                Fq x, y;
                memcpy(&x, h_a + 6 * i, 48); memcpy(&y, h_b + 6 * i, 48);
                const Fq z = op == 6 ? fq_mul(x, y) : fq_mul_c(x, y);
                memcpy(h_out + 6 * i, &z, 48);
            }
        }
    }
    return GM_OK;
}

// elementwise device point ops (tests of the device formulas): same op codes 0..4 as gm_g1_host
namespace gm {
__global__ void __launch_bounds__(128) k_g1_batch(int op, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                                   uint64_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Jac* ja = reinterpret_cast<const G1Jac*>(a);
    const G1Jac* jb = reinterpret_cast<const G1Jac*>(b);
    const G1Aff* aa = reinterpret_cast<const G1Aff*>(a);
    const G1Aff* ab = reinterpret_cast<const G1Aff*>(b);
    G1Jac* jo = reinterpret_cast<G1Jac*>(out);
    switch (op) {
        case 0: g1_store(jo + i, g1_add(g1_load(ja + i), g1_load(jb + i))); break;
        case 1: g1_store(jo + i, g1_dbl(g1_load(ja + i))); break;
        case 2: g1_store(jo + i, g1_add_mixed(g1_load(ja + i), g1_aff_load(ab + i))); break;
        case 3: g1_aff_store(reinterpret_cast<G1Aff*>(out) + i, g1_to_aff(g1_load(ja + i))); break;
        default: g1_store(jo + i, g1_add_aff(g1_aff_load(aa + i), g1_aff_load(ab + i))); break;
    }
}
}  // namespace gm

extern "C" int32_t gm_g1_batch(int32_t op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n, void* stream) {
    GM_REQUIRE(op >= 0 && op <= 4 && d_a && d_out, "bad argument");
    GM_REQUIRE(!(op == 0 || op == 2 || op == 4) || d_b, "second operand missing");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(k_g1_batch, dim3(ceil_div(n, 128)), dim3(128), 0, as_stream(stream), op, d_a, d_b, d_out, n);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int32_t gm_g1_gen_points(uint64_t* d_points_aff, uint64_t n, uint64_t seed, void* stream) {
    GM_REQUIRE(d_points_aff, "null argument");
    if (n == 0) return GM_OK;
    // the standard generator (ark-bls12-381 G1_GENERATOR_X / _Y), canonical -> Montgomery on the host
    static const uint32_t GX[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                    0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    static const uint32_t GY[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                    0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    G1Aff g;
    memcpy(&g.x, GX, 48);
    memcpy(&g.y, GY, 48);
    g.x = fq_to_mont(g.x);
    g.y = fq_to_mont(g.y);
    GM_REQUIRE(g1_aff_on_curve(g), "generator constant is off the curve");
    hipLaunchKernelGGL(k_g1_gen_points, dim3(ceil_div(n, 128)), dim3(128), 0, as_stream(stream), g, n, seed,
                       reinterpret_cast<G1Aff*>(d_points_aff));
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// the standard G1 generator in the affine wire form
extern "C" int32_t gm_g1_generator(uint64_t* h_out_aff) {
    GM_REQUIRE(h_out_aff, "null argument");
    static const uint32_t X[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                   0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    static const uint32_t Y[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                   0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    G1Aff g;
    for (int i = 0; i < 12; i++) { g.x.l[i] = X[i]; g.y.l[i] = Y[i]; }
    g.x = fq_to_mont(g.x);
    g.y = fq_to_mont(g.y);
    GM_REQUIRE(g1_aff_on_curve(g), "generator constant is wrong");
    memcpy(h_out_aff, &g, sizeof(G1Aff));
    return GM_OK;
}

extern "C" int32_t gm_g1_mock_srs(const uint64_t* h_tau, const uint64_t* h_g0_aff, uint64_t n, uint64_t* d_out_aff, void* stream) {
    GM_REQUIRE(h_tau && h_g0_aff && d_out_aff, "null argument");
    if (n == 0) return GM_OK;
    Fr tau;
    G1Aff g0;
    memcpy(&tau, h_tau, 32);
    memcpy(&g0, h_g0_aff, sizeof(G1Aff));
    GM_REQUIRE(g1_aff_on_curve(g0) && !g1_aff_is_inf(g0), "g0 is not a finite point of the curve");
    hipLaunchKernelGGL(k_g1_mock_srs, dim3(ceil_div(n, 128)), dim3(128), 0, as_stream(stream), g0, tau, n, reinterpret_cast<G1Aff*>(d_out_aff));
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int32_t gm_g1_msm(const uint64_t* d_bases_aff, const uint64_t* d_scalars, uint64_t n, int32_t scalars_mont,
                             uint32_t nbits, uint64_t* h_out_aff, void* stream) {
    GM_REQUIRE((d_bases_aff && d_scalars) || n == 0, "null argument");
    GM_REQUIRE(h_out_aff, "null output");
    G1Jac r;
    G1FixedBase fb;
    int32_t rc;
    if (n && g1_fixed_lookup(reinterpret_cast<const G1Aff*>(d_bases_aff), n, &fb, as_stream(stream)))
        rc = g1_msm_fixed_core(fb, d_scalars, n, scalars_mont, nbits, &r, as_stream(stream));
    else
        rc = g1_msm_core(reinterpret_cast<const G1Aff*>(d_bases_aff), nullptr, d_scalars, n, scalars_mont, nbits, &r, as_stream(stream));
    if (rc) return rc;
    put_aff(h_out_aff, r);
    return GM_OK;
}

// Precompute the window multiples of a base array that many MSMs will use (a KZG proving key): afterwards every gm_g1_msm whose
// d_bases_aff is this pointer (any n up to the registered one) takes the fixed-base path.  The bases must stay unchanged and
// allocated until gm_g1_fixed_base_release.  Table: 16 windows of 16 bits, 96 * 16 bytes per base.
extern "C" int32_t gm_g1_fixed_base_register(const uint64_t* d_bases_aff, uint64_t n, void* stream) {
    GM_REQUIRE(d_bases_aff && n >= 1, "bad argument");
    G1FixedBase fb;
    fb.bases = reinterpret_cast<const G1Aff*>(d_bases_aff);
    fb.n = n;
    GM_REQUIRE(fb.n * fb.nwin < (1ull << 31), "base array too long for a fixed-base table");
    {
        std::lock_guard<std::mutex> lock(g1_fixed_mu());
        for (const G1FixedBase& f : g1_fixed_registry())
            if (f.bases == fb.bases) return set_err(GM_ERR_STATE, "these bases are already registered");
    }
    hipError_t e = dev_alloc((void**)&fb.tab, (size_t)fb.n * fb.nwin * sizeof(G1Aff));
    if (e != hipSuccess) return set_err(GM_ERR_HIP, "fixed-base table (%.1f GiB): %s", fb.n * fb.nwin * 96.0 / (1 << 30), hipGetErrorString(e));
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_g1_fixed_tables, dim3(ceil_div(n, 128)), dim3(128), 0, s, fb.bases, n, fb.c, fb.nwin, fb.tab);
    (void)hipGetDevice(&fb.dev);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(&fb.first, fb.bases, sizeof(G1Aff), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&fb.last, fb.bases + (n - 1), sizeof(G1Aff), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        dev_free(fb.tab);
        return set_err(GM_ERR_HIP, "fixed-base table kernel failed");
    }
    std::lock_guard<std::mutex> lock(g1_fixed_mu());
    g1_fixed_registry().push_back(fb);
    return GM_OK;
}
extern "C" int32_t gm_g1_fixed_base_release(const uint64_t* d_bases_aff) {
    std::lock_guard<std::mutex> lock(g1_fixed_mu());
    auto& r = g1_fixed_registry();
    for (size_t i = 0; i < r.size(); i++)
        if (r[i].bases == reinterpret_cast<const G1Aff*>(d_bases_aff)) {
            dev_free(r[i].tab);
            r.erase(r.begin() + i);
            return GM_OK;
        }
    return set_err(GM_ERR_INVALID, "these bases are not registered");
}

extern "C" int32_t gm_g1_msm_nonaff(const uint64_t* d_bases_jac, const uint64_t* d_scalars, uint64_t n, int32_t scalars_mont,
                                    uint32_t nbits, uint64_t* h_out_aff, void* stream) {
    GM_REQUIRE((d_bases_jac && d_scalars) || n == 0, "null argument");
    GM_REQUIRE(h_out_aff, "null output");
    G1Jac r;
    int32_t rc = g1_msm_core(nullptr, reinterpret_cast<const G1Jac*>(d_bases_jac), d_scalars, n, scalars_mont, nbits, &r,
                             as_stream(stream));
    if (rc) return rc;
    put_aff(h_out_aff, r);
    return GM_OK;
}

extern "C" int32_t gm_g1_msm_nonaff_grouped(const uint64_t* d_bases_jac, uint64_t stride, const uint32_t* h_n, uint32_t n_groups,
                                            const uint64_t* d_scalars, int32_t scalars_mont, uint32_t nbits, uint64_t* h_out_aff,
                                            void* stream) {
    GM_REQUIRE(d_bases_jac && h_n && d_scalars && h_out_aff && n_groups >= 1, "null argument");
    std::vector<G1Jac> r(n_groups);
    int32_t rc = g1_msm_grouped_core(reinterpret_cast<const G1Jac*>(d_bases_jac), stride, h_n, n_groups, d_scalars, scalars_mont, nbits,
                                     r.data(), as_stream(stream));
    if (rc) return rc;
    for (uint32_t g = 0; g < n_groups; g++) put_aff(h_out_aff + 12 * (size_t)g, r[g]);
    return GM_OK;
}

extern "C" int32_t gm_g1_bucket_sums(const uint64_t* d_bases_aff, const uint32_t* d_mapping, uint64_t n, uint32_t n_buckets,
                                     uint64_t* d_out_jac, void* stream) {
    GM_REQUIRE(d_bases_aff && d_mapping && d_out_jac && n_buckets >= 1, "bad argument");
    hipStream_t s = as_stream(stream);
    G1Scratch& ws = g1_scratch();
    std::lock_guard<std::mutex> lock(ws.mu);
    int32_t rc = ws.reserve(2 * al(n * 4 + 4) + al(64) + g1_engine_bytes(n, n_buckets) + 8192);
    if (rc) return rc;
    ws.used = 0;
    uint32_t* keys = (uint32_t*)ws.carve(n * 4 + 4);
    uint32_t* idx = (uint32_t*)ws.carve(n * 4 + 4);
    uint32_t* bad = (uint32_t*)ws.carve(64);
    GM_HIP(hipMemsetAsync(bad, 0, 4, s));
    if (n) {
        hipLaunchKernelGGL(k_g1_map_tasks, dim3(ceil_div(n, 256)), dim3(256), 0, s, d_mapping, n, n_buckets, keys, idx, bad);
        GM_LAUNCH_CHECK();
    }
    rc = g1_sum_by_key(ws, reinterpret_cast<const G1Aff*>(d_bases_aff), nullptr, keys, idx, n, n_buckets,
                       reinterpret_cast<G1Jac*>(d_out_jac), s);
    if (rc) return rc;
    uint32_t nbad = 0;
    GM_HIP(hipMemcpyAsync(&nbad, bad, 4, hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    GM_REQUIRE(nbad == 0, "%u mapping entries out of range (index out of bounds panic in pullback.rs:45)", nbad);
    return GM_OK;
}

extern "C" int32_t gm_g1_pullback_msm(const uint64_t* d_bases_aff, const uint32_t* d_mapping, uint64_t n,
                                      const uint64_t* d_image, uint32_t image_len, uint64_t* h_out_aff, void* stream) {
    GM_REQUIRE(d_bases_aff && d_mapping && d_image && h_out_aff && image_len >= 1, "bad argument");
    G1Jac* d_b = nullptr;
    GM_HIP(dev_alloc((void**)&d_b, (size_t)image_len * sizeof(G1Jac)));
    int32_t rc = gm_g1_bucket_sums(d_bases_aff, d_mapping, n, image_len, reinterpret_cast<uint64_t*>(d_b), stream);
    if (!rc) rc = gm_g1_msm_nonaff(reinterpret_cast<const uint64_t*>(d_b), d_image, image_len, 1, 255, h_out_aff, stream);
    dev_free(d_b);
    return rc;
}

extern "C" int32_t gm_g1_weighted_sum(const uint64_t* d_buckets_jac, uint32_t n_groups, uint32_t group_len, uint64_t* h_out_aff,
                                      void* stream) {
    GM_REQUIRE(d_buckets_jac && h_out_aff && n_groups >= 1 && group_len >= 1, "bad argument");
    G1Scratch& ws = g1_scratch();
    std::lock_guard<std::mutex> lock(ws.mu);
    int32_t rc = ws.reserve(g1_weighted_bytes(n_groups, group_len));
    if (rc) return rc;
    ws.used = 0;
    std::vector<G1Jac> r;
    rc = g1_weighted_sums(ws, reinterpret_cast<const G1Jac*>(d_buckets_jac), n_groups, group_len, &r, as_stream(stream));
    if (rc) return rc;
    for (uint32_t g = 0; g < n_groups; g++) put_aff(h_out_aff + 12 * (size_t)g, r[g]);
    return GM_OK;
}

extern "C" int32_t gm_g1_prepare_bases(const uint64_t* d_bases_aff, uint64_t n, uint32_t gamma, uint64_t* d_tables_aff,
                                       void* stream) {
    GM_REQUIRE(d_bases_aff && d_tables_aff && gamma >= 1 && gamma <= 8, "bad argument (1 <= gamma <= 8: coefficients are u8)");
    if (n == 0) return GM_OK;
    const uint64_t total = ((n + gamma - 1) / gamma) * ((1u << gamma) - 1);
    hipLaunchKernelGGL(k_g1_prepare_tables, dim3(ceil_div(total, 128)), dim3(128), 0, as_stream(stream),
                       reinterpret_cast<const G1Aff*>(d_bases_aff), n, gamma, reinterpret_cast<G1Aff*>(d_tables_aff));
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int32_t gm_g1_binary_msm(const uint8_t* d_coefs, const uint64_t* d_tables_aff, uint64_t n_chunks, uint32_t gamma,
                                    uint64_t* h_out_aff, void* stream) {
    GM_REQUIRE(h_out_aff && gamma >= 1 && gamma <= 8, "bad argument");
    GM_REQUIRE((d_coefs && d_tables_aff) || n_chunks == 0, "null argument");
    hipStream_t s = as_stream(stream);
    if (n_chunks == 0) { put_aff(h_out_aff, g1_inf()); return GM_OK; }
    GM_REQUIRE(n_chunks * ((1ull << gamma) - 1) < (1ull << 31), "table too large");
    G1Scratch& ws = g1_scratch();
    std::lock_guard<std::mutex> lock(ws.mu);
    int32_t rc = ws.reserve(2 * al(n_chunks * 4 + 4) + al(sizeof(G1Jac)) + g1_engine_bytes(n_chunks, 1) + 8192);
    if (rc) return rc;
    ws.used = 0;
    uint32_t* keys = (uint32_t*)ws.carve(n_chunks * 4 + 4);
    uint32_t* idx = (uint32_t*)ws.carve(n_chunks * 4 + 4);
    G1Jac* d_r = (G1Jac*)ws.carve(sizeof(G1Jac));
    hipLaunchKernelGGL(k_g1_binary_tasks, dim3(ceil_div(n_chunks, 256)), dim3(256), 0, s, d_coefs, n_chunks, (1u << gamma) - 1, keys,
                       idx);
    GM_LAUNCH_CHECK();
    rc = g1_sum_by_key(ws, reinterpret_cast<const G1Aff*>(d_tables_aff), nullptr, keys, idx, n_chunks, 1, d_r, s);
    if (rc) return rc;
    G1Jac r;
    GM_HIP(hipMemcpyAsync(&r, d_r, sizeof(G1Jac), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    put_aff(h_out_aff, r);
    return GM_OK;
}

// PushForwardState::new, G1 part (pushforward.rs:395-456, 504-524) from the digits / counter of the plan's last run.
//   d_basis_aff: kzg_basis(), >= 2^(x_logsize + clm) affine points
//   d_d_outer:  n_mat * 2^d_logsize Jacobian points, d_c_outer: n_mat * c_stride (c_stride = longest bucket row of the run,
//               returned in *c_stride; the reference's c_outer_buckets[m] is the prefix of length c_upper_bound[m] <= c_stride,
//               the remaining entries are the point at infinity); h_d_comm / h_c_comm: n_mat affine points.
// shared by gm_msm_g1_outer (whole key, all windows) and gm_msm_g1_outer_part (one rank's windows and key slices)
static int32_t g1_outer_core(const gm_msm_plan* plan, const G1Aff* basis, const G1Slots& sl, uint32_t clm, G1Jac* d_d_outer, G1Jac* d_c_outer,
                             uint64_t c_outer_cap, uint32_t* c_stride, std::vector<G1Jac>* d_sums, std::vector<G1Jac>* c_sums, hipStream_t s) {
    const uint64_t N = plan->N;
    const uint32_t ny = plan->nwin, nd = 1u << plan->d_log;
    const uint32_t m0 = plan->y0 >> clm, n_mat = ((plan->y1 - 1) >> clm) - m0 + 1;   // the matrices this plan's windows touch
    const uint64_t ntasks = N * ny;
    // longest bucket row = max counter + 1
    std::vector<uint32_t> rl(plan->nrows);
    GM_HIP(hipMemcpyAsync(rl.data(), plan->row_len, (size_t)plan->nrows * 4, hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    uint32_t cmax = 1;
    for (uint32_t v : rl) if (v > cmax) cmax = v;
    *c_stride = cmax;
    GM_REQUIRE((uint64_t)n_mat * cmax <= c_outer_cap, "c_outer buffer too small: need %llu points", (unsigned long long)n_mat * cmax);
    G1Scratch& ws = g1_scratch();
    std::lock_guard<std::mutex> lock(ws.mu);
    const uint32_t kmax = n_mat * (cmax > nd ? cmax : nd);
    size_t need = 2 * al(ntasks * 4) + g1_engine_bytes(ntasks, kmax) + 8192;
    const size_t wneed = g1_weighted_bytes(n_mat, cmax > nd ? cmax : nd);
    if (wneed > need) need = wneed;
    int32_t rc = ws.reserve(need);
    if (rc) return rc;
    for (int which = 0; which < 2; which++) {
        ws.used = 0;
        uint32_t* keys = (uint32_t*)ws.carve(ntasks * 4);
        uint32_t* idx = (uint32_t*)ws.carve(ntasks * 4);
        const uint32_t stride = which ? cmax : nd;
        if (which == 0)
            hipLaunchKernelGGL((k_g1_outer_tasks<uint16_t>), dim3(ceil_div(ntasks, 256)), dim3(256), 0, s, plan->digits, N, plan->y0, ny, clm, m0, sl,
                               stride, keys, idx);
        else
            hipLaunchKernelGGL((k_g1_outer_tasks<uint32_t>), dim3(ceil_div(ntasks, 256)), dim3(256), 0, s, plan->counter, N, plan->y0, ny, clm, m0, sl,
                               stride, keys, idx);
        GM_LAUNCH_CHECK();
        G1Jac* outp = which ? d_c_outer : d_d_outer;
        rc = g1_sum_by_key(ws, basis, nullptr, keys, idx, ntasks, n_mat * stride, outp, s);
        if (rc) return rc;
        std::vector<G1Jac>* sums = which ? c_sums : d_sums;
        if (sums) {
            ws.used = 0;
            rc = g1_weighted_sums(ws, outp, n_mat, stride, sums, s);
            if (rc) return rc;
        }
    }
    return GM_OK;
}

extern "C" int32_t gm_msm_g1_outer(const gm_msm_plan* plan, const uint64_t* d_basis_aff, uint32_t clm, uint64_t* d_d_outer,
                                   uint64_t* d_c_outer, uint64_t c_outer_cap, uint32_t* c_stride, uint64_t* h_d_comm,
                                   uint64_t* h_c_comm, void* stream) {
    GM_REQUIRE(plan && d_basis_aff && d_d_outer && d_c_outer && c_stride, "null argument");
    GM_REQUIRE(plan->y0 == 0 && plan->nwin == plan->y_size, "the outer buckets need a plan over all windows (sharded: gm_msm_g1_outer_part)");
    GM_REQUIRE(clm <= 8 && plan->x_log + clm < 31, "bad commitment_log_multiplicity");
    G1Slots sl;
    for (uint32_t i = 0; i < 256; i++) sl.slot[i] = (uint16_t)i;   // the whole key: slice s at s
    std::vector<G1Jac> ds, cs;
    const int32_t rc = g1_outer_core(plan, reinterpret_cast<const G1Aff*>(d_basis_aff), sl, clm, reinterpret_cast<G1Jac*>(d_d_outer),
                                     reinterpret_cast<G1Jac*>(d_c_outer), c_outer_cap, c_stride, h_d_comm ? &ds : nullptr,
                                     h_c_comm ? &cs : nullptr, as_stream(stream));
    if (rc) return rc;
    for (size_t m = 0; m < ds.size(); m++) put_aff(h_d_comm + 12 * m, ds[m]);
    for (size_t m = 0; m < cs.size(); m++) put_aff(h_c_comm + 12 * m, cs[m]);
    return GM_OK;
}

// One rank's part of the same, for a window-sharded plan (commitment_log_multiplicity > 0 makes one commitment matrix
// span 2^clm windows, i.e. several ranks: pushforward.rs:395-396, 431-456).  The rank holds only the key slices its windows use:
//   d_basis_local   n_slots * 2^x_logsize affine points; h_slot[s] (s < 2^clm) = where slice s (kzg_basis[s * 2^x .. (s + 1) * 2^x))
//                   sits in it, or -1 when the rank does not hold it (using such a slice is an error)
//   d_d_outer / d_c_outer   this rank's PARTIAL outer buckets of the matrices m0 .. m0 + n_mat_local - 1 its windows touch
//                   (m0 = y_begin >> clm), n_mat_local * 2^d_logsize and n_mat_local * c_stride Jacobian points
//   h_d_part / h_c_part     n_mat_local Jacobian points each: the rank's share of d_comm[m], c_comm[m] = sum_i i * bucket_i
// Everything the protocol does with the outer buckets is LINEAR in them (the weighted sums here, the eq-weighted MSMs of the second
// phase, pushforward.rs:599-604), so what crosses GPUs is one group element per matrix and commitment: gm_g1_combine_parts.
extern "C" int32_t gm_msm_g1_outer_part(const gm_msm_plan* plan, const uint64_t* d_basis_local, const int32_t* h_slot, uint32_t clm,
                                        uint64_t* d_d_outer, uint64_t* d_c_outer, uint64_t c_outer_cap, uint32_t* c_stride,
                                        uint32_t* first_matrix, uint32_t* n_matrices, uint64_t* h_d_part_jac, uint64_t* h_c_part_jac,
                                        void* stream) {
    GM_REQUIRE(plan && d_basis_local && h_slot && d_d_outer && d_c_outer && c_stride && h_d_part_jac && h_c_part_jac, "null argument");
    GM_REQUIRE(clm <= 8, "bad commitment_log_multiplicity");
    G1Slots sl;
    uint32_t n_slots = 0;
    for (uint32_t i = 0; i < 256; i++) sl.slot[i] = 0;
    for (uint32_t y = plan->y0; y < plan->y1; y++) {
        const int32_t v = h_slot[y & ((1u << clm) - 1)];
        GM_REQUIRE(v >= 0 && v < 256, "window %u needs key slice %u, which this rank does not hold", y, y & ((1u << clm) - 1));
        sl.slot[y & ((1u << clm) - 1)] = (uint16_t)v;
        if ((uint32_t)v + 1 > n_slots) n_slots = (uint32_t)v + 1;
    }
    GM_REQUIRE(((uint64_t)n_slots << plan->x_log) < (1ull << 32), "local key too large for 32-bit point indices");
    std::vector<G1Jac> ds, cs;
    const int32_t rc = g1_outer_core(plan, reinterpret_cast<const G1Aff*>(d_basis_local), sl, clm, reinterpret_cast<G1Jac*>(d_d_outer),
                                     reinterpret_cast<G1Jac*>(d_c_outer), c_outer_cap, c_stride, &ds, &cs, as_stream(stream));
    if (rc) return rc;
    if (first_matrix) *first_matrix = plan->y0 >> clm;
    if (n_matrices) *n_matrices = (uint32_t)ds.size();
    memcpy(h_d_part_jac, ds.data(), ds.size() * sizeof(G1Jac));
    memcpy(h_c_part_jac, cs.data(), cs.size() * sizeof(G1Jac));
    return GM_OK;
}

// The cross-GPU EC combine: every rank contributes n Jacobian points (the point at infinity where it has nothing), the ranks'
// contributions are all-gathered (n * 144 bytes per rank) and added per slot on the host; h_out_aff: n affine points, the same on
// every rank.  Group addition is exact and commutative: the result does not depend on the order of the ranks.
extern "C" int32_t gm_g1_combine_parts(const gm_comm* comm, const uint64_t* h_parts_jac, uint32_t n, uint64_t* h_out_aff) {
    GM_REQUIRE(comm && comm->all_gather && h_parts_jac && h_out_aff && n >= 1, "bad argument");
    const size_t bytes = (size_t)n * sizeof(G1Jac);
    std::vector<char> all((size_t)comm->world * bytes, 0);
    memcpy(all.data() + (size_t)comm->rank * bytes, h_parts_jac, bytes);
    const int32_t rc = comm_all_gather(comm, all.data(), bytes);
    if (rc) return set_err(GM_ERR_STATE, "gm_comm all_gather failed with %d", rc);
    const G1Jac* a = reinterpret_cast<const G1Jac*>(all.data());
    for (uint32_t i = 0; i < n; i++) {
        G1Jac acc = g1_inf();
        for (uint32_t r = 0; r < comm->world; r++) acc = g1_add(acc, a[(size_t)r * n + i]);
        put_aff(h_out_aff + 12 * (size_t)i, acc);
    }
    return GM_OK;
}

// ---- gen-1 column commitments (gkr_msm_prove, gkr_msm_simple.rs:117-151): bit columns through binary_msm, the point column
// through a plain MSM.  prepare_coefs (binary_msm.rs:51-53) packs gamma bits MSB-first into one byte per chunk.
namespace gm {
__global__ void __launch_bounds__(256) k_g1_pack_coefs(const uint8_t* __restrict__ bits, uint64_t nbits, uint32_t gamma,
                                                        uint64_t nchunks, uint8_t* __restrict__ coefs) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint32_t v = 0;
    for (uint32_t b = 0; b < gamma; b++) {
        const uint64_t i = t * gamma + b;
        if (i < nbits) v = (v << 1) + (bits[i] ? 1u : 0u);   // into_u8: s = (s << 1) + bit, over the (possibly short) chunk
    }
    coefs[t] = (uint8_t)v;
}
// all bit columns at once: task (column, chunk) -> key column, table entry of the chunk's packed bits (prepare_coefs + binary_msm's
// lookup, binary_msm.rs:19-29, 51-53); zero chunks produce no task
__global__ void __launch_bounds__(256) k_g1_binary_tasks_cols(const uint8_t* __restrict__ bits, uint64_t col_size, uint32_t gamma,
                                                               uint64_t nchunks, uint32_t ncols, uint32_t tab_len,
                                                               uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks * ncols) return;
    const uint64_t col = t / nchunks, ch = t % nchunks;
    const uint8_t* b8 = bits + col * col_size;
    uint32_t v = 0;
    for (uint32_t b = 0; b < gamma; b++) {
        const uint64_t i = ch * gamma + b;
        if (i < col_size) v = (v << 1) + (b8[i] ? 1u : 0u);
    }
    keys[t] = v ? (uint32_t)col : ncols;
    idx[t] = (uint32_t)(ch * tab_len + (v ? v - 1 : 0));
}
// pts_prep = x coordinates, then y coordinates, then zeros up to col_size (gkr_msm_simple.rs:139-146)
__global__ void __launch_bounds__(256) k_g1_pts_prep(const Fr* __restrict__ points_xy, uint64_t npts, uint64_t col_size,
                                                      Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= col_size) return;
    Fr v = fr_zero();
    if (i < npts) v = fr_load(points_xy + 2 * i);
    else if (i < 2 * npts) v = fr_load(points_xy + 2 * (i - npts) + 1);
    fr_store(out + i, v);
}
}  // namespace gm

extern "C" int32_t gm_g1_binary_msm(const uint8_t* d_coefs, const uint64_t* d_tables_aff, uint64_t n_chunks, uint32_t gamma,
                                    uint64_t* h_out_aff, void* stream);

extern "C" int32_t gm_gkr_msm_commit(const uint64_t* d_points_xy, const uint8_t* d_scalar_bits, uint32_t log_num_points,
                                     uint32_t log_num_scalar_bits, uint32_t log_num_bit_columns, const uint64_t* d_bases_aff,
                                     const uint64_t* d_binary_tables_aff, uint32_t gamma, uint64_t* h_bit_comms_aff,
                                     uint64_t* h_pts_comm_aff, void* stream) {
    GM_REQUIRE(d_points_xy && d_scalar_bits && d_bases_aff && d_binary_tables_aff && h_bit_comms_aff && h_pts_comm_aff, "null argument");
    GM_REQUIRE(gamma >= 1 && gamma <= 8, "1 <= gamma <= 8");
    const uint32_t nv = log_num_points + log_num_scalar_bits;
    GM_REQUIRE(nv <= 32 && log_num_bit_columns <= nv, "bad sizes");
    const uint64_t size = 1ull << nv, ncols = 1ull << log_num_bit_columns, col_size = size >> log_num_bit_columns;
    const uint64_t npts = 1ull << log_num_points;
    GM_REQUIRE(col_size >= 2 * npts, "Points should fit in a single column. Please reduce the amount of columns. (gkr_msm_simple.rs:134-137)");
    hipStream_t s = as_stream(stream);
    const uint64_t nchunks = (col_size + gamma - 1) / gamma;
    Fr* prep = nullptr;
    GM_HIP(dev_alloc((void**)&prep, col_size * sizeof(Fr)));
    int32_t rc = GM_OK;
    {   // every bit column in one engine pass (one launch sequence instead of ncols of them)
        const uint64_t ntasks = nchunks * ncols;
        const uint32_t tab_len = (1u << gamma) - 1;
        if (ntasks >= (1ull << 31) || nchunks * tab_len >= (1ull << 31)) rc = set_err(GM_ERR_INVALID, "bit columns too large for one call");
        G1Scratch& ws = g1_scratch();
        std::vector<G1Jac> sums(ncols);
        if (rc == GM_OK) {
            std::lock_guard<std::mutex> lock(ws.mu);
            rc = ws.reserve(2 * al(ntasks * 4 + 4) + al(ncols * sizeof(G1Jac)) + g1_engine_bytes(ntasks, (uint32_t)ncols) + 8192);
            if (rc == GM_OK) {
                ws.used = 0;
                uint32_t* keys = (uint32_t*)ws.carve(ntasks * 4 + 4);
                uint32_t* idx = (uint32_t*)ws.carve(ntasks * 4 + 4);
                G1Jac* d_r = (G1Jac*)ws.carve(ncols * sizeof(G1Jac));
                hipLaunchKernelGGL(k_g1_binary_tasks_cols, dim3(ceil_div(ntasks, 256)), dim3(256), 0, s, d_scalar_bits, col_size, gamma,
                                   nchunks, (uint32_t)ncols, tab_len, keys, idx);
                if (hipGetLastError() != hipSuccess) rc = set_err(GM_ERR_HIP, "k_g1_binary_tasks_cols launch failed");
                if (rc == GM_OK)
                    rc = g1_sum_by_key(ws, reinterpret_cast<const G1Aff*>(d_binary_tables_aff), nullptr, keys, idx, ntasks, (uint32_t)ncols,
                                       d_r, s);
                if (rc == GM_OK) {
                    if (hipMemcpyAsync(sums.data(), d_r, ncols * sizeof(G1Jac), hipMemcpyDeviceToHost, s) != hipSuccess ||
                        hipStreamSynchronize(s) != hipSuccess)
                        rc = set_err(GM_ERR_HIP, "reading the column sums failed");
                }
            }
        }
        if (rc == GM_OK)
            host_parallel_for((uint32_t)ncols, [&](uint32_t i) { put_aff(h_bit_comms_aff + 12 * (size_t)i, sums[i]); }, 4);
    }
    if (rc == GM_OK) {
        hipLaunchKernelGGL(k_g1_pts_prep, dim3(ceil_div(col_size, 256)), dim3(256), 0, s, reinterpret_cast<const Fr*>(d_points_xy), npts,
                           col_size, prep);
        if (hipGetLastError() != hipSuccess) rc = set_err(GM_ERR_HIP, "k_g1_pts_prep launch failed");
    }
    if (rc == GM_OK) rc = gm_g1_msm(d_bases_aff, reinterpret_cast<const uint64_t*>(prep), col_size, 1, 255, h_pts_comm_aff, stream);
    (void)hipStreamSynchronize(s);
    dev_free(prep);
    return rc;
}

extern "C" int32_t gm_g1_to_affine(const uint64_t* d_in_jac, uint64_t n, uint64_t* d_out_aff, void* stream) {
    GM_REQUIRE(d_in_jac && d_out_aff, "null argument");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(k_g1_to_affine, dim3(ceil_div(n, 128)), dim3(128), 0, as_stream(stream),
                       reinterpret_cast<const G1Jac*>(d_in_jac), n, reinterpret_cast<G1Aff*>(d_out_aff));
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int32_t gm_g1_from_affine(const uint64_t* d_in_aff, uint64_t n, uint64_t* d_out_jac, void* stream) {
    GM_REQUIRE(d_in_aff && d_out_jac, "null argument");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(k_g1_from_affine, dim3(ceil_div(n, 128)), dim3(128), 0, as_stream(stream),
                       reinterpret_cast<const G1Aff*>(d_in_aff), n, reinterpret_cast<G1Jac*>(d_out_jac));
    GM_LAUNCH_CHECK();
    return GM_OK;
}
