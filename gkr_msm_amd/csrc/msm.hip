// Pippenger MSM over Bandersnatch for gfx950: digits -> stable bucket scatter -> bucket sums by the
// reference's pairwise tree -> bucket reduction ("triangle") -> window points.
//
// What is restated (see include/gkrmsm.h for the seam list):
//   digits / bucket scatter   /root/reference/src/cleanup/protocols/pushforward/pushforward.rs:351-361, 401-429
//   bucket sums               /root/reference/src/cleanup/protocols/gkrs/bintree_add.rs:137-239 (+ vecvec.rs:542-654)
//   bucket reduction          /root/reference/src/cleanup/protocols/gkrs/triangle_add.rs:101-158,
//                             /root/reference/src/cleanup/protocols/pippenger_ending.rs:46-58
//   recombination             /root/reference/src/cleanup/protocols/pippenger.rs:586-602
//
// MI355X mapping.  The bucket image is a ragged matrix: row (y << d | digit) holds the points whose
// window-y digit is `digit`, in x order, padded to even length with the identity (pushforward.rs:380-381,
// vecvec.rs:181-186).  The reference adds elements (2i, 2i+1) of every row, level by level, x_logsize
// times; projective coordinates of the bucket sums depend on exactly that association, so it is kept.
// Rows are stored back to back (offsets always even), which makes every level a flat, perfectly
// load-balanced map over output cells regardless of how skewed the bucket populations are; a thread
// finds its row by binary search in the offsets table.  Each level is one launch; the three layer
// functions l1/l2/l3 are fused in registers (the per-layer outputs are only materialised by the
// witness builder in witness.hip, which the sumcheck needs; the MSM does not).
#include <algorithm>
#include <vector>

#include "algfn.hip.h"
#include "common.hpp"
#include "msm_plan.hpp"
#include "ragged.hip.h"
#include "fr9.hip.h"

namespace gm {

static constexpr int CHUNK = 1024;  // x-range handled by one wave in the scatter passes

struct Point3 {
    Fr x, y, z;
};
typedef uint32_t gm_u4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ Point3 pt_identity() {
    Point3 p;
    p.x = fr_zero();
    p.y = fr_one();
    p.z = fr_one();
    return p;
}

// The MSM only needs the OUTPUT of l3(l2(l1(.))) -- the per-layer polynomials are materialised by the witness builder,
// not here -- so the composition is evaluated with fewer multiplications while producing the same field elements:
//   x1y2 + x2y1 = (x1 + y1)(x2 + y2) - x1x2 - y1y2          (one product instead of two)
//   (x1y2)(x2y1) = (x1x2)(y1y2)                              (reuses the two products l1 needs anyway)
// Field arithmetic is exact, so X, Y, Z are bit-identical to the layer-by-layer evaluation
// (twisted_edwards_ops.rs:10-65); 8 instead of 9 multiplications (affine), 12 instead of 13 (projective).

// (x1,y1) + (x2,y2) = affine l3(l2(l1(.)))  (bintree level 0)
__device__ __forceinline__ Point3 aff_add(const Fr& x1, const Fr& y1, const Fr& x2, const Fr& y2) {
    const Fr A = fr_mul(x1, x2), B = fr_mul(y1, y2);
    const Fr s = fr_sub(fr_sub(fr_mul(fr_add(x1, y1), fr_add(x2, y2)), A), B);  // x1y2 + x2y1
    const Fr t = fr_sub(B, fr_mul_by_a(A));                                     // y1y2 - a x1x2
    const Fr dxy = fr_mul_by_d(fr_mul(A, B));                                   // d (x1y2)(x2y1)
    const Fr m = fr_sub(fr_one(), dxy), q = fr_add(fr_one(), dxy);
    Point3 r;
    r.x = fr_mul(m, s); r.y = fr_mul(q, t); r.z = fr_mul(m, q);
    return r;
}

// projective l3(l2(l1(.)))  (bintree levels >= 1, triangle)
__device__ __forceinline__ Point3 proj_add(const Point3& p, const Point3& g) {
    const Fr A = fr_mul(p.x, g.x), B = fr_mul(p.y, g.y), zz = fr_mul(p.z, g.z);
    const Fr s = fr_sub(fr_sub(fr_mul(fr_add(p.x, p.y), fr_add(g.x, g.y)), A), B);  // x1y2 + x2y1
    const Fr t = fr_sub(B, fr_mul_by_a(A));
    const Fr X = fr_mul(s, zz), Y = fr_mul(t, zz), z2 = fr_sqr(zz);
    const Fr dxy = fr_mul_by_d(fr_mul(A, B));
    const Fr m = fr_sub(z2, dxy), q = fr_add(z2, dxy);
    Point3 r;
    r.x = fr_mul(m, X); r.y = fr_mul(q, Y); r.z = fr_mul(m, q);
    return r;
}

// The same two compositions in the 9 x 29-bit form (fr9.hip.h): this is what the level kernels run.  Identical field values --
// the results are canonicalised when stored -- with 205-instruction products, carry-free sums and no conditional subtraction
// inside the formula.  Bounds per line: L = limb bound, S = value / p bound (2^261 / p = 70.66).
struct Point9 {
    Fr9 x, y, z;
};

// inputs: loaded values (L 2^29, S 32) or the identity's constants
__device__ __forceinline__ Point9 aff_add9(const Fr9& x1, const Fr9& y1, const Fr9& x2, const Fr9& y2) {
    const Fr9 A = fr9_mul(x1, x2), B = fr9_mul(y1, y2);                      // L 2^29, S 15.5
    const Fr9 C = fr9_mul(fr9_add(x1, y1), fr9_add(x2, y2));                 // operands L 2^30, S 64 -> S 59
    const Fr9 s = fr9_sub2_32(C, A, B);                                      // x1y2 + x2y1: L 2^31, S 91
    const Fr9 t = fr9_norm(fr9_add(B, fr9_mul5(A)));                         // y1y2 - a x1x2, a = -5: S 93, top limb < 2^29.4
    const Fr9 dxy = fr9_mul(fr9_mul(A, B), fr9_coeff_d());                   // S 4.4 -> S 1.07
    const Fr9 m = fr9_norm(fr9_sub8(fr9_one(), dxy));                        // S 9
    const Fr9 q = fr9_add(fr9_one(), dxy);                                   // L 2^30, S 2.07
    Point9 r;
    r.x = fr9_mul(m, s);                                                     // 2^29 x 2^31; S 12.6
    r.y = fr9_mul(q, t);                                                     // 2^30 x 2^29.4; S 3.7
    r.z = fr9_mul(m, q);                                                     // S 1.3
    return r;
}

// inputs: loaded coordinates (L 2^29, S 32)
__device__ __forceinline__ Point9 proj_add9(const Point9& p, const Point9& g) {
#ifdef GM_ADD9_PAIRED
    // independent products two at a time, their multiply-add chains interleaved (fr9_mul2)
    Fr9 A, B, zz, C, X, Y, AB, dxy;
    fr9_mul2(p.x, g.x, p.y, g.y, A, B);                                                // S 15.5
    fr9_mul2(p.z, g.z, fr9_add(p.x, p.y), fr9_add(g.x, g.y), zz, C);                   // S 15.5, 59
    const Fr9 s = fr9_sub2_32(C, A, B);                                                // L 2^31, S 91
    const Fr9 t = fr9_add(B, fr9_mul5(A));                                             // L 6 2^29, S 93
    fr9_mul2(s, zz, t, zz, X, Y);                                                      // S 21, 21.4
    const Fr9 z2 = fr9_sqr(zz);                                                        // S 4.4
    AB = fr9_mul(A, B);
    dxy = fr9_mul(AB, fr9_coeff_d());                                                  // S 1.07
    const Fr9 m = fr9_norm(fr9_sub8(z2, dxy));                                         // S 12.4
    const Fr9 q = fr9_add(z2, dxy);                                                    // L 2^30, S 5.5
    Point9 r;
    fr9_mul2(m, X, q, Y, r.x, r.y);                                                    // S 4.7, 2.7
    r.z = fr9_mul(m, q);                                                               // S 2
    return r;
#else
    const Fr9 A = fr9_mul(p.x, g.x), B = fr9_mul(p.y, g.y), zz = fr9_mul(p.z, g.z);   // S 15.5
    const Fr9 C = fr9_mul(fr9_add(p.x, p.y), fr9_add(g.x, g.y));                       // S 59
    const Fr9 s = fr9_sub2_32(C, A, B);                                                // L 2^31, S 91
    const Fr9 t = fr9_add(B, fr9_mul5(A));                                             // L 6 2^29, S 93
    const Fr9 X = fr9_mul(s, zz);                                                      // 2^31 x 2^29; S 21
    const Fr9 Y = fr9_mul(t, zz);                                                      // 54 2^58 + 9 2^58 + 2^36 < 2^64; S 21.4
    const Fr9 z2 = fr9_sqr(zz);                                                        // S 4.4
    const Fr9 dxy = fr9_mul(fr9_mul(A, B), fr9_coeff_d());                             // S 1.07
    const Fr9 m = fr9_norm(fr9_sub8(z2, dxy));                                         // S 12.4
    const Fr9 q = fr9_add(z2, dxy);                                                    // L 2^30, S 5.5
    Point9 r;
    r.x = fr9_mul(m, X);                                                               // S 4.7
    r.y = fr9_mul(q, Y);                                                               // S 2.7
    r.z = fr9_mul(m, q);                                                               // S 2
    return r;
#endif
}

__device__ __forceinline__ Point9 pt9_load(const Fr* x, const Fr* y, const Fr* z, uint64_t i) {
    Point9 p;
    p.x = fr9_load(x + i); p.y = fr9_load(y + i); p.z = fr9_load(z + i);
    return p;
}
__device__ __forceinline__ void pt9_store(Fr* x, Fr* y, Fr* z, uint64_t i, const Point9& p) {
    fr9_store(x + i, p.x); fr9_store(y + i, p.y); fr9_store(z + i, p.z);
}
// Cells that one level kernel writes and the next one reads stay in the 9 x 29 form, exactly as the products leave them (limbs
// < 2^29, value below 13 p: inside what the formulas assume of a loaded value, L 2^29 and S 32): no division by 32, no
// conditional subtraction and no repacking on the way out (~100 VALU instructions per coordinate), no shifting on the way in (17).
// 9 words = 36 bytes per cell, moved as 16 + 16 + 4 bytes (4-byte aligned: global memory takes unaligned wide accesses).
// Only the last level canonicalises (its output is the bucket sums).
struct __attribute__((packed, aligned(4))) Raw9Quad {
    uint32_t a, b, c, d;
};
__device__ __forceinline__ Fr9 fr9_load_raw9(const uint32_t* __restrict__ col, uint64_t i) {
    const uint32_t* p = col + 9 * i;
    const Raw9Quad q0 = *reinterpret_cast<const Raw9Quad*>(p), q1 = *reinterpret_cast<const Raw9Quad*>(p + 4);
    Fr9 r;
    r.l[0] = q0.a; r.l[1] = q0.b; r.l[2] = q0.c; r.l[3] = q0.d; r.l[4] = q1.a; r.l[5] = q1.b; r.l[6] = q1.c; r.l[7] = q1.d; r.l[8] = p[8];
    return r;
}
__device__ __forceinline__ void fr9_store_raw9(uint32_t* __restrict__ col, uint64_t i, const Fr9& v) {
    uint32_t* p = col + 9 * i;
    Raw9Quad q0, q1;
    q0.a = v.l[0]; q0.b = v.l[1]; q0.c = v.l[2]; q0.d = v.l[3]; q1.a = v.l[4]; q1.b = v.l[5]; q1.c = v.l[6]; q1.d = v.l[7];
    *reinterpret_cast<Raw9Quad*>(p) = q0;
    *reinterpret_cast<Raw9Quad*>(p + 4) = q1;
    p[8] = v.l[8];
}
__device__ __forceinline__ Point9 pt9_load_raw9(const uint32_t* x, const uint32_t* y, const uint32_t* z, uint64_t i) {
    Point9 p;
    p.x = fr9_load_raw9(x, i); p.y = fr9_load_raw9(y, i); p.z = fr9_load_raw9(z, i);
    return p;
}
__device__ __forceinline__ void pt9_store_raw9(uint32_t* x, uint32_t* y, uint32_t* z, uint64_t i, const Point9& p) {
    fr9_store_raw9(x, i, p.x); fr9_store_raw9(y, i, p.y); fr9_store_raw9(z, i, p.z);
}
__device__ __forceinline__ Point9 pt9_identity() {
    Point9 p;
    p.x = fr9_zero(); p.y = fr9_one(); p.z = fr9_one();
    return p;
}
__device__ __forceinline__ void pt_store(Fr* x, Fr* y, Fr* z, uint64_t i, const Point3& p) {
    fr_store(x + i, p.x); fr_store(y + i, p.y); fr_store(z + i, p.z);
}

// ---------------------------------------------------------------------------------------------
// digits[(y - y0) * N + x] = d-bit window y of scalar x      (pushforward.rs:351-361)
__global__ void k_digits(const uint32_t* __restrict__ scalars, uint16_t* __restrict__ digits, uint64_t N,
                         uint32_t d_log, uint32_t y0, uint32_t nwin) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= N) return;
    uint32_t s[9];
    const uint4* q = reinterpret_cast<const uint4*>(scalars + x * 8);
    uint4 lo = q[0], hi = q[1];
    s[0] = lo.x; s[1] = lo.y; s[2] = lo.z; s[3] = lo.w;
    s[4] = hi.x; s[5] = hi.y; s[6] = hi.z; s[7] = hi.w;
    s[8] = 0;
    const uint32_t mask = (1u << d_log) - 1;
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t bit = (y0 + w) * d_log;
        uint32_t v = 0;
        if (bit < 256) {
            uint32_t li = bit >> 5, sh = bit & 31;
            uint64_t two = (uint64_t)s[li] | ((uint64_t)s[li + 1] << 32);
            v = (uint32_t)(two >> sh) & mask;
        }
        digits[(uint64_t)w * N + x] = (uint16_t)v;
    }
}

// per (window, chunk) histogram of digits; one wave per chunk
__global__ void k_hist(const uint16_t* __restrict__ digits, uint32_t* __restrict__ hist, uint64_t N,
                       uint32_t nd, uint32_t nchunks, uint32_t chunk, uint64_t ntasks) {
    extern __shared__ uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t waves = blockDim.x >> 6;
    uint32_t* cnt = lds + wave * nd;
    const uint64_t task = (uint64_t)blockIdx.x * waves + wave;  // = w * nchunks + c
    if (task >= ntasks) return;  // whole wave; only wave-level barriers below
    for (uint32_t i = lane; i < nd; i += 64) cnt[i] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint64_t w = task / nchunks, c = task % nchunks;
    const uint16_t* row = digits + w * N;
    for (uint32_t i = lane; i < chunk; i += 64) {
        const uint64_t x = c * chunk + i;
        if (x < N) atomicAdd(&cnt[row[x]], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t* out = hist + task * nd;
    for (uint32_t i = lane; i < nd; i += 64) out[i] = cnt[i];
}

// exclusive scan over chunks for every (window, digit); totals -> row_len.
// hist[w][chunk][digit]: one 1024-thread block owns G = min(nd, 64) consecutive digits of one window; lane d walks digit
// d (coalesced across the G lanes) over 1 / PARTS of the chunks, the PARTS partial sums are combined through LDS, and a
// second walk writes the exclusive prefixes.  (A single thread per (window, digit) walking all chunks took 0.24 ms.)
__global__ void __launch_bounds__(1024) k_scan_chunks(uint32_t* __restrict__ hist, uint32_t* __restrict__ row_len, uint32_t nd,
                                                       uint32_t nchunks, uint32_t nrows) {
    __shared__ uint32_t part_sum[1024];
    const uint32_t G = nd < 64 ? nd : 64, parts = 1024 / G;
    const uint32_t groups_per_win = nd / G;
    const uint32_t w = blockIdx.x / groups_per_win, g = blockIdx.x % groups_per_win;
    const uint32_t d = threadIdx.x % G, part = threadIdx.x / G;
    const uint32_t dg = g * G + d;
    const uint32_t per = (nchunks + parts - 1) / parts;
    const uint32_t c0 = part * per, c1 = (c0 + per < nchunks) ? c0 + per : nchunks;
    uint32_t* p = hist + (uint64_t)w * nchunks * nd + dg;
    uint32_t sum = 0;
    for (uint32_t c = c0; c < c1; c++) sum += p[(uint64_t)c * nd];
    part_sum[threadIdx.x] = sum;
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t q = 0; q < part; q++) acc += part_sum[q * G + d];
    for (uint32_t c = c0; c < c1; c++) {
        const uint32_t v = p[(uint64_t)c * nd];
        p[(uint64_t)c * nd] = acc;
        acc += v;
    }
    if (part == parts - 1) row_len[w * nd + dg] = acc;
}

// Exclusive scan of per-row lengths into row offsets, one 1024-thread block: every thread sums a contiguous
// chunk of rows serially, the 1024 chunk sums are scanned with wave shuffles, then the chunk is written out.
//   MODE 0: v(r) = pad2(len[r])                      image rows (vecvec.rs:181-186)
//   MODE 1: v(r) = pad2((off_in[r+1] - off_in[r])/2)  next level / next round (vecvec.rs:579-594, 420-441)
template <int MODE>
__device__ __forceinline__ uint32_t row_value(const uint32_t* __restrict__ src, uint32_t r) {
    if (MODE == 0) { const uint32_t l = src[r]; return l + (l & 1u); }
    const uint32_t h = (src[r + 1] - src[r]) >> 1;
    return h + (h & 1u);
}

template <int MODE>
__global__ void __launch_bounds__(1024) k_offsets_scan(const uint32_t* __restrict__ src, uint32_t* __restrict__ off,
                                                        uint32_t nrows) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (nrows + 1023) / 1024;
    const uint32_t r0 = tid * per;
    const uint32_t r1 = (r0 + per < nrows) ? r0 + per : nrows;
    uint32_t sum = 0;
    for (uint32_t r = r0; r < r1; r++) sum += row_value<MODE>(src, r);
    // inclusive scan inside the wave
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
    uint32_t run = base + inc - sum;  // exclusive prefix of this thread's chunk
    for (uint32_t r = r0; r < r1; r++) {
        off[r] = run;
        run += row_value<MODE>(src, r);
    }
    if (tid == 1023) {
        uint32_t tot = 0;
        for (int w = 0; w < 16; w++) tot += wave_tot[w];
        off[nrows] = tot;
    }
}

// Stable rank of every x inside its bucket row: counter[w][x] = #{x' < x : digit[w][x'] == digit[w][x]}
// (pushforward.rs:411-426), and the scatter of x into the row (cells[off[row] + counter] = x).
// One wave walks one chunk in x order; equal-digit lanes are found with d_log ballots.
__global__ void k_rank_scatter(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ chunk_base,
                               const uint32_t* __restrict__ off, uint32_t* __restrict__ counter,
                               uint32_t* __restrict__ cells, uint64_t N, uint32_t d_log, uint32_t nchunks,
                               uint32_t chunk, uint64_t ntasks) {
    extern __shared__ uint32_t lds[];
    const uint32_t nd = 1u << d_log;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t waves = blockDim.x >> 6;
    uint32_t* cnt = lds + wave * nd;
    const uint64_t task = (uint64_t)blockIdx.x * waves + wave;
    if (task >= ntasks) return;  // whole wave exits together; no block-level barrier below
    const uint64_t w = task / nchunks, c = task % nchunks;
    const uint32_t* base = chunk_base + task * nd;
    for (uint32_t i = lane; i < nd; i += 64) cnt[i] = base[i];
    __builtin_amdgcn_wave_barrier();
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t i = 0; i < chunk; i += 64) {
        const uint64_t x = c * chunk + i + lane;
        const bool valid = x < N;
        const uint32_t dg = valid ? digits[w * N + x] : 0u;
        uint64_t peers = __ballot(valid);
        for (uint32_t b = 0; b < d_log; b++) {
            uint64_t m = __ballot((dg >> b) & 1u);
            peers &= ((dg >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = __popcll(peers & lane_lt);
        const uint32_t pos = cnt[dg] + before;
        __builtin_amdgcn_wave_barrier();
        // the highest peer lane publishes the new count
        if (valid && (peers >> lane) == 1ull) cnt[dg] = pos + 1;
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            counter[w * N + x] = pos;
            const uint32_t row = (uint32_t)w * nd + dg;
            cells[(uint64_t)off[row] + pos] = (uint32_t)x;
        }
    }
}

// The same ranking with the scatter regrouped: a workgroup of four waves owns four consecutive chunks of ONE window (4096 points),
// ranks them as above, regroups the tile by digit in LDS and writes every digit's run out contiguously (16 points per bucket on
// average at config B = one 64-byte line) instead of 4 bytes per line: k_rank_scatter wrote 1.14 GB through L2 for 128 MB of
// cell indices.  counter[] is written in x order as before.  Needs chunk == 1024, nchunks % 4 == 0, nd <= 1024.
__global__ void __launch_bounds__(256) k_rank_scatter_tile(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ chunk_base,
                                                           const uint32_t* __restrict__ off, uint32_t* __restrict__ counter,
                                                           uint32_t* __restrict__ cells, uint64_t N, uint32_t d_log, uint32_t nchunks) {
    extern __shared__ uint32_t lds[];
    const uint32_t nd = 1u << d_log;
    uint32_t* cnt = lds;                       // [4][nd] running position of every digit, per wave
    uint32_t* base0 = cnt + 4 * nd;            // [nd] position of the tile's first point of digit d in its row
    uint32_t* dig_start = base0 + nd;          // [nd + 1] first slot of digit d in the regrouped tile
    uint32_t* st_val = dig_start + nd + 1;     // [4096]
    uint16_t* st_dig = reinterpret_cast<uint16_t*>(st_val + 4096);   // [4096]
    __shared__ uint32_t wave_tot[4];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint64_t task = (uint64_t)blockIdx.x * 4 + wave;   // = w * nchunks + c; the four tasks of a block share w
    const uint64_t w = ((uint64_t)blockIdx.x * 4) / nchunks, c = task % nchunks;
    const uint32_t* base = chunk_base + task * nd;
    for (uint32_t i = lane; i < nd; i += 64) cnt[wave * nd + i] = base[i];
    if (wave == 0) for (uint32_t i = lane; i < nd; i += 64) base0[i] = base[i];
    __builtin_amdgcn_wave_barrier();
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t xs[16], ps[16];
    uint16_t ds[16];
#pragma unroll
    for (uint32_t st = 0; st < 16; st++) {
        const uint64_t x = c * 1024 + st * 64 + lane;
        const bool valid = x < N;
        const uint32_t dg = valid ? digits[w * N + x] : 0u;
        uint64_t peers = __ballot(valid);
        for (uint32_t b = 0; b < d_log; b++) {
            const uint64_t m = __ballot((dg >> b) & 1u);
            peers &= ((dg >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = __popcll(peers & lane_lt);
        const uint32_t pos = cnt[wave * nd + dg] + before;
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers >> lane) == 1ull) cnt[wave * nd + dg] = pos + 1;
        __builtin_amdgcn_wave_barrier();
        if (valid) counter[w * N + x] = pos;
        xs[st] = valid ? (uint32_t)x : 0xffffffffu;
        ps[st] = pos;
        ds[st] = (uint16_t)dg;
    }
    __syncthreads();
    // points of digit d in the tile = where the last wave's counter ended minus where the first one's began
    uint32_t mine[4], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t d = tid * 4 + k;   // 4 consecutive digits per thread (nd <= 1024)
        mine[k] = d < nd ? cnt[3 * nd + d] - base0[d] : 0u;
        sum += mine[k];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t t = __shfl_up(inc, dd, 64);
        if ((int)lane >= dd) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t k = 0; k < wave; k++) run += wave_tot[k];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t d = tid * 4 + k;
        if (d < nd) dig_start[d] = run;
        run += mine[k];
    }
    if (tid == 255) dig_start[nd] = run;
    __syncthreads();
#pragma unroll
    for (uint32_t st = 0; st < 16; st++) {
        if (xs[st] != 0xffffffffu) {
            const uint32_t slot = dig_start[ds[st]] + (ps[st] - base0[ds[st]]);
            st_val[slot] = xs[st];
            st_dig[slot] = ds[st];
        }
    }
    __syncthreads();
    const uint32_t total = dig_start[nd];
    for (uint32_t p = tid; p < total; p += 256) {
        const uint32_t dg = st_dig[p];
        const uint32_t row = (uint32_t)w * nd + dg;
        cells[(uint64_t)off[row] + base0[dg] + (p - dig_start[dg])] = st_val[p];
    }
}

// identity padding of odd rows (vecvec.rs:181-186)
__global__ void k_pad_cells(const uint32_t* __restrict__ row_len, const uint32_t* __restrict__ off,
                            uint32_t* __restrict__ cells, uint32_t nrows) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    uint32_t l = row_len[r];
    if (l & 1u) cells[(uint64_t)off[r] + l] = PAD_IDX;
}

// First row of every 128-cell block of every level's OUTPUT layout, all levels in one launch: the level kernels then bracket
// their rows with two loads instead of two 13-step binary searches of dependent loads at the head of every workgroup.
struct BlockRowsArgs {
    uint32_t nlev;
    uint32_t first[33];   // first entry of level l in blk_row; first[nlev] = total entries
};
__global__ void __launch_bounds__(256) k_block_rows(BlockRowsArgs a, const uint32_t* __restrict__ off_all, uint32_t nrows,
                                                    uint32_t* __restrict__ blk_row) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a.first[a.nlev]) return;
    uint32_t l = 0;
    while (l + 1 < a.nlev && e >= a.first[l + 1]) l++;
    const uint32_t* off = off_all + (uint64_t)(l + 1) * (nrows + 1);   // output layout of level l
    const uint32_t j = (e - a.first[l]) * 128, total = off[nrows];
    blk_row[e] = j < total ? find_row(off, nrows, j) : nrows - 1;
}
// row of cell j for a thread of block blockIdx.x, from the block-row table of its level
__device__ __forceinline__ uint32_t find_row_tab(const uint32_t* __restrict__ off, uint32_t nrows, const uint32_t* __restrict__ br, uint32_t j) {
    uint32_t lo = br[blockIdx.x], hi = br[blockIdx.x + 1] + 1;  // off[lo] <= j < off[hi]
    if (hi > nrows) hi = nrows;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// bintree level 0: gather affine points by index, add pairs (2p, 2p+1) of every row
__global__ void k_add_level0(const Fr* __restrict__ points_xy, const uint32_t* __restrict__ cells,
                             const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_out,
                             uint32_t nrows, uint32_t* __restrict__ ox, uint32_t* __restrict__ oy, uint32_t* __restrict__ oz,
                             const uint32_t* __restrict__ br) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = off_out[nrows];
    if (j >= total) return;
    const uint32_t r = find_row_tab(off_out, nrows, br, j);
    const uint32_t p = j - off_out[r];
    const uint32_t in0 = off_in[r], half = (off_in[r + 1] - in0) >> 1;
    if (p < half) {
        const uint32_t i0 = cells[(uint64_t)in0 + 2 * p], i1 = cells[(uint64_t)in0 + 2 * p + 1];
        const Fr9 x1 = fr9_load(points_xy + 2ull * i0), y1 = fr9_load(points_xy + 2ull * i0 + 1);
        Fr9 x2, y2;
        if (i1 != PAD_IDX) {
            x2 = fr9_load(points_xy + 2ull * i1);
            y2 = fr9_load(points_xy + 2ull * i1 + 1);
        } else {
            x2 = fr9_zero();
            y2 = fr9_one();
        }
        pt9_store_raw9(ox, oy, oz, j, aff_add9(x1, y1, x2, y2));
    } else {
        pt9_store_raw9(ox, oy, oz, j, pt9_identity());  // f(row_pad): l3(l2(l1(0,1,0,1))) = (0,1,1)
    }
}

// level-0 output cell c of a row (cells in0 .., `half0` pairs): the sum of its two gathered points, or the image of the row pad
__device__ __forceinline__ Point9 level0_cell(const Fr* __restrict__ points_xy, const uint32_t* __restrict__ cells, uint32_t in0,
                                              uint32_t half0, uint32_t c) {
    if (c >= half0) return pt9_identity();
    const uint32_t i0 = cells[(uint64_t)in0 + 2 * c], i1 = cells[(uint64_t)in0 + 2 * c + 1];
    const Fr9 x1 = fr9_load(points_xy + 2ull * i0), y1 = fr9_load(points_xy + 2ull * i0 + 1);
    Fr9 x2, y2;
    if (i1 != PAD_IDX) {
        x2 = fr9_load(points_xy + 2ull * i1);
        y2 = fr9_load(points_xy + 2ull * i1 + 1);
    } else {
        x2 = fr9_zero();
        y2 = fr9_one();
    }
    return aff_add9(x1, y1, x2, y2);
}

// bintree levels 0 AND 1 in one pass: a lane gathers up to four affine points, adds the two pairs (level 0) and adds their sums (level
// 1), and writes ONE level-1 cell.  The level-0 cells never go to memory: the unfused pair writes 2^(x + 5) of them (108 bytes each)
// and reads them back -- 3.6 of the 9.3 GB the level kernels move per step at config B.  Same operations on the same operands in
// the same order as k_add_level0 followed by k_add_level: the same field elements.
//   off0: layout of the cells (level-0 input), off1: level-0 output = level-1 input, off2: level-1 output; br: block rows of off2
__global__ void __launch_bounds__(128) k_add_level01(const Fr* __restrict__ points_xy, const uint32_t* __restrict__ cells,
                                                      const uint32_t* __restrict__ off0, const uint32_t* __restrict__ off1,
                                                      const uint32_t* __restrict__ off2, uint32_t nrows, uint32_t* __restrict__ ox,
                                                      uint32_t* __restrict__ oy, uint32_t* __restrict__ oz, const uint32_t* __restrict__ br) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = off2[nrows];
    if (j >= total) return;
    const uint32_t r = find_row_tab(off2, nrows, br, j);
    const uint32_t q = j - off2[r];
    const uint32_t half1 = (off1[r + 1] - off1[r]) >> 1;
    if (q >= half1) {
        pt9_store_raw9(ox, oy, oz, j, pt9_identity());
        return;
    }
    const uint32_t in0 = off0[r], half0 = (off0[r + 1] - in0) >> 1;
    // (two calls of one inlined helper, NOT a loop over an array of two points: indexed by the loop variable the array lived in
    // scratch memory -- 216 bytes per lane written and read back, WRITE_SIZE 2.72 GB per launch for 0.92 GB of cells: profiles/r04)
    const Point9 p0 = level0_cell(points_xy, cells, in0, half0, 2 * q);
    const Point9 p1 = level0_cell(points_xy, cells, in0, half0, 2 * q + 1);
    pt9_store_raw9(ox, oy, oz, j, proj_add9(p0, p1));
}

// bintree level >= 1
__global__ void __launch_bounds__(128) k_add_level(const uint32_t* __restrict__ ix, const uint32_t* __restrict__ iy, const uint32_t* __restrict__ iz,
                            const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ off_out,
                            uint32_t nrows, uint32_t* __restrict__ ox, uint32_t* __restrict__ oy, uint32_t* __restrict__ oz,
                            const uint32_t* __restrict__ br) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = off_out[nrows];
    if (j >= total) return;
    const uint32_t r = find_row_tab(off_out, nrows, br, j);
    const uint32_t p = j - off_out[r];
    const uint32_t in0 = off_in[r], half = (off_in[r + 1] - in0) >> 1;
    if (p < half) {
        const uint64_t a = (uint64_t)in0 + 2 * p;
        pt9_store_raw9(ox, oy, oz, j, proj_add9(pt9_load_raw9(ix, iy, iz, a), pt9_load_raw9(ix, iy, iz, a + 1)));
    } else {
        pt9_store_raw9(ox, oy, oz, j, pt9_identity());
    }
}

// Offsets of every level, one workgroup per level: off_all[l * (nrows + 1) + r]; level 0 from the bucket populations (or, FROM_OFF,
// from an offsets table whose even row lengths it reproduces -- the VecVec sumcheck precomputes the row layouts of all its sparse
// rounds this way: bind_21, vecvec.rs:420-441).  The length of a row at level l follows from its level-0 length alone (halve and
// re-pad l times), so the levels do not depend on each other: 20 scans side by side instead of 20 in a row (80 us -> ~14 us at
// config B, on a stretch of the step where nothing else runs).
template <bool FROM_OFF>
__global__ void __launch_bounds__(1024) k_offsets_levels_par(const uint32_t* __restrict__ row_len, uint32_t* __restrict__ off_all,
                                                              uint32_t nrows) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t lvl = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (nrows + 1023) / 1024;
    const uint32_t r0 = tid * per;
    const uint32_t r1 = (r0 + per < nrows) ? r0 + per : nrows;
    auto len_at = [&](uint32_t r) -> uint32_t {
        uint32_t l = FROM_OFF ? row_len[r + 1] - row_len[r] : row_len[r];
        l += l & 1u;
        for (uint32_t k = 0; k < lvl; k++) { const uint32_t h = l >> 1; l = h + (h & 1u); }
        return l;
    };
    uint32_t sum = 0;
    for (uint32_t r = r0; r < r1; r++) sum += len_at(r);
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < 16; w++) { const uint32_t v = wave_tot[w]; tot += v; if (w < wave) base += v; }
    uint32_t* off = off_all + (uint64_t)lvl * (nrows + 1);
    uint32_t run = base + inc - sum;
    for (uint32_t r = r0; r < r1; r++) {
        off[r] = run;
        run += len_at(r);
    }
    if (tid == 1023) off[nrows] = tot;
}

// last bintree level: every row has 0 or 2 cells; output is dense over rows
// (vecvec_map_split_to_dense, vecvec.rs:608-654: an empty row contributes the row pad)
template <bool LEVEL0>
__global__ void k_add_last(const Fr* __restrict__ points_xy, const uint32_t* __restrict__ cells,
                           const uint32_t* __restrict__ ix, const uint32_t* __restrict__ iy, const uint32_t* __restrict__ iz,
                           const uint32_t* __restrict__ off_in, uint32_t nrows, Fr* __restrict__ ox,
                           Fr* __restrict__ oy, Fr* __restrict__ oz) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t in0 = off_in[r], len = off_in[r + 1] - in0;
    Point3 res;
    if (LEVEL0) {
        Fr x1 = fr_zero(), y1 = fr_one(), x2 = fr_zero(), y2 = fr_one();
        if (len) {
            uint32_t i0 = cells[in0], i1 = cells[in0 + 1];
            x1 = fr_load(points_xy + 2ull * i0); y1 = fr_load(points_xy + 2ull * i0 + 1);
            if (i1 != PAD_IDX) { x2 = fr_load(points_xy + 2ull * i1); y2 = fr_load(points_xy + 2ull * i1 + 1); }
        }
        res = aff_add(x1, y1, x2, y2);
    } else {
        Point9 P = pt9_identity(), Q = pt9_identity();   // the level buffers hold the 9 x 29 form
        if (len) {
            P = pt9_load_raw9(ix, iy, iz, in0);
            Q = pt9_load_raw9(ix, iy, iz, (uint64_t)in0 + 1);
        }
        const Point9 R = proj_add9(P, Q);
        res.x = fr9_to(R.x); res.y = fr9_to(R.y); res.z = fr9_to(R.z);
    }
    fr_store(ox + r, res.x);
    fr_store(oy + r, res.y);
    fr_store(oz + r, res.z);
}

// ---------------------------------------------------------------------------------------------
// The late levels in ONE launch (levels L0 .. x_logsize - 1), thread = bucket row.
//
// With uniformly distributed digits a row holds ~2^(x - d) points, so at level L0 = x - d - 2 it is down to ~4 cells; the flat
// kernels then run 10+ more launches of a few thousand additions each, 12-14 us per launch of pure latency.  Here a thread takes
// its whole row through the remaining levels in registers:
//   * the pairwise tree over its (at most 16) cells, exactly the association of the flat levels (cells 2i, 2i + 1 of every level;
//     a missing right operand is the level's pad cell = the identity (0, 1, 1), pushforward.rs:380-381 / vecvec.rs:579-594);
//   * then one "add the identity" per remaining level.  The projective formulas with (0, 1, 1) as the second operand collapse to
//         (x, y, z) + (0, 1, 1) = (x z^3, y z^3, z^4)
//     (A = 0, B = y, zz = z, s = x, t = y, X = x z, Y = y z, dxy = 0, m = q = z^2 in twisted_edwards_ops.rs:36-65): the SAME field
//     elements as the 12-multiplication evaluation, with 2 squarings and 3 products.
// Rows longer than 16 cells at L0 (skewed digits: e.g. the top window of 252-bit scalars has 16 buckets for all 2^20 points) go to
// dedicated waves of the same launch: 64 lanes on one row, the tree in registers with the sums moving through the cross-lane
// network (rows of more than 128 cells first run whole levels through the level buffers, like the flat kernels).
// Measured at config B: 96 us for what took the flat launches ~130 us (a lone wave needs ~0.8 us per dependent multiplication
// here); on one rank's share of an 8-way sharded run, whose step is a chain of small launches, 0.716 -> 0.671 ms per step.
__device__ __forceinline__ Point9 add_identity9(const Point9& p) {
    // inputs: an addition's outputs or loaded cells (L 2^29, S <= 32)
    const Fr9 z2 = fr9_sqr(p.z);                       // S 15.5
    const Fr9 z3 = fr9_mul(z2, p.z);                   // S 8
    Point9 r;
    r.x = fr9_mul(p.x, z3);                            // S 4.7
    r.y = fr9_mul(p.y, z3);
    r.z = fr9_sqr(z2);                                 // S 4.4
    return r;
}
// 16 + 16 + 4 bytes with the device-coherent bit (another lane of this wave wrote / will read them through memory)
__device__ __forceinline__ Fr9 fr9_load_raw9_coh(const uint32_t* __restrict__ col, uint64_t i) {
    const uint32_t* p = col + 9 * i;
    gm_u4_t q0, q1;
    uint32_t w;
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\tglobal_load_dword %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(w) : "v"(p) : "memory");
    Fr9 r;
    r.l[0] = q0.x; r.l[1] = q0.y; r.l[2] = q0.z; r.l[3] = q0.w; r.l[4] = q1.x; r.l[5] = q1.y; r.l[6] = q1.z; r.l[7] = q1.w; r.l[8] = w;
    return r;
}
__device__ __forceinline__ void fr9_store_raw9_coh(uint32_t* __restrict__ col, uint64_t i, const Fr9& v) {
    uint32_t* p = col + 9 * i;
    const gm_u4_t q0 = {v.l[0], v.l[1], v.l[2], v.l[3]}, q1 = {v.l[4], v.l[5], v.l[6], v.l[7]};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\tglobal_store_dword %0, %3, off offset:32 sc1"
                 :: "v"(p), "v"(q0), "v"(q1), "v"(v.l[8]) : "memory");
}
__device__ __forceinline__ Point9 pt9_load_raw9_coh(const uint32_t* x, const uint32_t* y, const uint32_t* z, uint64_t i) {
    Point9 p;
    p.x = fr9_load_raw9_coh(x, i); p.y = fr9_load_raw9_coh(y, i); p.z = fr9_load_raw9_coh(z, i);   // rare path (rows > 128 cells): kept simple
    return p;
}
struct LvlBufs {
    uint32_t* c[2][3];   // the two level buffers (x, y, z columns of 9-word cells)
};

// lane q of the wave takes the Point9 of lane `from` (27 dwords through the cross-lane network)
__device__ __forceinline__ Point9 pt9_shfl(const Point9& v, int from) {
    Point9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.l[i] = __shfl(v.x.l[i], from, 64);
        r.y.l[i] = __shfl(v.y.l[i], from, 64);
        r.z.l[i] = __shfl(v.z.l[i], from, 64);
    }
    return r;
}

#define GM_TAIL_FAT_WAVES 64
// grid: ceil(nrows / 64) "thin" blocks (thread = row; rows of at most 16 cells), then GM_TAIL_FAT_WAVES blocks that look for the
// longer rows (one row per group of GM_TAIL_FAT_WAVES consecutive rows each) and take each of them with all 64 lanes.
// Code size matters here: an addition is ~2 700 instructions (22 KB) and a lone wave that runs straight-line code fetches every
// line of it from L2 (the first version, 13 inlined additions, took 620 us for 50 us of arithmetic).  Both paths are therefore
// LOOPS around ONE proj_add9 and ONE add_identity9 site; which operands a step takes is decided by moves around them.
__global__ void __launch_bounds__(64) k_add_tail(LvlBufs lb, const uint32_t* __restrict__ off_all, uint32_t nrows, uint32_t L0,
                                                 uint32_t x_log, uint32_t n_thin_blocks, Fr* __restrict__ ox, Fr* __restrict__ oy,
                                                 Fr* __restrict__ oz) {
    const uint32_t lane = threadIdx.x;
    const uint32_t stride = nrows + 1;
    const uint32_t* off0 = off_all + (uint64_t)L0 * stride;
    const int src = (int)((L0 - 1) & 1u);                           // level l reads buffer (l - 1) & 1 and writes l & 1
    const uint32_t nlev = x_log - L0;                               // >= 4 (host)
    if (blockIdx.x < n_thin_blocks) {
        const uint32_t r = lane * n_thin_blocks + blockIdx.x;
        if (r >= nrows) return;
        const uint32_t in0 = off0[r];
        const uint32_t c = off0[r + 1] - in0;                       // even: the rows are padded (k_add_level writes the pad cells)
        if (c > 16) return;                                         // a long row: one of the fat waves has it
        const uint32_t* ix = lb.c[src][0];
        const uint32_t* iy = lb.c[src][1];
        const uint32_t* iz = lb.c[src][2];
        // The 16-leaf tree over the row's cells as 15 steps through one addition site (a missing operand = the level's pad cell, the
        // identity; steps without a left operand cost nothing), then one add-the-identity step per remaining level.  A binary
        // counter with one pending sum per tree level (U, V, W):
        //   step   0      1      2        3      4      5      6         7      8      9        10     11     12     13     14
        //   T =   c0+c1  c2+c3  U+T      c4+c5  c6+c7  U+T    V+T       c8+c9  c10+11 U+T      c12+13 c14+15 U+T    V+T    W+T
        //   keep  U             V        U                    W         U             V        U
        Point9 T = pt9_identity(), U = T, V = T, W = T;
        bool hasT = false, hasU = false, hasV = false, hasW = false;
#pragma unroll 1
        for (uint32_t st = 0; st < nlev + 11; st++) {
            Point9 A = T, B = T;
            bool pa = hasT, pb = false;
            const bool leaf_step = st == 0 || st == 1 || st == 3 || st == 4 || st == 7 || st == 8 || st == 10 || st == 11;
            if (leaf_step) {
                const uint32_t leaf = st == 0 ? 0u : st == 1 ? 2u : st == 3 ? 4u : st == 4 ? 6u : st == 7 ? 8u : st == 8 ? 10u : st == 10 ? 12u : 14u;
                pa = pb = leaf < c;
                if (pa) {
                    A = pt9_load_raw9(ix, iy, iz, (uint64_t)in0 + leaf);
                    B = pt9_load_raw9(ix, iy, iz, (uint64_t)in0 + leaf + 1);
                }
            } else if (st == 2 || st == 5 || st == 9 || st == 12) {
                A = U; pa = hasU; pb = hasT;
            } else if (st == 6 || st == 13) {
                A = V; pa = hasV; pb = hasT;
            } else if (st == 14) {
                A = W; pa = hasW; pb = hasT;
            }
            Point9 S = pt9_identity();
            if (__any(pa && pb)) {
                const Point9 F = proj_add9(A, B);
                if (pa && pb) S = F;
            }
            if (__any(pa && !pb)) {
                const Point9 I = add_identity9(A);
                if (pa && !pb) S = I;
            }
            T = S;
            hasT = pa;
            if (st == 0 || st == 3 || st == 7 || st == 10) { U = T; hasU = hasT; }
            if (st == 2 || st == 9) { V = T; hasV = hasT; }
            if (st == 6) { W = T; hasW = hasT; }
        }
        // c == 0: an empty row stays the pad, (0, 1, 1) at every level
        fr_store(ox + r, fr9_to(T.x));
        fr_store(oy + r, fr9_to(T.y));
        fr_store(oz + r, fr9_to(T.z));
        return;
    }
    // ---- a fat wave: one row of every group of F = 64 consecutive rows; every long one among them with all 64 lanes.  WHICH row of
    // group hi is mixed with hi: with "row % F == w" the 32 rows (window y, digit 0) of an all-zero scalar vector -- rows 256 y -- all
    // fell to fat wave 0, which then added 32 rows of 1 024 cells one after the other (4.6 ms for a 3 ms MSM step; uniform scalars
    // have no long rows at all).  Rows y 2^d + const now spread over min(y_size, 64) waves for every d.
    const uint32_t w = blockIdx.x - n_thin_blocks;
    for (uint32_t hb = 0; hb * GM_TAIL_FAT_WAVES < nrows; hb += 64u) {
        const uint32_t hi = hb + lane;
        const uint32_t rr = hi * GM_TAIL_FAT_WAVES + ((w ^ hi ^ (hi >> 6) ^ (hi >> 12)) & (GM_TAIL_FAT_WAVES - 1u));
        const uint32_t cc = rr < nrows ? off0[rr + 1] - off0[rr] : 0u;
        uint64_t fat = __ballot(cc > 16);
        while (fat) {
            const int fl = __ffsll((unsigned long long)fat) - 1;
            fat &= fat - 1;
            const uint32_t rf = __shfl(rr, fl, 64);
            int cur = src;
            uint32_t lvl = L0;
            uint32_t n_in = __shfl(cc, fl, 64);
            // (i) rows of more than 128 cells: whole levels through the level buffers, like the flat kernels
            for (; n_in > 128; lvl++) {
                const uint32_t* offi = off_all + (uint64_t)lvl * stride;
                const uint32_t* offo = offi + stride;
                const uint32_t i0 = offi[rf], o0 = offo[rf], n_out = offo[rf + 1] - o0;
                const bool first = lvl == L0;
                for (uint32_t p = lane; p < n_out; p += 64) {
                    Point9 v;
                    if (p < (n_in >> 1)) {
                        const uint64_t a = (uint64_t)i0 + 2 * p;
                        const Point9 A = first ? pt9_load_raw9(lb.c[cur][0], lb.c[cur][1], lb.c[cur][2], a)
                                               : pt9_load_raw9_coh(lb.c[cur][0], lb.c[cur][1], lb.c[cur][2], a);
                        const Point9 B = first ? pt9_load_raw9(lb.c[cur][0], lb.c[cur][1], lb.c[cur][2], a + 1)
                                               : pt9_load_raw9_coh(lb.c[cur][0], lb.c[cur][1], lb.c[cur][2], a + 1);
                        v = proj_add9(A, B);
                    } else {
                        v = pt9_identity();
                    }
                    fr9_store_raw9_coh(lb.c[cur ^ 1][0], (uint64_t)o0 + p, v.x);
                    fr9_store_raw9_coh(lb.c[cur ^ 1][1], (uint64_t)o0 + p, v.y);
                    fr9_store_raw9_coh(lb.c[cur ^ 1][2], (uint64_t)o0 + p, v.z);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left before its lanes read them back
                __builtin_amdgcn_wave_barrier();
                n_in = n_out;
                cur ^= 1;
            }
            // (ii) at most 64 pairs: one per lane from memory, then the tree above them in registers (lane q takes the sums of lanes
            // 2 q and 2 q + 1 through the cross-lane network; a missing right neighbour is the level's pad cell, the identity), then
            // the identity once per remaining level -- all through one addition site
            uint32_t n = n_in;                      // operands of the coming step: cells in memory (first step), sums in lanes after
            Point9 v = pt9_identity();
            bool from_mem = true;
            const uint32_t i0 = off_all[(uint64_t)lvl * stride + rf];
            const bool first = lvl == L0;
#pragma unroll 1
            for (; lvl < x_log; lvl++) {
                Point9 A, B;
                if (from_mem) {
                    A = B = pt9_identity();
                    if (2 * lane + 1 < n) {
                        const uint64_t a = (uint64_t)i0 + 2 * lane;
                        A = first ? pt9_load_raw9(lb.c[cur][0], lb.c[cur][1], lb.c[cur][2], a) : pt9_load_raw9_coh(lb.c[cur][0], lb.c[cur][1], lb.c[cur][2], a);
                        B = first ? pt9_load_raw9(lb.c[cur][0], lb.c[cur][1], lb.c[cur][2], a + 1)
                                  : pt9_load_raw9_coh(lb.c[cur][0], lb.c[cur][1], lb.c[cur][2], a + 1);
                    }
                } else {
                    A = pt9_shfl(v, (int)((2 * lane) & 63));
                    B = pt9_shfl(v, (int)((2 * lane + 1) & 63));
                }
                const uint32_t m = (n + 1) >> 1;                     // sums after this step
                const bool pa = lane < m, pb = 2 * lane + 1 < n;
                Point9 S = pt9_identity();
                if (__any(pa && pb)) {
                    const Point9 F = proj_add9(A, B);
                    if (pa && pb) S = F;
                }
                if (__any(pa && !pb)) {
                    const Point9 I = add_identity9(A);
                    if (pa && !pb) S = I;
                }
                v = S;
                n = m;
                from_mem = false;
            }
            if (lane == 0) {
                fr_store(ox + rf, fr9_to(v.x));
                fr_store(oy + rf, fr9_to(v.y));
                fr_store(oz + rf, fr9_to(v.z));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Bucket reduction, one workgroup per window.  Restates pippenger_ending.rs:46-58 (two splits at the
// top two digit bits) + triangle_add.rs:101-158 (layers 0..d-2) + `last_step` (triangle_add.rs:88-99).
// Window w owns nd = 2^d bucket sums; all intermediate points live in LDS.
//
// Layer l works on arrays of length n_l = nd >> (2 + l) holding points  a,b,c,d  and  l  carried pairs:
//   L1..L3:  P0 = a + c, P1 = b + d, P2 = c + d, S_k = (pair_k.0 + pair_k.1)        (l + 3 adds / index)
//   split at the next digit bit (HI(y_logsize), bundle 3): every output point is cut in lower / upper half,
//   so the next layer sees  a'=P0.lo b'=P0.hi c'=P1.lo d'=P1.hi  and pairs (P2.lo,P2.hi),(S_0.lo,S_0.hi),...
// The last layer is not split; its l + 3 = d + 1 points (arrays of length 1) are the window points.
template <bool USE_LDS>
__global__ void k_triangle(const Fr* __restrict__ bx, const Fr* __restrict__ by, const Fr* __restrict__ bz,
                           uint32_t d_log, uint32_t nwin, Fr* __restrict__ out_cols /* 3*(d+1) cols x nwin */,
                           Point3* __restrict__ scratch /* 2*nd points per window when !USE_LDS */) {
    extern __shared__ unsigned char smem[];
    const uint32_t nd = 1u << d_log;
    const uint32_t w = blockIdx.x;
    // nd points, layout described below (d_log = 9, 10 do not fit the default 64 KiB of LDS: global scratch)
    Point3* cur = USE_LDS ? reinterpret_cast<Point3*>(smem) : scratch + (uint64_t)w * 2 * nd;
    Point3* nxt = cur + nd;
    // load: cur[q * n0 + i], q in {a,b,c,d} = (top bit, next bit) = 00,01,10,11; i = low d-2 bits
    for (uint32_t t = threadIdx.x; t < nd; t += blockDim.x) {
        Point3 p;
        p.x = fr_load(bx + (uint64_t)w * nd + t);
        p.y = fr_load(by + (uint64_t)w * nd + t);
        p.z = fr_load(bz + (uint64_t)w * nd + t);
        cur[t] = p;  // digit t = q * n0 + i already (q = top two bits)
    }
    __syncthreads();
    const uint32_t num_layers = d_log - 2;
    uint32_t n = nd >> 2;  // array length at this layer
    for (uint32_t l = 0; l <= num_layers; l++) {
        // inputs: 4 + 2l arrays of length n, array k at cur[k * n + i]
        const uint32_t nadds = l + 3;
        for (uint32_t t = threadIdx.x; t < nadds * n; t += blockDim.x) {
            const uint32_t k = t / n, i = t % n;
            uint32_t ia, ib;
            if (k == 0) { ia = 0; ib = 2; }            // a + c
            else if (k == 1) { ia = 1; ib = 3; }       // b + d
            else if (k == 2) { ia = 2; ib = 3; }       // c + d
            else { ia = 4 + 2 * (k - 3); ib = ia + 1; }  // carried pair
            Point3 r = proj_add(cur[ia * n + i], cur[ib * n + i]);
            if (l < num_layers) {
                // split on the top bit of i: halves of length n/2; output point k -> arrays 2k (lo), 2k+1 (hi)
                const uint32_t h = n >> 1;
                const uint32_t hi = i / h, ii = i % h;
                nxt[(2 * k + hi) * h + ii] = r;
            } else {
                nxt[k] = r;  // n == 1
            }
        }
        __syncthreads();
        Point3* tmp = cur; cur = nxt; nxt = tmp;
        n >>= 1;
    }
    // cur[k], k = 0..d : window points; column order X0,Y0,Z0,X1,...
    for (uint32_t k = threadIdx.x; k <= d_log; k += blockDim.x) {
        fr_store(out_cols + (uint64_t)(3 * k + 0) * nwin + w, cur[k].x);
        fr_store(out_cols + (uint64_t)(3 * k + 1) * nwin + w, cur[k].y);
        fr_store(out_cols + (uint64_t)(3 * k + 2) * nwin + w, cur[k].z);
    }
}

}  // namespace gm

using namespace gm;

namespace gm {
int32_t launch_offsets_next(const uint32_t* off_in, uint32_t* off_out, uint32_t nrows, hipStream_t s) {
    hipLaunchKernelGGL((k_offsets_scan<1>), dim3(1), dim3(1024), 0, s, off_in, off_out, nrows);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
// off_all: nlevels tables of nrows + 1 entries; table 0 = off0, table l + 1 = the row layout after one more fold
int32_t launch_offsets_all_from_off(const uint32_t* off0, uint32_t* off_all, uint32_t nrows, uint32_t nlevels, hipStream_t s) {
    hipLaunchKernelGGL((k_offsets_levels_par<true>), dim3(nlevels), dim3(1024), 0, s, off0, off_all, nrows);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int32_t launch_offsets_from_len(const uint32_t* len, uint32_t* off, uint32_t nrows, hipStream_t s) {
    hipLaunchKernelGGL((k_offsets_scan<0>), dim3(1), dim3(1024), 0, s, len, off, nrows);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
}  // namespace gm

template <typename T>
static int32_t plan_alloc(gm_msm_plan* p, T** ptr, uint64_t count) {
    size_t b = (size_t)count * sizeof(T);
    if (b == 0) b = 16;
    GM_HIP(dev_alloc((void**)ptr, b));
    p->bytes += b;
    return GM_OK;
}

extern "C" int32_t gm_msm_plan_create(uint32_t x_logsize, uint32_t d_logsize, uint32_t y_size, uint32_t y_begin,
                                      uint32_t y_end, gm_msm_plan** out) {
    GM_REQUIRE(out != nullptr, "null out");
    GM_REQUIRE(x_logsize >= 1 && x_logsize <= 26, "x_logsize %u out of range [1,26]", x_logsize);
    GM_REQUIRE(d_logsize >= 2 && d_logsize <= 10, "d_logsize %u out of range [2,10] (examples/pippenger.rs:24)",
               d_logsize);
    GM_REQUIRE(y_size >= 1 && (uint64_t)y_size * d_logsize <= 256,
               "y_size*d_logsize = %u > 256 (pushforward.rs:358 would index past the scalar bits)",
               y_size * d_logsize);
    GM_REQUIRE(y_begin < y_end && y_end <= y_size, "bad window range [%u,%u) of %u", y_begin, y_end, y_size);
    gm_msm_plan* p = new gm_msm_plan();
    p->x_log = x_logsize; p->d_log = d_logsize; p->y_size = y_size; p->y0 = y_begin; p->y1 = y_end;
    p->nwin = y_end - y_begin; p->nd = 1u << d_logsize; p->nrows = p->nwin << d_logsize;
    p->N = 1ull << x_logsize;
    p->chunk = CHUNK;
    p->nchunks = (uint32_t)((p->N + p->chunk - 1) / p->chunk);
    const uint64_t cells_in = (uint64_t)p->nwin * p->N + p->nrows;  // + one pad per row at most
    GM_REQUIRE(cells_in < 0xfffffff0ull, "too many cells for 32-bit offsets");
    p->cap0 = cells_in / 2 + p->nrows + 2;
    p->cap1 = p->cap0 / 2 + p->nrows + 2;
    {   // block-row tables of the flat levels 0 .. x_logsize - 2 (the last level is the row-owned tail kernel)
        uint64_t cells = p->cap0;
        uint32_t tot = 0;
        p->blk_nlev = p->x_log >= 2 ? p->x_log - 1 : 0;
        if (p->blk_nlev > 32) p->blk_nlev = 32;
        for (uint32_t l = 0; l < p->blk_nlev; l++) {
            p->blk_first[l] = tot;
            tot += (uint32_t)((cells + 127) / 128) + 1;
            cells = cells / 2 + p->nrows + 2;
        }
        p->blk_first[p->blk_nlev] = tot;
    }
    int32_t rc = GM_OK;
#define ALLOC(ptr, n) if ((rc = plan_alloc(p, &(ptr), (n))) != GM_OK) { gm_msm_plan_destroy(p); return rc; }
    ALLOC(p->digits, (uint64_t)p->nwin * p->N);
    ALLOC(p->counter, (uint64_t)p->nwin * p->N);
    ALLOC(p->hist, (uint64_t)p->nwin * p->nchunks * p->nd);
    ALLOC(p->row_len, p->nrows);
    // off[0] holds the offsets of ALL levels: level l at off[0] + l * (nrows + 1); level 0 = the image rows
    ALLOC(p->off[0], (uint64_t)(x_logsize + 1) * (p->nrows + 1));
    ALLOC(p->blk_row, (uint64_t)p->blk_first[p->blk_nlev] + 1);
    ALLOC(p->cells, cells_in + 2);
    for (int c = 0; c < 3; c++) {
        ALLOC(p->lvl[0][c], 9 * p->cap0 + 4);
        ALLOC(p->lvl[1][c], 9 * p->cap1 + 4);
        ALLOC(p->bsum[c], p->nrows);
    }
    ALLOC(p->win_pts, (uint64_t)3 * (d_logsize + 1) * p->nwin);
    if (2ull * p->nd * sizeof(Point3) > 48 * 1024) ALLOC(p->tri_scratch, (uint64_t)p->nwin * 2 * p->nd * 3);
#undef ALLOC
    *out = p;
    return GM_OK;
}

extern "C" int32_t gm_msm_plan_destroy(gm_msm_plan* p) {
    if (!p) return GM_OK;
    dev_free(p->digits); dev_free(p->counter); dev_free(p->hist); dev_free(p->row_len);
    dev_free(p->off[0]); dev_free(p->off[1]); dev_free(p->off[2]); dev_free(p->cells); dev_free(p->blk_row);
    for (int c = 0; c < 3; c++) { dev_free(p->lvl[0][c]); dev_free(p->lvl[1][c]); dev_free(p->bsum[c]); }
    dev_free(p->win_pts);
    dev_free(p->tri_scratch);
    for (int i = 0; i <= GM_MSM_NSTAGE; i++) if (p->ev[i]) (void)hipEventDestroy(p->ev[i]);
    delete p;
    return GM_OK;
}

extern "C" int32_t gm_msm_profile(gm_msm_plan* p, int32_t mode) {
    GM_REQUIRE(p && mode >= 0 && mode <= 2, "bad argument");
    if (mode && !p->ev[0])
        for (int i = 0; i <= GM_MSM_NSTAGE; i++) GM_HIP(hipEventCreate(&p->ev[i]));
    p->prof_mode = mode;
    return GM_OK;
}

// ms per stage of the LAST run: 0 digits, 1 histogram, 2 chunk scan + offsets, 3 scatter, 4 level-0 add,
// 5 levels >= 1, 6 triangle; -1 where not recorded.  Synchronises on the recorded events.
extern "C" int32_t gm_msm_profile_read(gm_msm_plan* p, float* h_ms, int32_t n) {
    GM_REQUIRE(p && h_ms && n >= GM_MSM_NSTAGE, "bad argument");
    for (int i = 0; i < GM_MSM_NSTAGE; i++) {
        h_ms[i] = -1.f;
        if (p->ev_rec[i] && p->ev_rec[i + 1]) {
            GM_HIP(hipEventSynchronize(p->ev[i + 1]));
            GM_HIP(hipEventElapsedTime(&h_ms[i], p->ev[i], p->ev[i + 1]));
        }
    }
    return GM_OK;
}

extern "C" size_t gm_msm_plan_workspace_bytes(const gm_msm_plan* p) { return p ? p->bytes : 0; }

// how the plan's last gm_msm_run was launched: *fused01 = 1 when bintree levels 0 and 1 ran as one kernel (k_add_level01: stage
// "add_level0" of gm_msm_profile_read then covers both levels and "add_levels_ge1" the levels from 2 on)
extern "C" int32_t gm_msm_run_info(const gm_msm_plan* p, int32_t* fused01) {
    GM_REQUIRE(p && fused01, "null argument");
    *fused01 = p->fused01 ? 1 : 0;
    return GM_OK;
}

extern "C" int32_t gm_msm_level_cells(const gm_msm_plan* p, uint64_t* h_cells, uint32_t n, void* stream) {
    GM_REQUIRE(p && h_cells && n >= p->x_log + 1, "need x_logsize + 1 counts");
    std::vector<uint32_t> v(p->x_log + 1, 0u);
    // table l of off[0] is the layout level l reads; its last entry is the layout's cell count (levels 0 .. x_log - 1 exist)
    for (uint32_t l = 0; l < p->x_log; l++)
        GM_HIP(hipMemcpyAsync(&v[l], p->off[0] + (uint64_t)l * (p->nrows + 1) + p->nrows, 4, hipMemcpyDeviceToHost, as_stream(stream)));
    GM_HIP(hipStreamSynchronize(as_stream(stream)));
    for (uint32_t l = 0; l <= p->x_log; l++) h_cells[l] = v[l];
    return GM_OK;
}

extern "C" int32_t gm_msm_run(gm_msm_plan* p, const uint64_t* d_points_xy, const uint64_t* d_scalars, void* stream) {
    GM_REQUIRE(p && d_points_xy && d_scalars, "null argument");
    hipStream_t s = as_stream(stream);
    const uint64_t N = p->N;
    const uint32_t nrows = p->nrows, nd = p->nd;
    const Fr* pts = reinterpret_cast<const Fr*>(d_points_xy);
    for (int i = 0; i <= GM_MSM_NSTAGE; i++) p->ev_rec[i] = false;
#define STAGE_MARK(i)                                                                                   \
    do {                                                                                                \
        if (p->prof_mode == 2 || (p->prof_mode == 1 && ((i) == 4 || (i) == 5))) {                        \
            GM_HIP(hipEventRecord(p->ev[i], s));                                                        \
            p->ev_rec[i] = true;                                                                        \
        }                                                                                               \
    } while (0)

    STAGE_MARK(0);
    // 1. digits
    hipLaunchKernelGGL(k_digits, dim3(ceil_div(N, 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t*>(d_scalars), p->digits, N, p->d_log, p->y0, p->nwin);
    GM_LAUNCH_CHECK();
    STAGE_MARK(1);
    // 2. histogram per (window, chunk), scan over chunks, row offsets
    const uint64_t ntasks = (uint64_t)p->nwin * p->nchunks;
    const uint32_t waves = 4;
    hipLaunchKernelGGL(k_hist, dim3(ceil_div(ntasks, waves)), dim3(64 * waves), waves * nd * sizeof(uint32_t), s,
                       p->digits, p->hist, N, nd, p->nchunks, p->chunk, ntasks);
    GM_LAUNCH_CHECK();
    STAGE_MARK(2);
    hipLaunchKernelGGL(k_scan_chunks, dim3(p->nwin * (nd / (nd < 64 ? nd : 64))), dim3(1024), 0, s, p->hist, p->row_len, nd,
                       p->nchunks, nrows);
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_offsets_levels_par<false>), dim3(p->x_log), dim3(1024), 0, s, p->row_len, p->off[0], nrows);
    GM_LAUNCH_CHECK();
    if (p->blk_nlev) {
        BlockRowsArgs ba;
        ba.nlev = p->blk_nlev;
        for (uint32_t l = 0; l <= p->blk_nlev; l++) ba.first[l] = p->blk_first[l];
        hipLaunchKernelGGL(k_block_rows, dim3(ceil_div(p->blk_first[p->blk_nlev], 256)), dim3(256), 0, s, ba, p->off[0], nrows, p->blk_row);
        GM_LAUNCH_CHECK();
    }
    STAGE_MARK(3);
    // 3. stable scatter
    if (p->chunk == 1024 && p->nchunks % 4 == 0 && nd <= 1024) {
        const size_t lds = ((size_t)7 * nd + 1 + 4096) * sizeof(uint32_t) + 4096 * sizeof(uint16_t);
        hipLaunchKernelGGL(k_rank_scatter_tile, dim3((unsigned)(ntasks / 4)), dim3(256), lds, s, p->digits, p->hist, p->off[0], p->counter,
                           p->cells, N, p->d_log, p->nchunks);
    } else {
        hipLaunchKernelGGL(k_rank_scatter, dim3(ceil_div(ntasks, waves)), dim3(64 * waves),
                           waves * nd * sizeof(uint32_t), s, p->digits, p->hist, p->off[0], p->counter, p->cells, N,
                           p->d_log, p->nchunks, p->chunk, ntasks);
    }
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pad_cells, dim3(ceil_div(nrows, 256)), dim3(256), 0, s, p->row_len, p->off[0], p->cells,
                       nrows);
    GM_LAUNCH_CHECK();
    // 4. bucket sums: x_log levels of pairwise adds.  Level l reads the layout off_all[l] and writes off_all[l + 1];
    //    flat launches over all cells; the last level (every row 0 or 2 cells) by one thread per row (k_add_last).
    uint64_t cap_out = p->cap0;
    const uint32_t stride = nrows + 1;
    if (p->x_log == 1) {
        hipLaunchKernelGGL((k_add_last<true>), dim3(ceil_div(nrows, 128)), dim3(128), 0, s, pts, p->cells,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, p->off[0], nrows, p->bsum[0],
                           p->bsum[1], p->bsum[2]);
        GM_LAUNCH_CHECK();
    } else {
        // levels tail_L0 .. x_log - 1 by the row-owned tail kernel (k_add_tail) when at least three levels are left for it
        static const bool no_tail = [] { const char* e = getenv("GM_MSM_NO_TAIL"); return e && e[0] == '1'; }();
        uint32_t tail_L0 = 0;
        if (!no_tail && p->x_log >= 5) {   // the 16-leaf tree of the tail kernel spans four levels
            const int want = (int)p->x_log - (int)p->d_log - 2;
            tail_L0 = (uint32_t)(want < 1 ? 1 : want);
            if (tail_L0 > p->x_log - 4) tail_L0 = p->x_log - 4;
        }
        const uint32_t tail_level = tail_L0 ? tail_L0 : p->x_log - 1;
        // levels 0 and 1 fused (k_add_level01) whenever level 1 is a flat launch of its own; GM_MSM_FUSE01=0: the two launches (A/B)
        static const bool fuse01_on = [] { const char* e = getenv("GM_MSM_FUSE01"); return !(e && e[0] == '0'); }();
        const bool fuse01 = fuse01_on && tail_level >= 2 && p->blk_nlev >= 2;
        p->fused01 = fuse01;
        STAGE_MARK(4);
        if (fuse01) {
            const uint64_t cells1 = cap_out / 2 + nrows + 2;
            hipLaunchKernelGGL(k_add_level01, dim3(ceil_div(cells1, 128)), dim3(128), 0, s, pts, p->cells, p->off[0], p->off[0] + stride,
                               p->off[0] + 2ull * stride, nrows, p->lvl[1][0], p->lvl[1][1], p->lvl[1][2], p->blk_row + p->blk_first[1]);
        } else {
            hipLaunchKernelGGL(k_add_level0, dim3(ceil_div(cap_out, 128)), dim3(128), 0, s, pts, p->cells, p->off[0],
                               p->off[0] + stride, nrows, p->lvl[0][0], p->lvl[0][1], p->lvl[0][2], p->blk_row + p->blk_first[0]);
        }
        GM_LAUNCH_CHECK();
        STAGE_MARK(5);
        // (A first row-owned kernel for the trailing levels, one workgroup per row, was measured and dropped in round 1: its lanes
        // idled.  k_add_tail gives every LANE a row and adds the identity with 5 multiplications instead of 12.)
        int cur_lvl = fuse01 ? 1 : 0;
        uint64_t cells_cur = fuse01 ? cap_out / 2 + nrows + 2 : cap_out;
        for (uint32_t level = fuse01 ? 2 : 1; level < tail_level; level++) {
            const uint64_t cells_next = cells_cur / 2 + nrows + 2;
            hipLaunchKernelGGL(k_add_level, dim3(ceil_div(cells_next, 128)), dim3(128), 0, s, p->lvl[cur_lvl][0],
                               p->lvl[cur_lvl][1], p->lvl[cur_lvl][2], p->off[0] + (uint64_t)level * stride,
                               p->off[0] + (uint64_t)(level + 1) * stride, nrows, p->lvl[cur_lvl ^ 1][0],
                               p->lvl[cur_lvl ^ 1][1], p->lvl[cur_lvl ^ 1][2], p->blk_row + p->blk_first[level]);
            GM_LAUNCH_CHECK();
            cur_lvl ^= 1;
            cells_cur = cells_next;
        }
        // the last level: every row is down to 0 or 2 cells (x_logsize halvings of at most 2^x_logsize cells): one thread per row
        // (a row-owned workgroup per row spent 86 us here with one busy lane in 64)
        if (tail_L0) {
            LvlBufs lb;
            // level l reads buffer (l - 1) & 1: that is cur_lvl here (level 0 wrote buffer 0; the fused first launch wrote level 1's buffer)
            for (int b = 0; b < 2; b++)
                for (int c = 0; c < 3; c++) lb.c[b][c] = p->lvl[b][c];
            const uint32_t n_thin = ceil_div(nrows, 64);
            hipLaunchKernelGGL(k_add_tail, dim3(n_thin + GM_TAIL_FAT_WAVES), dim3(64), 0, s, lb, p->off[0], nrows, tail_L0, p->x_log, n_thin,
                               p->bsum[0], p->bsum[1], p->bsum[2]);
        } else {
        hipLaunchKernelGGL((k_add_last<false>), dim3(ceil_div(nrows, 128)), dim3(128), 0, s, (const Fr*)nullptr, (const uint32_t*)nullptr,
                           p->lvl[cur_lvl][0], p->lvl[cur_lvl][1], p->lvl[cur_lvl][2], p->off[0] + (uint64_t)tail_level * stride, nrows,
                           p->bsum[0], p->bsum[1], p->bsum[2]);
        }
        GM_LAUNCH_CHECK();
    }
    STAGE_MARK(6);
    // 5. bucket reduction per window
    const size_t lds = 2 * (size_t)nd * sizeof(Point3);
    const dim3 tb(nd < 256 ? (nd < 64 ? 64 : nd) : 256);
    if (lds <= 48 * 1024) {
        hipLaunchKernelGGL((k_triangle<true>), dim3(p->nwin), tb, lds, s, p->bsum[0], p->bsum[1], p->bsum[2],
                           p->d_log, p->nwin, p->win_pts, (Point3*)nullptr);
    } else {
        hipLaunchKernelGGL((k_triangle<false>), dim3(p->nwin), tb, 0, s, p->bsum[0], p->bsum[1], p->bsum[2],
                           p->d_log, p->nwin, p->win_pts, reinterpret_cast<Point3*>(p->tri_scratch));
    }
    GM_LAUNCH_CHECK();
    STAGE_MARK(7);
#undef STAGE_MARK
    return GM_OK;
}

extern "C" int32_t gm_msm_bucket_sums(const gm_msm_plan* p, const uint64_t** d_x, const uint64_t** d_y,
                                      const uint64_t** d_z, uint64_t* n_rows_local) {
    GM_REQUIRE(p, "null plan");
    if (d_x) *d_x = reinterpret_cast<const uint64_t*>(p->bsum[0]);
    if (d_y) *d_y = reinterpret_cast<const uint64_t*>(p->bsum[1]);
    if (d_z) *d_z = reinterpret_cast<const uint64_t*>(p->bsum[2]);
    if (n_rows_local) *n_rows_local = p->nrows;
    return GM_OK;
}

extern "C" int32_t gm_msm_window_points(const gm_msm_plan* p, const uint64_t** d_cols, uint64_t* n_cols,
                                        uint64_t* col_len) {
    GM_REQUIRE(p, "null plan");
    if (d_cols) *d_cols = reinterpret_cast<const uint64_t*>(p->win_pts);
    if (n_cols) *n_cols = 3ull * (p->d_log + 1);
    if (col_len) *col_len = p->nwin;
    return GM_OK;
}

extern "C" int32_t gm_msm_digits(const gm_msm_plan* p, const uint16_t** d_digits, const uint32_t** d_counter,
                                 const uint32_t** d_row_len) {
    GM_REQUIRE(p, "null plan");
    if (d_digits) *d_digits = p->digits;
    if (d_counter) *d_counter = p->counter;
    if (d_row_len) *d_row_len = p->row_len;
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Fr-side data of the pushforward argument derived from the bucketing (PushForwardState, pushforward.rs:489-510 and
// second_phase :572-596): c / d as field elements, negated access counts, and the eq "pullbacks".
namespace gm {

__global__ void __launch_bounds__(256) k_u_to_fr(const uint16_t* __restrict__ dg, const uint32_t* __restrict__ ct,
                                                  uint64_t n, Fr* __restrict__ d_out, Fr* __restrict__ c_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(d_out + i, fr_from_u64(dg[i]));
    fr_store(c_out + i, fr_from_u64(ct[i]));
}

__global__ void __launch_bounds__(256) k_access_counts(const uint16_t* __restrict__ dg, const uint32_t* __restrict__ ct,
                                                        uint64_t n, uint32_t* __restrict__ cnt_d, uint32_t* __restrict__ cnt_c) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAdd(&cnt_d[dg[i]], 1u);
    atomicAdd(&cnt_c[ct[i]], 1u);
}

__global__ void __launch_bounds__(256) k_neg_count_to_fr(const uint32_t* __restrict__ cnt, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(out + i, fr_neg(fr_from_u64(cnt[i])));
}

// out[k] = -(number of entries of sorted[0..n) greater than k)
__global__ void __launch_bounds__(256) k_ac_c_from_sorted(const uint32_t* __restrict__ sorted, uint32_t n, uint64_t N, Fr* __restrict__ out) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    uint32_t lo = 0, hi = n;   // first index with sorted[i] > k
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sorted[mid] <= k) lo = mid + 1; else hi = mid;
    }
    fr_store(out + k, fr_neg(fr_from_u64(n - lo)));
}

__global__ void __launch_bounds__(256) k_pull(const uint16_t* __restrict__ dg, const uint32_t* __restrict__ ct, uint64_t n,
                                               const Fr* __restrict__ eq_d, const Fr* __restrict__ eq_c,
                                               Fr* __restrict__ d_pull, Fr* __restrict__ c_pull) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(d_pull + i, fr_load(eq_d + dg[i]));
    fr_store(c_pull + i, fr_load(eq_c + ct[i]));
}

int32_t launch_eq_sequence(const Fr& mult, const Fr* pt, uint32_t nvars, Fr* const* levels, hipStream_t s, const Fr* extra = nullptr,
                           uint32_t n_extra = 0, Fr* extra_dst = nullptr);

}  // namespace gm

// d_c, d_d: y_size * 2^x elements each ([y][x]); d_ac_c: 2^x_logsize, d_ac_d: 2^d_logsize elements
extern "C" int32_t gm_msm_phase1_polys(const gm_msm_plan* p, uint64_t* d_c, uint64_t* d_d, uint64_t* d_ac_c,
                                       uint64_t* d_ac_d, void* stream) {
    GM_REQUIRE(p && d_c && d_d && d_ac_c && d_ac_d, "null argument");
    hipStream_t s = as_stream(stream);
    const uint64_t n = (uint64_t)p->nwin * p->N;
    hipLaunchKernelGGL(k_u_to_fr, dim3(ceil_div(n, 256)), dim3(256), 0, s, p->digits, p->counter, n,
                       reinterpret_cast<Fr*>(d_d), reinterpret_cast<Fr*>(d_c));
    GM_LAUNCH_CHECK();
    // access counts (pushforward.rs:496-508) from the bucket populations instead of 2^25 contended atomics:
    //   ac_d[digit] = sum over windows of the population of bucket (window, digit)
    //   ac_c[k]     = number of buckets with more than k points (counter value k occurs once in each of them)
    //               = nrows - upper_bound(sorted populations, k): the host sorts the nrows populations (a few thousand),
    //                 the device does one binary search per k
    uint32_t* cnt = nullptr;
    const uint64_t nc = (uint64_t)p->nrows + p->nd;
    GM_HIP(dev_alloc((void**)&cnt, nc * 4));
    {
        std::vector<uint32_t> rl(p->nrows), h(nc, 0);
        GM_HIP(hipMemcpyAsync(rl.data(), p->row_len, (size_t)p->nrows * 4, hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
        for (uint32_t r = 0; r < p->nrows; r++) h[p->nrows + (r % p->nd)] += rl[r];
        std::sort(rl.begin(), rl.end());
        memcpy(h.data(), rl.data(), (size_t)p->nrows * 4);
        GM_HIP(hipMemcpyAsync(cnt, h.data(), nc * 4, hipMemcpyHostToDevice, s));
        GM_HIP(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(k_ac_c_from_sorted, dim3(ceil_div(p->N, 256)), dim3(256), 0, s, cnt, p->nrows, p->N, reinterpret_cast<Fr*>(d_ac_c));
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_neg_count_to_fr, dim3(ceil_div(p->nd, 256)), dim3(256), 0, s, cnt + p->nrows, (uint64_t)p->nd,
                       reinterpret_cast<Fr*>(d_ac_d));
    GM_LAUNCH_CHECK();
    GM_HIP(hipStreamSynchronize(s));
    dev_free(cnt);
    return GM_OK;
}

// h_r: y_logsize + d_logsize + x_logsize elements, layout [r_y | r_d | r_c] (pushforward.rs:574-580)
extern "C" int32_t gm_msm_second_phase(const gm_msm_plan* p, const uint64_t* h_r, uint32_t y_logsize, uint64_t* d_c_pull,
                                       uint64_t* d_d_pull, void* stream) {
    GM_REQUIRE(p && h_r && d_c_pull && d_d_pull, "null argument");
    hipStream_t s = as_stream(stream);
    const uint32_t dl = p->d_log, xl = p->x_log;
    std::vector<Fr> r(y_logsize + dl + xl);
    memcpy(r.data(), h_r, r.size() * sizeof(Fr));
    Fr* buf = nullptr;
    const uint64_t tot = (2ull << xl) + (2ull << dl);
    GM_HIP(dev_alloc((void**)&buf, tot * sizeof(Fr)));
    Fr* eq_c = buf;                       // 2^xl, scratch 2^xl after it
    Fr* eq_d = buf + (2ull << xl);        // 2^dl, scratch after it
    std::vector<Fr*> lv(xl + 1);
    for (uint32_t i = 0; i < xl; i++) lv[i] = eq_c + (1ull << xl) + ((1ull << i) - 1);
    lv[xl] = eq_c;
    int32_t rc = launch_eq_sequence(fr_one(), r.data() + y_logsize + dl, xl, lv.data(), s);
    if (rc == GM_OK) {
        lv.assign(dl + 1, nullptr);
        for (uint32_t i = 0; i < dl; i++) lv[i] = eq_d + (1ull << dl) + ((1ull << i) - 1);
        lv[dl] = eq_d;
        rc = launch_eq_sequence(fr_one(), r.data() + y_logsize, dl, lv.data(), s);
    }
    if (rc == GM_OK) {
        const uint64_t n = (uint64_t)p->nwin * p->N;
        hipLaunchKernelGGL(k_pull, dim3(ceil_div(n, 256)), dim3(256), 0, s, p->digits, p->counter, n, eq_d, eq_c,
                           reinterpret_cast<Fr*>(d_d_pull), reinterpret_cast<Fr*>(d_c_pull));
        if (hipGetLastError() != hipSuccess) rc = set_err(GM_ERR_HIP, "k_pull launch failed");
    }
    (void)hipStreamSynchronize(s);
    dev_free(buf);
    return rc;
}
