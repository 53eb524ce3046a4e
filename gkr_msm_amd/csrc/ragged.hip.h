// Ragged ("VecVec") row structure helpers shared by the MSM levels, the witness builders and the
// VecVec sumcheck: rows are stored back to back, every stored row has even length (the odd ones carry one
// explicit pad cell, /root/reference/src/cleanup/polys/vecvec.rs:181-186), off[r] is the first cell of row r.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace gm {

static constexpr uint32_t PAD_IDX = 0xffffffffu;

// largest r with off[r] <= j   (requires off[0] <= j < off[nrows]); empty rows are skipped naturally
__device__ __forceinline__ uint32_t find_row(const uint32_t* __restrict__ off, uint32_t nrows, uint32_t j) {
    uint32_t lo = 0, hi = nrows;  // invariant: off[lo] <= j < off[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// Row lookup through a coarse table: coarse[c] = row of cell c << GM_COARSE_SHIFT (clamped to the last cell), so the row of
// cell j lies in [coarse[j >> S], coarse[(j >> S) + 1]]: a search over the rows that intersect one 2^S-cell block.
#define GM_COARSE_SHIFT 8
__device__ __forceinline__ uint32_t find_row_coarse(const uint32_t* __restrict__ off, uint32_t nrows, const uint32_t* __restrict__ coarse,
                                                    uint32_t j) {
    const uint32_t c = j >> GM_COARSE_SHIFT;
    uint32_t lo = coarse[c], hi = coarse[c + 1] + 1;  // off[lo] <= j < off[hi]
    if (hi > nrows) hi = nrows;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// Block-cooperative row lookup for a block whose threads own consecutive cells [j0, j0 + blockDim.x): two lanes
// bracket the block's row range with a full binary search, every thread then searches only inside that bracket
// (usually one or two rows).  This replaces log2(nrows) dependent global loads per thread by ~1.
// Every thread of the block must call it (it contains a barrier); `valid` = this thread's j is in range.
__device__ __forceinline__ uint32_t find_row_block(const uint32_t* __restrict__ off, uint32_t nrows, uint32_t j, bool valid,
                                                   uint32_t total) {
    __shared__ uint32_t s_range[2];
    const uint32_t j0 = blockIdx.x * blockDim.x;
    if (threadIdx.x < 2 && j0 < total) {
        uint32_t jj = threadIdx.x == 0 ? j0 : j0 + blockDim.x - 1;
        if (jj >= total) jj = total - 1;
        s_range[threadIdx.x] = find_row(off, nrows, jj);
    }
    __syncthreads();
    if (!valid) return 0;
    uint32_t lo = s_range[0], hi = s_range[1] + 1;  // off[lo] <= j < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// Same idea for a block whose threads own the cells j_first .. j_last (block-uniform bounds, j_first <= j <= j_last for
// every valid thread), usable inside grid-stride loops: two barriers per call, every thread of the block must call it.
__device__ __forceinline__ uint32_t find_row_span(const uint32_t* __restrict__ off, uint32_t nrows, uint32_t j, bool valid,
                                                  uint32_t j_first, uint32_t j_last) {
    __shared__ uint32_t s_span[2];
    __syncthreads();  // the previous call's readers are done with s_span
    if (threadIdx.x < 2) s_span[threadIdx.x] = find_row(off, nrows, threadIdx.x == 0 ? j_first : j_last);
    __syncthreads();
    if (!valid) return 0;
    uint32_t lo = s_span[0], hi = s_span[1] + 1;  // off[lo] <= j < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

}  // namespace gm
