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

}  // namespace gm
