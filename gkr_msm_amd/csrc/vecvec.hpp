// gm_vv: k VecVec polynomials over one shared ragged row structure (host-side handle).
#pragma once
#include <memory>
#include <vector>

#include "internal.hpp"

struct gm_msm_plan;
struct gm_vv {
    uint32_t k = 0;
    uint32_t nrows = 0;        // stored rows (data.len() in the reference); rows past it are col_pad
    uint64_t total = 0;        // cells = sum of (even) stored row lengths
    uint32_t row_logsize = 0;  // low variables: index inside a row; the ones a sumcheck runs over
    uint32_t col_logsize = 0;  // high variables: row index
    uint32_t max_row_len = 0;  // EQPolyData::new needs it (vecvec.rs:86); the GLOBAL maximum when sharded
    bool sharded = false;      // this handle is one rank's slice of the rows (dense conversions then produce the slice only)
    uint32_t row_base = 0;     // sharded: global index of stored row 0 (this handle holds rows row_base .. row_base + nrows)
    std::shared_ptr<gm::DevBuf> off;                // u32[nrows + 1]
    // Every row layout this shape goes through (rows halve and are re-padded to even length with each split / sparse fold:
    // vecvec.rs:420-441, 579-594): level l at l * (nrows + 1) words, `off` = level off_level.  Present for the bucket image
    // (the MSM plan computes all x_logsize layouts in one launch); splits and sumcheck objects then take their layouts from
    // here instead of recomputing them per layer.
    std::shared_ptr<gm::DevBuf> off_levels;
    uint32_t off_level = 0, n_off_levels = 0;
    std::shared_ptr<std::vector<uint32_t>> level_totals;   // host copy: cells per level
    // coarse row tables of every level (find_row_coarse, ragged.hip.h): level l at word coarse_off[l], (cells >> 8) + 2 entries
    std::shared_ptr<gm::DevBuf> coarse;
    std::shared_ptr<std::vector<uint64_t>> coarse_off;
    void share_levels(const gm_vv* in, uint32_t level_shift) {
        off_levels = in->off_levels; n_off_levels = in->n_off_levels; level_totals = in->level_totals;
        coarse = in->coarse; coarse_off = in->coarse_off;
        off_level = in->off_level + level_shift;
    }
    std::vector<std::shared_ptr<gm::DevBuf>> cols;  // k arrays of `total` elements
    std::vector<gm::Fr> row_pad, col_pad;
    int32_t alloc_cols(uint32_t k_, uint64_t total_);
};

namespace gm {
int32_t vv_from_msm(const gm_msm_plan* p, const uint64_t* d_points_xy, uint32_t y_logsize, bool partial, gm_vv** out, void* stream);
int32_t vv_map(const SegPlan& sp, const gm_vv* in, gm_vv** out, hipStream_t s);
int32_t vv_map_split(const SegPlan& sp, const gm_vv* in, uint32_t bundle, gm_vv** out, hipStream_t s);
int32_t vv_map_split_to_dense(const SegPlan& sp, const gm_vv* in, uint32_t bundle, Fr* const* d_out, hipStream_t s);
}  // namespace gm
