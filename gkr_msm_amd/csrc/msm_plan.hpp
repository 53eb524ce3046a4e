// The MSM plan (workspaces of gm_msm_run); shared with the witness builders that start from the bucket image.
#pragma once
#include "common.hpp"
#include "fr.hip.h"

using gm::Fr;

#define GM_MSM_NSTAGE 7
struct gm_msm_plan {
    uint32_t x_log, d_log, y_size, y0, y1, nwin, nd, nrows, nchunks, chunk;
    uint64_t N;
    uint16_t* digits = nullptr;
    uint32_t* counter = nullptr;
    uint32_t* hist = nullptr;
    uint32_t* row_len = nullptr;
    uint32_t* off[3] = {nullptr, nullptr, nullptr};  // [0] = image rows (kept), [1],[2] ping-pong over levels
    uint32_t* cells = nullptr;
    uint32_t* blk_row = nullptr;   // first row of every 128-cell block of every level's output layout (k_block_rows)
    uint32_t blk_first[33] = {};  // level l's entries start at blk_first[l]; blk_nlev levels
    uint32_t blk_nlev = 0;
    // level buffers: cells in the 9 x 29 form as the products leave it (msm.hip), 9 words = 36 bytes per cell and column
    uint32_t* lvl[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    Fr* bsum[3] = {nullptr, nullptr, nullptr};
    Fr* win_pts = nullptr;
    Fr* tri_scratch = nullptr;
    uint64_t cap0, cap1;  // cell capacity of level buffers
    bool fused01 = false;  // the last run added levels 0 and 1 in one launch (k_add_level01)
    size_t bytes = 0;
    // stage timing (bench only): events bracket the stages of gm_msm_run on the launch stream
    int prof_mode = 0;  // 0 off, 1 dominant kernel only (level-0 add), 2 all stages
    hipEvent_t ev[GM_MSM_NSTAGE + 1] = {};
    bool ev_rec[GM_MSM_NSTAGE + 1] = {};
};

