// VecVec polynomials on the device: k polynomials sharing one ragged row structure.
//
// Mirrors /root/reference/src/cleanup/polys/vecvec.rs:
//   VecVecPolynomial{data,row_pad,col_pad,row_logsize,col_logsize}   :149-160  (+ ::new, odd rows padded :178-189)
//   vecvec_map                                                       :480-540
//   vecvec_map_split (split on the LSB, re-pad)                      :542-606
//   vecvec_map_split_to_dense                                        :608-654
//   to_dense                                                         :446-476
// and the bucket image of PushForwardState::new (pushforward/pushforward.rs:342-349, 380-381, 411-426, 477-487).
//
// Layout: rows back to back, off[r] = first cell of row r, every stored row length is even; the k columns
// are separate arrays over the same cells, so one offsets table and one binary search serve all of them.
#include "internal.hpp"
#include "msm_plan.hpp"
#include "ragged.hip.h"
#include "vecvec.hpp"

namespace gm {

// image cells: x, y, z of the point scattered to each cell; pad cells (0, 1, 0)
__global__ void __launch_bounds__(256) k_image_gather(const Fr* __restrict__ pts, const uint32_t* __restrict__ cells,
                                                       uint64_t total, Fr* __restrict__ ox, Fr* __restrict__ oy,
                                                       Fr* __restrict__ oz) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const uint32_t i = cells[j];
    Fr x = fr_zero(), y = fr_one(), z = fr_zero();
    if (i != PAD_IDX) {
        x = fr_load(pts + 2ull * i);
        y = fr_load(pts + 2ull * i + 1);
        z = fr_one();
    }
    fr_store(ox + j, x);
    fr_store(oy + j, y);
    fr_store(oz + j, z);
}

// PRIM: as k_dense_map (poly.hip): the plan's one primitive, or 0 for any plan
template <int PRIM>
__global__ void __launch_bounds__(256) k_vv_map(SegPlan sp, ColPtrs in, ColPtrsMut out, uint64_t total) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    constexpr int NI = PrimShape<PRIM>::n_in, NO = PrimShape<PRIM>::n_out;
    for (int s = 0; s < sp.nseg; s++) {
        const Seg g = sp.seg[s];
        Fr a[NI], o[NO];
#pragma unroll
        for (int q = 0; q < NI; q++)
            if (PRIM || q < g.n_in) a[q] = fr_load(in.p[g.in[q]] + i);
        if (PRIM) prim_exec(PRIM, a, o); else prim_exec(g.prim, a, o);
#pragma unroll
        for (int q = 0; q < NO; q++)
            if (PRIM || q < g.n_out) fr_store(out.p[g.out0 + q] + i, o[q]);
    }
}

#define GM_VV_MAX_OUTS 16  // VecVec layers are narrow (<= 6 in, <= 5 out); keeps kernel arguments small
struct PadVals {
    Fr v[GM_VV_MAX_OUTS];
};

// split on the LSB of the in-row index: output cell p of row r holds f(in[2p]) in the "left" columns and
// f(in[2p+1]) in the "right" columns; the cell past len/2 (when len/2 is odd) holds the output row pad.
template <int PRIM>
__global__ void __launch_bounds__(256) k_vv_map_split(SegPlan sp, ColPtrs in, ColPtrsMut out,
                                                       const uint32_t* __restrict__ off_in,
                                                       const uint32_t* __restrict__ off_out, uint32_t nrows,
                                                       uint32_t bundle, PadVals pad, const uint32_t* __restrict__ coarse_out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= off_out[nrows]) return;
    // the coarse table of the output layout (gm_vv::coarse) when the shape carries one: one load instead of 13 dependent ones
    const uint32_t r = coarse_out ? find_row_coarse(off_out, nrows, coarse_out, j) : find_row(off_out, nrows, j);
    const uint32_t p = j - off_out[r];
    const uint32_t in0 = off_in[r], half_len = (off_in[r + 1] - in0) >> 1;
    constexpr int NI = PrimShape<PRIM>::n_in, NO = PrimShape<PRIM>::n_out;
    if (p < half_len) {
        for (int s = 0; s < sp.nseg; s++) {
            const Seg g = sp.seg[s];
            for (uint32_t h = 0; h < 2; h++) {
                Fr a[NI], o[NO];
                const uint64_t src = (uint64_t)in0 + 2 * p + h;
#pragma unroll
                for (int q = 0; q < NI; q++)
                    if (PRIM || q < g.n_in) a[q] = fr_load(in.p[g.in[q]] + src);
                if (PRIM) prim_exec(PRIM, a, o); else prim_exec(g.prim, a, o);
#pragma unroll
                for (int q = 0; q < NO; q++)
                    if (PRIM || q < g.n_out) {
                        const uint32_t oc = g.out0 + q;
                        const uint32_t col = 2 * (oc / bundle) * bundle + h * bundle + oc % bundle;
                        fr_store(out.p[col] + j, o[q]);
                    }
            }
        }
    } else {
        for (int oc = 0; oc < sp.n_outs; oc++)
            for (uint32_t h = 0; h < 2; h++) {
                const uint32_t col = 2 * (oc / bundle) * bundle + h * bundle + oc % bundle;
                fr_store(out.p[col] + j, pad.v[oc]);
            }
    }
}

// row_logsize == 1: every stored row has 0 or 2 cells; output is dense over 2^col_logsize rows
// (empty row -> output row pad, missing row -> output col pad)
__global__ void __launch_bounds__(256) k_vv_map_split_to_dense(SegPlan sp, ColPtrs in, ColPtrsMut out,
                                                                const uint32_t* __restrict__ off_in, uint32_t nrows,
                                                                uint32_t nrows_dense, uint32_t bundle, PadVals row_pad,
                                                                PadVals col_pad) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows_dense) return;
    const bool stored = r < nrows;
    const uint32_t in0 = stored ? off_in[r] : 0;
    const uint32_t len = stored ? off_in[r + 1] - in0 : 0;
    if (len) {
        for (int s = 0; s < sp.nseg; s++) {
            const Seg g = sp.seg[s];
            for (uint32_t h = 0; h < 2; h++) {
                Fr a[6], o[4];
#pragma unroll
                for (int q = 0; q < 6; q++)
                    if (q < g.n_in) a[q] = fr_load(in.p[g.in[q]] + in0 + h);
                prim_exec(g.prim, a, o);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < g.n_out) {
                        const uint32_t oc = g.out0 + q;
                        const uint32_t col = 2 * (oc / bundle) * bundle + h * bundle + oc % bundle;
                        fr_store(out.p[col] + r, o[q]);
                    }
            }
        }
    } else {
        for (int oc = 0; oc < sp.n_outs; oc++)
            for (uint32_t h = 0; h < 2; h++) {
                const uint32_t col = 2 * (oc / bundle) * bundle + h * bundle + oc % bundle;
                fr_store(out.p[col] + r, stored ? row_pad.v[oc] : col_pad.v[oc]);
            }
    }
}

__global__ void __launch_bounds__(256) k_vv_to_dense(ColPtrs in, ColPtrsMut out, const uint32_t* __restrict__ off,
                                                      uint32_t nrows, uint32_t row_logsize, uint64_t n_dense,
                                                      PadVals row_pad, PadVals col_pad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dense) return;
    const uint32_t r = (uint32_t)(i >> row_logsize);
    const uint32_t c = (uint32_t)(i & ((1ull << row_logsize) - 1));
    const int col = blockIdx.y;
    Fr v;
    if (r >= nrows) v = col_pad.v[col];
    else {
        const uint32_t o0 = off[r], len = off[r + 1] - o0;
        v = (c < len) ? fr_load(in.p[col] + o0 + c) : row_pad.v[col];
    }
    fr_store(out.p[col] + i, v);
}

// scatter unpadded host-order rows into the padded layout
__global__ void __launch_bounds__(256) k_vv_pack(const Fr* __restrict__ src, const uint32_t* __restrict__ src_off,
                                                  const uint32_t* __restrict__ off, uint32_t nrows, Fr* __restrict__ dst,
                                                  Fr pad) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= off[nrows]) return;
    const uint32_t r = find_row(off, nrows, j);
    const uint32_t p = j - off[r];
    const uint32_t len = src_off[r + 1] - src_off[r];
    fr_store(dst + j, p < len ? fr_load(src + src_off[r] + p) : pad);
}

}  // namespace gm

using namespace gm;

int32_t gm_vv::alloc_cols(uint32_t k_, uint64_t total_) {
    k = k_;
    total = total_;
    cols.clear();
    for (uint32_t i = 0; i < k; i++) {
        cols.emplace_back(new DevBuf());
        int32_t rc = cols.back()->alloc((size_t)total * sizeof(Fr));
        if (rc) return rc;
    }
    row_pad.assign(k, fr_zero());
    col_pad.assign(k, fr_zero());
    return GM_OK;
}

static void pads_through(const SegPlan& sp, const gm_vv* in, std::vector<Fr>* rp, std::vector<Fr>* cp) {
    Fr a[GM_MAX_COLS], o[GM_MAX_COLS];
    for (int i = 0; i < sp.n_ins; i++) a[i] = in->row_pad[i];
    seg_plan_exec_host(sp, a, o);
    rp->assign(o, o + sp.n_outs);
    for (int i = 0; i < sp.n_ins; i++) a[i] = in->col_pad[i];
    seg_plan_exec_host(sp, a, o);
    cp->assign(o, o + sp.n_outs);
}

namespace gm {

// coarse row tables of all levels in one launch: blockIdx.y = level; entry c = row of cell min(c << S, cells - 1)
__global__ void __launch_bounds__(256) k_coarse_rows(const uint32_t* __restrict__ off_levels, uint32_t nrows, const uint64_t* __restrict__ tab_off,
                                                     uint32_t* __restrict__ coarse) {
    const uint32_t lvl = blockIdx.y;
    const uint32_t* off = off_levels + (size_t)lvl * (nrows + 1);
    const uint32_t cells = off[nrows];
    const uint32_t n = (cells >> GM_COARSE_SHIFT) + 2;
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    uint32_t j = c << GM_COARSE_SHIFT;
    uint32_t r = nrows ? nrows - 1 : 0;
    if (cells) {
        if (j >= cells) j = cells - 1;
        r = find_row(off, nrows, j);
    }
    coarse[tab_off[lvl] + c] = r;
}

// a non-owning DevBuf over part of another allocation (the holder keeps the allocation alive through off_levels)
static std::shared_ptr<DevBuf> off_view(const std::shared_ptr<DevBuf>& base, uint32_t level, uint32_t nrows) {
    std::shared_ptr<DevBuf> v(new DevBuf());
    v->p = static_cast<char*>(base->p) + (size_t)level * (nrows + 1) * 4;
    v->bytes = (size_t)(nrows + 1) * 4;
    v->owned = false;
    return v;
}

int32_t vv_map(const SegPlan& sp, const gm_vv* in, gm_vv** out, hipStream_t s) {
    // exec reads args[0..n_ins): extra trailing polys are ignored, as in the reference
    // (bintree level 0 maps affine l1 over the 6-poly GlueSplit output, bintree_add.rs:213-215)
    GM_REQUIRE((int)in->k >= sp.n_ins, "vecvec_map: %u polys for a %d-input function", in->k, sp.n_ins);
    std::unique_ptr<gm_vv> o(new gm_vv());
    o->nrows = in->nrows; o->row_logsize = in->row_logsize; o->col_logsize = in->col_logsize;
    o->max_row_len = in->max_row_len;
    o->row_base = in->row_base; o->sharded = in->sharded;
    o->off = in->off;  // shared shape
    o->share_levels(in, 0);
    int32_t rc = o->alloc_cols(sp.n_outs, in->total);
    if (rc) return rc;
    pads_through(sp, in, &o->row_pad, &o->col_pad);
    ColPtrs ci;
    ColPtrsMut co;
    for (int i = 0; i < sp.n_ins; i++) ci.p[i] = in->cols[i]->fr();
    for (int i = 0; i < sp.n_outs; i++) co.p[i] = o->cols[i]->fr();
    if (in->total) {
#define GM_LAUNCH_VV_MAP(P) hipLaunchKernelGGL(k_vv_map<P>, dim3(ceil_div(in->total, 256)), dim3(256), 0, s, sp, ci, co, in->total)
        GM_MAP_DISPATCH(uniform_prim_of(sp), GM_LAUNCH_VV_MAP)
#undef GM_LAUNCH_VV_MAP
        GM_LAUNCH_CHECK();
    }
    *out = o.release();
    return GM_OK;
}

static inline uint32_t pad2(uint32_t v) { return v + (v & 1u); }

int32_t vv_map_split(const SegPlan& sp, const gm_vv* in, uint32_t bundle, gm_vv** out, hipStream_t s) {
    GM_REQUIRE((int)in->k == sp.n_ins, "vecvec_map_split: %u polys for a %d-input function", in->k, sp.n_ins);
    GM_REQUIRE(in->row_logsize >= 1, "cannot split row_logsize 0");
    GM_REQUIRE(bundle >= 1 && sp.n_outs % (int)bundle == 0 && sp.n_outs <= GM_VV_MAX_OUTS, "bad bundle / width");
    std::unique_ptr<gm_vv> o(new gm_vv());
    o->nrows = in->nrows; o->row_logsize = in->row_logsize - 1; o->col_logsize = in->col_logsize;
    o->max_row_len = pad2(in->max_row_len / 2);
    o->row_base = in->row_base; o->sharded = in->sharded;
    int32_t rc = GM_OK;
    uint32_t tot = 0;
    if (in->off_levels && in->off_level + 1 < in->n_off_levels) {
        // the next layout is already in the table: no launch, no read-back
        o->share_levels(in, 1);
        o->off = off_view(in->off_levels, o->off_level, in->nrows);
        tot = (*in->level_totals)[o->off_level];
    } else {
        o->off.reset(new DevBuf());
        rc = o->off->alloc((size_t)(in->nrows + 1) * 4);
        if (rc) return rc;
        const uint32_t* off_in = reinterpret_cast<const uint32_t*>(in->off->p);
        uint32_t* off_out = reinterpret_cast<uint32_t*>(o->off->p);
        rc = launch_offsets_next(off_in, off_out, in->nrows, s);
        if (rc) return rc;
        // upper bound on the new total without a sync: total/2 + one pad per row; exact value read back
        GM_HIP(hipMemcpyAsync(&tot, off_out + in->nrows, 4, hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
    }
    rc = o->alloc_cols(2 * sp.n_outs, tot);
    if (rc) return rc;
    std::vector<Fr> rp, cp;
    pads_through(sp, in, &rp, &cp);
    PadVals pv;
    for (int oc = 0; oc < sp.n_outs; oc++) {
        pv.v[oc] = rp[oc];
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t col = 2 * (oc / bundle) * bundle + h * bundle + oc % bundle;
            o->row_pad[col] = rp[oc];
            o->col_pad[col] = cp[oc];
        }
    }
    ColPtrs ci;
    ColPtrsMut co;
    for (int i = 0; i < sp.n_ins; i++) ci.p[i] = in->cols[i]->fr();
    for (int i = 0; i < 2 * sp.n_outs; i++) co.p[i] = o->cols[i]->fr();
    if (tot) {
        const uint32_t* coarse_tab = (o->coarse && o->coarse_off && o->off_level < o->coarse_off->size())
                                         ? reinterpret_cast<const uint32_t*>(o->coarse->p) + (*o->coarse_off)[o->off_level] : (const uint32_t*)nullptr;
#define GM_LAUNCH_VV_MAP_SPLIT(P)                                                                                             \
    hipLaunchKernelGGL(k_vv_map_split<P>, dim3(ceil_div(tot, 256)), dim3(256), 0, s, sp, ci, co,                                 \
                       reinterpret_cast<const uint32_t*>(in->off->p), reinterpret_cast<const uint32_t*>(o->off->p), in->nrows, bundle, pv, coarse_tab)
        GM_MAP_DISPATCH(uniform_prim_of(sp), GM_LAUNCH_VV_MAP_SPLIT)
#undef GM_LAUNCH_VV_MAP_SPLIT
        GM_LAUNCH_CHECK();
    }
    *out = o.release();
    return GM_OK;
}

int32_t vv_map_split_to_dense(const SegPlan& sp, const gm_vv* in, uint32_t bundle, Fr* const* d_out, hipStream_t s) {
    GM_REQUIRE((int)in->k == sp.n_ins, "vecvec_map_split_to_dense: %u polys for a %d-input function", in->k, sp.n_ins);
    GM_REQUIRE(in->row_logsize == 1, "row_logsize must be 1 (vecvec.rs:618)");
    GM_REQUIRE(bundle >= 1 && sp.n_outs % (int)bundle == 0 && sp.n_outs <= GM_VV_MAX_OUTS, "bad bundle / width");
    std::vector<Fr> rp, cp;
    pads_through(sp, in, &rp, &cp);
    PadVals prow, pcol;
    for (int oc = 0; oc < sp.n_outs; oc++) { prow.v[oc] = rp[oc]; pcol.v[oc] = cp[oc]; }
    ColPtrs ci;
    ColPtrsMut co;
    for (int i = 0; i < sp.n_ins; i++) ci.p[i] = in->cols[i]->fr();
    for (int i = 0; i < 2 * sp.n_outs; i++) co.p[i] = d_out[i];
    const uint32_t nd = in->sharded ? in->nrows : (1u << in->col_logsize);  // sharded: this rank's rows only
    hipLaunchKernelGGL(k_vv_map_split_to_dense, dim3(ceil_div(nd, 256)), dim3(256), 0, s, sp, ci, co,
                       reinterpret_cast<const uint32_t*>(in->off->p), in->nrows, nd, bundle, prow, pcol);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // namespace gm

// ------------------------------------------------------------------------------------------- C ABI
extern "C" int32_t gm_vv_destroy(gm_vv* v) {
    delete v;
    return GM_OK;
}

extern "C" int32_t gm_vv_info(const gm_vv* v, uint32_t* k, uint32_t* nrows, uint64_t* total_cells,
                              uint32_t* row_logsize, uint32_t* col_logsize) {
    GM_REQUIRE(v, "null vv");
    if (k) *k = v->k;
    if (nrows) *nrows = v->nrows;
    if (total_cells) *total_cells = v->total;
    if (row_logsize) *row_logsize = v->row_logsize;
    if (col_logsize) *col_logsize = v->col_logsize;
    return GM_OK;
}

extern "C" int32_t gm_vv_from_host(uint32_t k, uint32_t nrows, const uint32_t* h_row_len,
                                   const uint64_t* const* h_data, const uint64_t* h_row_pad, const uint64_t* h_col_pad,
                                   uint32_t row_logsize, uint32_t col_logsize, gm_vv** out, void* stream) {
    GM_REQUIRE(out && k >= 1 && k <= GM_MAX_COLS && (nrows == 0 || h_row_len) && h_data && h_row_pad && h_col_pad,
               "bad argument");
    GM_REQUIRE(col_logsize <= 31 && row_logsize <= 31 && (uint64_t)nrows <= (1ull << col_logsize),
               "more rows than 2^col_logsize (vecvec.rs:179)");
    hipStream_t s = as_stream(stream);
    std::unique_ptr<gm_vv> v(new gm_vv());
    v->nrows = nrows; v->row_logsize = row_logsize; v->col_logsize = col_logsize;
    std::vector<uint32_t> src_off(nrows + 1, 0), off(nrows + 1, 0);
    uint32_t mx = 0;
    for (uint32_t r = 0; r < nrows; r++) {
        GM_REQUIRE((uint64_t)h_row_len[r] <= (1ull << row_logsize), "row %u longer than 2^row_logsize (vecvec.rs:181)", r);
        src_off[r + 1] = src_off[r] + h_row_len[r];
        const uint32_t pl = h_row_len[r] + (h_row_len[r] & 1u);
        off[r + 1] = off[r] + pl;
        mx = pl > mx ? pl : mx;
    }
    v->max_row_len = mx;
    v->off.reset(new DevBuf());
    int32_t rc = v->off->alloc((size_t)(nrows + 1) * 4);
    if (rc) return rc;
    GM_HIP(hipMemcpyAsync(v->off->p, off.data(), (size_t)(nrows + 1) * 4, hipMemcpyHostToDevice, s));
    rc = v->alloc_cols(k, off[nrows]);
    if (rc) return rc;
    DevBuf d_src_off, d_src;
    rc = d_src_off.alloc((size_t)(nrows + 1) * 4);
    if (rc) return rc;
    GM_HIP(hipMemcpyAsync(d_src_off.p, src_off.data(), (size_t)(nrows + 1) * 4, hipMemcpyHostToDevice, s));
    rc = d_src.alloc((size_t)src_off[nrows] * 32);
    if (rc) return rc;
    for (uint32_t c = 0; c < k; c++) {
        memcpy(&v->row_pad[c], h_row_pad + 4 * c, 32);
        memcpy(&v->col_pad[c], h_col_pad + 4 * c, 32);
        if (src_off[nrows]) GM_HIP(hipMemcpyAsync(d_src.p, h_data[c], (size_t)src_off[nrows] * 32, hipMemcpyHostToDevice, s));
        if (off[nrows]) {
            hipLaunchKernelGGL(k_vv_pack, dim3(ceil_div(off[nrows], 256)), dim3(256), 0, s, d_src.fr(),
                               reinterpret_cast<const uint32_t*>(d_src_off.p), reinterpret_cast<const uint32_t*>(v->off->p),
                               nrows, v->cols[c]->fr(), v->row_pad[c]);
            GM_LAUNCH_CHECK();
        }
        GM_HIP(hipStreamSynchronize(s));  // d_src is reused
    }
    *out = v.release();
    return GM_OK;
}

// The bucket image of the last gm_msm_run as 3 VecVec polynomials (x, y, z).  partial = false: the plan must cover all
// windows.  partial = true (sharded prover): the handle holds the rows of the plan's windows, row_base = y0 << d_logsize.
namespace gm {
int32_t vv_from_msm(const gm_msm_plan* p, const uint64_t* d_points_xy, uint32_t y_logsize, bool partial, gm_vv** out, void* stream);
}
extern "C" int32_t gm_vv_from_msm(const gm_msm_plan* p, const uint64_t* d_points_xy, uint32_t y_logsize, gm_vv** out,
                                  void* stream) {
    return gm::vv_from_msm(p, d_points_xy, y_logsize, false, out, stream);
}
int32_t gm::vv_from_msm(const gm_msm_plan* p, const uint64_t* d_points_xy, uint32_t y_logsize, bool partial, gm_vv** out,
                        void* stream) {
    GM_REQUIRE(p && d_points_xy && out, "null argument");
    GM_REQUIRE(partial || (p->y0 == 0 && p->y1 == p->y_size), "the image needs a plan over all windows");
    GM_REQUIRE((1u << y_logsize) >= p->y_size, "y_logsize too small");
    hipStream_t s = as_stream(stream);
    std::unique_ptr<gm_vv> v(new gm_vv());
    v->nrows = p->nrows; v->row_logsize = p->x_log; v->col_logsize = y_logsize + p->d_log;
    v->row_base = p->y0 << p->d_log;
    v->sharded = partial;
    // the plan holds the row layouts of all x_logsize levels (k_offsets_levels_par): keep a copy with the image, so that the
    // witness builder's splits and every layer's sumcheck object take their layouts from it
    const size_t lvl_words = (size_t)p->nrows + 1, tab_bytes = (size_t)p->x_log * lvl_words * 4;
    v->off_levels.reset(new DevBuf());
    int32_t rc = v->off_levels->alloc(tab_bytes);
    if (rc) return rc;
    GM_HIP(hipMemcpyAsync(v->off_levels->p, p->off[0], tab_bytes, hipMemcpyDeviceToDevice, s));
    v->n_off_levels = p->x_log;
    v->off_level = 0;
    v->off = off_view(v->off_levels, 0, p->nrows);
    std::vector<uint32_t> tab((size_t)p->x_log * lvl_words);
    GM_HIP(hipMemcpyAsync(tab.data(), p->off[0], tab_bytes, hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    v->level_totals.reset(new std::vector<uint32_t>(p->x_log));
    for (uint32_t l = 0; l < p->x_log; l++) (*v->level_totals)[l] = tab[(size_t)l * lvl_words + p->nrows];
    const uint32_t* off = tab.data();
    {   // coarse row tables of every level (the large VecVec round kernels bracket their row search with them)
        v->coarse_off.reset(new std::vector<uint64_t>(p->x_log));
        uint64_t words = 0;
        uint32_t max_entries = 1;
        for (uint32_t l = 0; l < p->x_log; l++) {
            (*v->coarse_off)[l] = words;
            const uint32_t n = ((*v->level_totals)[l] >> GM_COARSE_SHIFT) + 2;
            words += n;
            if (n > max_entries) max_entries = n;
        }
        v->coarse.reset(new DevBuf());
        rc = v->coarse->alloc(words * 4);
        if (rc) return rc;
        DevBuf d_tab_off;
        rc = d_tab_off.alloc((size_t)p->x_log * 8);
        if (rc) return rc;
        GM_HIP(hipMemcpyAsync(d_tab_off.p, v->coarse_off->data(), (size_t)p->x_log * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_coarse_rows, dim3(ceil_div(max_entries, 256), p->x_log), dim3(256), 0, s,
                           reinterpret_cast<const uint32_t*>(v->off_levels->p), p->nrows, reinterpret_cast<const uint64_t*>(d_tab_off.p),
                           reinterpret_cast<uint32_t*>(v->coarse->p));
        GM_LAUNCH_CHECK();
        GM_HIP(hipStreamSynchronize(s));   // d_tab_off goes out of scope
    }
    uint32_t mx = 0;
    for (uint32_t r = 0; r < p->nrows; r++) mx = (off[r + 1] - off[r] > mx) ? off[r + 1] - off[r] : mx;
    v->max_row_len = mx;
    rc = v->alloc_cols(3, off[p->nrows]);
    if (rc) return rc;
    // pads (x, y, z) = (0, 1, 0), pushforward.rs:380-381
    v->row_pad[0] = fr_zero(); v->row_pad[1] = fr_one(); v->row_pad[2] = fr_zero();
    v->col_pad = v->row_pad;
    if (v->total) {
        hipLaunchKernelGGL(k_image_gather, dim3(ceil_div(v->total, 256)), dim3(256), 0, s,
                           reinterpret_cast<const Fr*>(d_points_xy), p->cells, v->total, v->cols[0]->fr(),
                           v->cols[1]->fr(), v->cols[2]->fr());
        GM_LAUNCH_CHECK();
    }
    *out = v.release();
    return GM_OK;
}

extern "C" int32_t gm_vv_map(const gm_fn* f, const gm_vv* in, gm_vv** out, void* stream) {
    GM_REQUIRE(in && out, "null argument");
    GmFn g;
    int32_t rc = to_gmfn(f, &g);
    if (rc) return rc;
    SegPlan sp;
    GM_REQUIRE(seg_plan_build(g, &sp), "function too wide");
    return vv_map(sp, in, out, as_stream(stream));
}

extern "C" int32_t gm_vv_map_split(const gm_fn* f, const gm_vv* in, uint32_t bundle, gm_vv** out, void* stream) {
    GM_REQUIRE(in && out, "null argument");
    GmFn g;
    int32_t rc = to_gmfn(f, &g);
    if (rc) return rc;
    SegPlan sp;
    GM_REQUIRE(seg_plan_build(g, &sp), "function too wide");
    return vv_map_split(sp, in, bundle, out, as_stream(stream));
}

extern "C" int32_t gm_vv_map_split_to_dense(const gm_fn* f, const gm_vv* in, uint32_t bundle, uint64_t* const* d_out,
                                            void* stream) {
    GM_REQUIRE(in && d_out, "null argument");
    GmFn g;
    int32_t rc = to_gmfn(f, &g);
    if (rc) return rc;
    SegPlan sp;
    GM_REQUIRE(seg_plan_build(g, &sp), "function too wide");
    return vv_map_split_to_dense(sp, in, bundle, reinterpret_cast<Fr* const*>(d_out), as_stream(stream));
}

// select polys [first, first+count) of a VecVec set as a new handle sharing the storage (GlueSplit::witness
// maps polys[0..2] and polys[2..3] separately, splits.rs:172-176)
extern "C" int32_t gm_vv_slice(const gm_vv* in, uint32_t first, uint32_t count, gm_vv** out) {
    GM_REQUIRE(in && out && count >= 1 && first + count <= in->k, "bad slice");
    gm_vv* v = new gm_vv();
    v->k = count; v->nrows = in->nrows; v->total = in->total; v->row_logsize = in->row_logsize;
    v->col_logsize = in->col_logsize; v->max_row_len = in->max_row_len; v->off = in->off; v->row_base = in->row_base; v->sharded = in->sharded;
    v->share_levels(in, 0);
    for (uint32_t i = 0; i < count; i++) {
        v->cols.push_back(in->cols[first + i]);
        v->row_pad.push_back(in->row_pad[first + i]);
        v->col_pad.push_back(in->col_pad[first + i]);
    }
    *out = v;
    return GM_OK;
}

// concatenate the polynomial lists of two sets with the same shape (out.extend(...), splits.rs:174)
extern "C" int32_t gm_vv_concat(const gm_vv* a, const gm_vv* b, gm_vv** out) {
    GM_REQUIRE(a && b && out, "null argument");
    GM_REQUIRE(a->nrows == b->nrows && a->total == b->total && a->row_logsize == b->row_logsize &&
                   a->col_logsize == b->col_logsize, "shape mismatch");
    gm_vv* v = new gm_vv();
    *v = *a;  // shares columns (shared_ptr) and offsets of a
    v->k = a->k + b->k;
    v->cols.insert(v->cols.end(), b->cols.begin(), b->cols.end());
    v->row_pad.insert(v->row_pad.end(), b->row_pad.begin(), b->row_pad.end());
    v->col_pad.insert(v->col_pad.end(), b->col_pad.begin(), b->col_pad.end());
    *out = v;
    return GM_OK;
}

extern "C" int32_t gm_vv_to_dense(const gm_vv* v, uint64_t* const* d_out, void* stream) {
    GM_REQUIRE(v && d_out, "null argument");
    const uint64_t n = 1ull << (v->row_logsize + v->col_logsize);
    for (uint32_t base = 0; base < v->k; base += GM_VV_MAX_OUTS) {
        ColPtrs ci;
        ColPtrsMut co;
        PadVals rp, cp;
        const uint32_t cnt = (v->k - base < GM_VV_MAX_OUTS) ? v->k - base : GM_VV_MAX_OUTS;
        for (uint32_t i = 0; i < cnt; i++) {
            ci.p[i] = v->cols[base + i]->fr();
            co.p[i] = reinterpret_cast<Fr*>(d_out[base + i]);
            rp.v[i] = v->row_pad[base + i];
            cp.v[i] = v->col_pad[base + i];
        }
        hipLaunchKernelGGL(k_vv_to_dense, dim3(ceil_div(n, 256), cnt), dim3(256), 0, as_stream(stream), ci, co,
                           reinterpret_cast<const uint32_t*>(v->off->p), v->nrows, v->row_logsize, n, rp, cp);
        GM_LAUNCH_CHECK();
    }
    return GM_OK;
}

// raw access for tests: offsets (nrows+1 u32) and the cells of one column
extern "C" int32_t gm_vv_read(const gm_vv* v, uint32_t col, uint32_t* h_off, uint64_t* h_cells, void* stream) {
    GM_REQUIRE(v && col < v->k, "bad argument");
    hipStream_t s = as_stream(stream);
    if (h_off) GM_HIP(hipMemcpyAsync(h_off, v->off->p, (size_t)(v->nrows + 1) * 4, hipMemcpyDeviceToHost, s));
    if (h_cells && v->total)
        GM_HIP(hipMemcpyAsync(h_cells, v->cols[col]->p, (size_t)v->total * 32, hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    return GM_OK;
}

extern "C" int32_t gm_vv_pads(const gm_vv* v, uint64_t* h_row_pad, uint64_t* h_col_pad) {
    GM_REQUIRE(v && h_row_pad && h_col_pad, "null argument");
    for (uint32_t c = 0; c < v->k; c++) {
        memcpy(h_row_pad + 4 * c, &v->row_pad[c], 32);
        memcpy(h_col_pad + 4 * c, &v->col_pad[c], 32);
    }
    return GM_OK;
}
