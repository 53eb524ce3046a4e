// Dense multilinear polynomials on the device: pointwise layer maps (with or without a split),
// LSB fold, eq tables.  Restates (semantics only)
//   Vec::algfn_map / algfn_map_split   /root/reference/src/cleanup/polys/dense.rs:115-184
//   bind_dense_poly                     /root/reference/src/cleanup/protocols/sumcheck.rs:160-163
//   bind_21 (same values, see below)    /root/reference/src/cleanup/polys/dense.rs:54-61
//   eq_poly_sequence_from_multiplier    /root/reference/src/utils.rs:222-250
// Columns are separate device arrays of 32-byte Montgomery elements (struct-of-columns, AoS inside a
// column): a lane moves whole elements with two 16-byte accesses and consecutive lanes touch consecutive
// elements, so every wave-level load covers one contiguous 2 KiB span.
#include "common.hpp"
#include "segfn.hip.h"

namespace gm {

__device__ __forceinline__ void seg_eval(const Seg& g, const Fr* a, Fr* o) { prim_exec(g.prim, a, o); }

// PRIM: the one primitive every segment of the plan applies (an instance of its own: its registers, not the widest primitive's), or 0:
// any plan through prim_exec's switch
template <int PRIM>
__global__ void __launch_bounds__(256) k_dense_map(SegPlan sp, ColPtrs in, ColPtrsMut out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NI = PrimShape<PRIM>::n_in, NO = PrimShape<PRIM>::n_out;
    for (int s = 0; s < sp.nseg; s++) {
        const Seg g = sp.seg[s];
        Fr a[NI], o[NO];
#pragma unroll
        for (int q = 0; q < NI; q++)
            if (PRIM || q < g.n_in) a[q] = fr_load(in.p[g.in[q]] + i);
        if (PRIM) prim_exec(PRIM, a, o); else seg_eval(g, a, o);
#pragma unroll
        for (int q = 0; q < NO; q++)
            if (PRIM || q < g.n_out) fr_store(out.p[g.out0 + q] + i, o[q]);
    }
}

// element i goes to half (i >> lo_bit) & 1 at position ((i >> (lo_bit+1)) << lo_bit) | (i & (2^lo_bit - 1));
// output o of half h lands in column 2*(o/bundle)*bundle + h*bundle + o%bundle   (dense.rs:126-138)
template <int PRIM>
__global__ void __launch_bounds__(256) k_dense_map_split(SegPlan sp, ColPtrs in, ColPtrsMut out, uint64_t n,
                                                          uint32_t lo_bit, uint32_t bundle) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t half = (uint32_t)(i >> lo_bit) & 1u;
    const uint64_t pos = ((i >> (lo_bit + 1)) << lo_bit) | (i & ((1ull << lo_bit) - 1));
    constexpr int NI = PrimShape<PRIM>::n_in, NO = PrimShape<PRIM>::n_out;
    for (int s = 0; s < sp.nseg; s++) {
        const Seg g = sp.seg[s];
        Fr a[NI], o[NO];
#pragma unroll
        for (int q = 0; q < NI; q++)
            if (PRIM || q < g.n_in) a[q] = fr_load(in.p[g.in[q]] + i);
        if (PRIM) prim_exec(PRIM, a, o); else seg_eval(g, a, o);
#pragma unroll
        for (int q = 0; q < NO; q++)
            if (PRIM || q < g.n_out) {
                const uint32_t oc = g.out0 + q;
                const uint32_t col = 2 * (oc / bundle) * bundle + half * bundle + oc % bundle;
                fr_store(out.p[col] + pos, o[q]);
            }
    }
}

// out[c][i] = in[c][2i] + t * (in[c][2i+1] - in[c][2i]);  blockIdx.y = column
__global__ void __launch_bounds__(256) k_dense_fold(ColPtrs in, ColPtrsMut out, uint64_t n_out, Fr t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const Fr* src = in.p[blockIdx.y];
    Fr p0 = fr_load(src + 2 * i), p1 = fr_load(src + 2 * i + 1);
    fr_store(out.p[blockIdx.y] + i, fr_add(p0, fr_mul(t, fr_sub(p1, p0))));
}

// one level of the eq table: next[2j] = w - r*w, next[2j+1] = r*w
__global__ void __launch_bounds__(256) k_eq_level(const Fr* __restrict__ prev, Fr* __restrict__ next, Fr r,
                                                   uint64_t n_prev) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_prev) return;
    Fr w = fr_load(prev + j);
    Fr m = fr_mul(r, w);
    fr_store(next + 2 * j, fr_sub(w, m));
    fr_store(next + 2 * j + 1, m);
}

__global__ void k_set1(Fr* dst, Fr v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) fr_store(dst, v);
}

}  // namespace gm

using namespace gm;

namespace gm {

int32_t to_gmfn(const gm_fn* f, GmFn* g) {
    if (!f || f->nseg < 1 || f->nseg > GM_FN_MAX_SEG) return set_err(GM_ERR_INVALID, "bad gm_fn");
    g->nseg = f->nseg;
    for (int s = 0; s < f->nseg; s++) {
        if (f->prim[s] < 1 || f->prim[s] > 12 || f->count[s] < 0) return set_err(GM_ERR_INVALID, "bad gm_fn segment %d", s);
        g->prim[s] = f->prim[s];
        g->count[s] = f->count[s];
    }
    return GM_OK;
}

int32_t launch_dense_map(const SegPlan& sp, const Fr* const* in, Fr* const* out, uint64_t n, hipStream_t s) {
    ColPtrs ci;
    ColPtrsMut co;
    for (int i = 0; i < sp.n_ins; i++) ci.p[i] = in[i];
    for (int i = 0; i < sp.n_outs; i++) co.p[i] = out[i];
    if (n == 0) return GM_OK;
#define GM_LAUNCH_DENSE_MAP(P) hipLaunchKernelGGL(k_dense_map<P>, dim3(ceil_div(n, 256)), dim3(256), 0, s, sp, ci, co, n)
    GM_MAP_DISPATCH(uniform_prim_of(sp), GM_LAUNCH_DENSE_MAP)
#undef GM_LAUNCH_DENSE_MAP
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int32_t launch_dense_map_split(const SegPlan& sp, const Fr* const* in, Fr* const* out, uint64_t n, uint32_t lo_bit,
                               uint32_t bundle, hipStream_t s) {
    ColPtrs ci;
    ColPtrsMut co;
    for (int i = 0; i < sp.n_ins; i++) ci.p[i] = in[i];
    for (int i = 0; i < 2 * sp.n_outs; i++) co.p[i] = out[i];
    if (n == 0) return GM_OK;
#define GM_LAUNCH_DENSE_MAP_SPLIT(P) hipLaunchKernelGGL(k_dense_map_split<P>, dim3(ceil_div(n, 256)), dim3(256), 0, s, sp, ci, co, n, lo_bit, bundle)
    GM_MAP_DISPATCH(uniform_prim_of(sp), GM_LAUNCH_DENSE_MAP_SPLIT)
#undef GM_LAUNCH_DENSE_MAP_SPLIT
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int32_t launch_dense_fold(const Fr* const* in, Fr* const* out, int k, uint64_t n_out, const Fr& t, hipStream_t s) {
    if (n_out == 0 || k == 0) return GM_OK;
    for (int base = 0; base < k; base += GM_MAX_COLS) {
        ColPtrs ci;
        ColPtrsMut co;
        int cnt = (k - base < GM_MAX_COLS) ? k - base : GM_MAX_COLS;
        for (int i = 0; i < cnt; i++) { ci.p[i] = in[base + i]; co.p[i] = out[base + i]; }
        hipLaunchKernelGGL(k_dense_fold, dim3(ceil_div(n_out, 256), cnt), dim3(256), 0, s, ci, co, n_out, t);
        GM_LAUNCH_CHECK();
    }
    return GM_OK;
}

// levels[i] must hold 2^i elements (i = 0..nvars); level i = eq(pt[0..i], .) * mult   (pt[0] = MSB)
// Small tables dominate the prover (one per layer, 2^13 entries and below): all their levels are built by ONE
// single-workgroup launch (a launch costs more than the arithmetic); larger levels continue one launch per level.
#define EQ_SMALL_LEVELS 14
struct EqSmallArgs {
    Fr* level[EQ_SMALL_LEVELS + 1];
    Fr pt[EQ_SMALL_LEVELS];
    Fr mult;
    uint32_t nlev;  // levels 1..nlev are computed here
};
__global__ void __launch_bounds__(1024) k_eq_small(EqSmallArgs a) {
    // Levels of up to 1024 entries ping-pong through LDS (a level is read back right after it is written: through global memory
    // that costs an L2 round trip per level, ~1 us of the ~1.8 us a level takes); every level is also stored to its global array.
    __shared__ Fr buf[2][1024];
    if (threadIdx.x == 0) { fr_store(a.level[0], a.mult); buf[0][0] = a.mult; }
    __syncthreads();
    for (uint32_t i = 1; i <= a.nlev; i++) {
        const uint32_t np = 1u << (i - 1);
        const Fr r = a.pt[i - 1];
        const bool src_lds = np <= 1024, dst_lds = 2 * np <= 1024;
        for (uint32_t j = threadIdx.x; j < np; j += blockDim.x) {
            const Fr w = src_lds ? buf[(i - 1) & 1][j] : fr_load(a.level[i - 1] + j);
            const Fr m = fr_mul(r, w);
            const Fr lo = fr_sub(w, m);
            fr_store(a.level[i] + 2 * j, lo);
            fr_store(a.level[i] + 2 * j + 1, m);
            if (dst_lds) { buf[i & 1][2 * j] = lo; buf[i & 1][2 * j + 1] = m; }
        }
        __syncthreads();  // same workgroup: the stores above are visible to the loads of the next level
    }
}

// Two independent small sequences in ONE launch, one workgroup each (a VecVec sumcheck object needs eq(point[0..col]) and the
// padded row sequence: 36 + 13 us back to back as two launches), plus up to 16 scalars stored by workgroup 1 (the scalar levels of
// the padded sequence: one more launch saved).
struct EqPairArgs {
    EqSmallArgs a[2];
    Fr scal[16];
    Fr* scal_dst;
    uint32_t n_scal;
    Fr extra[16];     // e.g. the layer's gamma powers: stored by workgroup (2, 0) of the tree form / workgroup 0 of the pair form
    Fr* extra_dst;
    uint32_t n_extra;
};
__global__ void __launch_bounds__(1024) k_eq_small_pair(EqPairArgs q) {
    __shared__ Fr buf[2][1024];
    const EqSmallArgs& a = q.a[blockIdx.x];
    if (blockIdx.x == 1 && threadIdx.x < q.n_scal) fr_store(q.scal_dst + threadIdx.x, q.scal[threadIdx.x]);
    if (blockIdx.x == 0 && threadIdx.x < q.n_extra) fr_store(q.extra_dst + threadIdx.x, q.extra[threadIdx.x]);
    if (threadIdx.x == 0) { fr_store(a.level[0], a.mult); buf[0][0] = a.mult; }
    __syncthreads();
    for (uint32_t i = 1; i <= a.nlev; i++) {
        const uint32_t np = 1u << (i - 1);
        const Fr r = a.pt[i - 1];
        const bool src_lds = np <= 1024, dst_lds = 2 * np <= 1024;
        for (uint32_t j = threadIdx.x; j < np; j += blockDim.x) {
            const Fr w = src_lds ? buf[(i - 1) & 1][j] : fr_load(a.level[i - 1] + j);
            const Fr m = fr_mul(r, w);
            const Fr lo = fr_sub(w, m);
            fr_store(a.level[i] + 2 * j, lo);
            fr_store(a.level[i] + 2 * j + 1, m);
            if (dst_lds) { buf[i & 1][2 * j] = lo; buf[i & 1][2 * j + 1] = m; }
        }
        __syncthreads();
    }
}
// The same sequences with the big levels spread over the chip (round 3).  The doubling structure is a tree: the entries of level i
// below node b of level T are a function of that node alone, so 2^T workgroups each walk the T multiplications from the root to
// their node (one lane) and then expand ONLY their subtree through LDS: the last level of a 2^13-entry table is 512 entries per
// workgroup instead of 8192 in one (38 us -> ~13 us per VecVec layer at config B, 54 layers per proof).  Every entry is computed by
// the same operations in the same order as in k_eq_small (m = r w, lo = w - m): identical field elements.
// grid (2^EQ_TOP, number of sequences); workgroup (0, s) also stores the levels 0 .. T of sequence s (<= 2^T entries each) and, for
// the pair form, workgroup (1, 1) the scalars.
#define EQ_TOP 4
__device__ __forceinline__ Fr eq_path_value(const EqSmallArgs& a, uint32_t level, uint32_t node) {
    Fr w = a.mult;
    for (uint32_t i = 1; i <= level; i++) {
        const Fr m = fr_mul(a.pt[i - 1], w);
        w = ((node >> (level - i)) & 1u) ? m : fr_sub(w, m);
    }
    return w;
}
__global__ void __launch_bounds__(256) k_eq_small_tree(EqPairArgs q) {
    __shared__ Fr buf[2][1024];
    const EqSmallArgs& a = q.a[blockIdx.y];
    const uint32_t b = blockIdx.x, i0 = threadIdx.x;
    if (blockIdx.y == 1 && b == 1 && i0 < q.n_scal) fr_store(q.scal_dst + i0, q.scal[i0]);
    if (blockIdx.y == 0 && b == 2 && i0 < q.n_extra) fr_store(q.extra_dst + i0, q.extra[i0]);
    const uint32_t T = a.nlev < EQ_TOP ? a.nlev : EQ_TOP;
    if (b == 0) {   // the top of the tree: levels 0 .. T, every node from its own path
        for (uint32_t e = i0; e < (2u << T) - 1u; e += blockDim.x) {
            uint32_t lv = 0;
            while ((2u << lv) - 1u <= e) lv++;
            const uint32_t node = e - ((1u << lv) - 1u);
            fr_store(a.level[lv] + node, eq_path_value(a, lv, node));
        }
    }
    if (b >= (1u << T) || a.nlev <= T) return;
    if (i0 == 0) buf[T & 1][0] = eq_path_value(a, T, b);
    __syncthreads();
    for (uint32_t i = T + 1; i <= a.nlev; i++) {
        const uint32_t np = 1u << (i - 1 - T);          // parents of this workgroup at level i - 1
        const Fr r = a.pt[i - 1];
        Fr* dst = a.level[i] + ((uint64_t)b << (i - T));
        for (uint32_t j = i0; j < np; j += blockDim.x) {
            const Fr w = buf[(i - 1) & 1][j];
            const Fr m = fr_mul(r, w);
            const Fr lo = fr_sub(w, m);
            fr_store(dst + 2 * j, lo);
            fr_store(dst + 2 * j + 1, m);
            if (2 * np <= 1024) { buf[i & 1][2 * j] = lo; buf[i & 1][2 * j + 1] = m; }
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(64) k_store_extra(EqPairArgs q) {
    if (threadIdx.x < q.n_extra) fr_store(q.extra_dst + threadIdx.x, q.extra[threadIdx.x]);
}
static bool eq_tree_enabled() {
    static const bool v = [] { const char* e = getenv("GM_EQ_TREE"); return !(e && e[0] == '0'); }();
    return v;
}

// false: does not fit one launch (a sequence longer than EQ_SMALL_LEVELS or too many scalars): use launch_eq_sequence
bool launch_eq_pair(const Fr& mult0, const Fr* pt0, uint32_t nvars0, Fr* const* levels0, const Fr& mult1, const Fr* pt1, uint32_t nvars1,
                    Fr* const* levels1, const Fr* scal, uint32_t n_scal, Fr* scal_dst, hipStream_t s, const Fr* extra, uint32_t n_extra,
                    Fr* extra_dst) {
    if (nvars0 > EQ_SMALL_LEVELS || nvars1 > EQ_SMALL_LEVELS || n_scal > 16 || n_extra > 16) return false;
    EqPairArgs q;
    q.a[0].mult = mult0; q.a[0].nlev = nvars0;
    for (uint32_t i = 0; i <= nvars0; i++) q.a[0].level[i] = levels0[i];
    for (uint32_t i = 0; i < nvars0; i++) q.a[0].pt[i] = pt0[i];
    q.a[1].mult = mult1; q.a[1].nlev = nvars1;
    for (uint32_t i = 0; i <= nvars1; i++) q.a[1].level[i] = levels1[i];
    for (uint32_t i = 0; i < nvars1; i++) q.a[1].pt[i] = pt1[i];
    for (uint32_t i = 0; i < n_scal; i++) q.scal[i] = scal[i];
    q.scal_dst = scal_dst;
    q.n_scal = n_scal;
    for (uint32_t i = 0; i < n_extra; i++) q.extra[i] = extra[i];
    q.extra_dst = extra_dst;
    q.n_extra = n_extra;
    // the tree form needs every level's local share to fit its LDS line: nlev - EQ_TOP <= 10
    if (eq_tree_enabled() && nvars0 <= EQ_TOP + 10 && nvars1 <= EQ_TOP + 10)
        hipLaunchKernelGGL(k_eq_small_tree, dim3(1u << EQ_TOP, 2), dim3(256), 0, s, q);
    else
        hipLaunchKernelGGL(k_eq_small_pair, dim3(2), dim3(1024), 0, s, q);
    return hipGetLastError() == hipSuccess;
}

int32_t launch_eq_sequence(const Fr& mult, const Fr* pt, uint32_t nvars, Fr* const* levels, hipStream_t s, const Fr* extra, uint32_t n_extra,
                           Fr* extra_dst) {
    const uint32_t small = nvars < EQ_SMALL_LEVELS ? nvars : EQ_SMALL_LEVELS;
    bool extra_done = n_extra == 0;
    {
        EqSmallArgs a;
        a.mult = mult;
        a.nlev = small;
        for (uint32_t i = 0; i <= small; i++) a.level[i] = levels[i];
        for (uint32_t i = 0; i < small; i++) a.pt[i] = pt[i];
        if (eq_tree_enabled() && small > EQ_TOP + 2) {
            EqPairArgs q;
            q.a[0] = a;
            q.a[1] = a;
            q.n_scal = 0;
            q.scal_dst = nullptr;
            q.n_extra = 0;
            q.extra_dst = nullptr;
            if (!extra_done && n_extra <= 16) {
                for (uint32_t i = 0; i < n_extra; i++) q.extra[i] = extra[i];
                q.extra_dst = extra_dst;
                q.n_extra = n_extra;
                extra_done = true;
            }
            hipLaunchKernelGGL(k_eq_small_tree, dim3(1u << EQ_TOP, 1), dim3(256), 0, s, q);
        } else {
            hipLaunchKernelGGL(k_eq_small, dim3(1), dim3(small >= 10 ? 1024 : 256), 0, s, a);
        }
        GM_LAUNCH_CHECK();
    }
    if (!extra_done) {   // no tree launch to ride on: a launch of its own
        for (uint32_t base = 0; base < n_extra; base += 16) {
            EqPairArgs q;
            q.a[0].nlev = 0; q.a[1].nlev = 0;
            q.n_scal = 0; q.scal_dst = nullptr;
            q.n_extra = n_extra - base < 16 ? n_extra - base : 16;
            for (uint32_t i = 0; i < q.n_extra; i++) q.extra[i] = extra[base + i];
            q.extra_dst = extra_dst + base;
            hipLaunchKernelGGL(k_store_extra, dim3(1), dim3(64), 0, s, q);
            GM_LAUNCH_CHECK();
        }
    }
    for (uint32_t i = small + 1; i <= nvars; i++) {
        const uint64_t np = 1ull << (i - 1);
        hipLaunchKernelGGL(k_eq_level, dim3(ceil_div(np, 256)), dim3(256), 0, s, levels[i - 1], levels[i], pt[i - 1], np);
        GM_LAUNCH_CHECK();
    }
    return GM_OK;
}

}  // namespace gm

// ------------------------------------------------------------------------------------------- C ABI
extern "C" int32_t gm_dense_map(const gm_fn* f, const uint64_t* const* d_in, uint64_t* const* d_out, uint64_t len,
                                void* stream) {
    GmFn g;
    int32_t rc = to_gmfn(f, &g);
    if (rc) return rc;
    SegPlan sp;
    GM_REQUIRE(seg_plan_build(g, &sp), "function too wide (max %d columns / %d segments)", GM_MAX_COLS, GM_MAX_SEGS);
    GM_REQUIRE(d_in && d_out, "null argument");
    return launch_dense_map(sp, reinterpret_cast<const Fr* const*>(d_in), reinterpret_cast<Fr* const*>(d_out), len,
                            as_stream(stream));
}

extern "C" int32_t gm_dense_map_split(const gm_fn* f, const uint64_t* const* d_in, uint64_t* const* d_out,
                                      uint64_t len, uint32_t split_lo_bit, uint32_t bundle, void* stream) {
    GmFn g;
    int32_t rc = to_gmfn(f, &g);
    if (rc) return rc;
    SegPlan sp;
    GM_REQUIRE(seg_plan_build(g, &sp), "function too wide");
    GM_REQUIRE(d_in && d_out && bundle >= 1, "bad argument");
    GM_REQUIRE(2 * sp.n_outs <= GM_MAX_COLS, "too many output columns");
    GM_REQUIRE((len & (len - 1)) == 0 && (2ull << split_lo_bit) <= len, "len must be a power of two > 2^split_lo_bit");
    GM_REQUIRE(sp.n_outs % (int)bundle == 0, "n_outs must be a multiple of the bundle size");
    return launch_dense_map_split(sp, reinterpret_cast<const Fr* const*>(d_in), reinterpret_cast<Fr* const*>(d_out),
                                  len, split_lo_bit, bundle, as_stream(stream));
}

extern "C" int32_t gm_dense_bind(const uint64_t* const* d_in, uint64_t* const* d_out, uint32_t k, uint64_t len,
                                 const uint64_t* h_t, void* stream) {
    GM_REQUIRE(d_in && d_out && h_t && len % 2 == 0, "bad argument");
    Fr t;
    memcpy(&t, h_t, 32);
    return launch_dense_fold(reinterpret_cast<const Fr* const*>(d_in), reinterpret_cast<Fr* const*>(d_out), (int)k,
                             len / 2, t, as_stream(stream));
}

extern "C" int32_t gm_eq_table(const uint64_t* h_multiplier, const uint64_t* h_point, uint32_t nvars,
                               uint64_t* d_scratch, uint64_t* d_out, void* stream) {
    GM_REQUIRE(h_multiplier && h_point && d_out && nvars <= 30, "bad argument");
    GM_REQUIRE(nvars == 0 || d_scratch, "scratch (2^nvars elements) required");
    Fr mult, pt[32];
    memcpy(&mult, h_multiplier, 32);
    memcpy(pt, h_point, 32 * (size_t)nvars);
    // levels 0..nvars-1 packed in scratch at offsets 2^i - 1, last level in d_out
    Fr* lv[33];
    Fr* sc = reinterpret_cast<Fr*>(d_scratch);
    for (uint32_t i = 0; i < nvars; i++) lv[i] = sc + ((1ull << i) - 1);
    lv[nvars] = reinterpret_cast<Fr*>(d_out);
    return launch_eq_sequence(mult, pt, nvars, lv, as_stream(stream), nullptr, 0, nullptr);
}
