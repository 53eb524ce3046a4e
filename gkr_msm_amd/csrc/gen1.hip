// gen-1 prover `gkr_msm_prove` (the bit-decomposed MSM circuit) on the device, C++ host driver mirroring
//   /root/reference/src/gkr_msm_simple.rs:86-338      gkr_msm_prove (layer list :248-269, pt_bit_choice :82-84)
//   /root/reference/src/protocol/bintree.rs:168-288   BintreeProtocol::witness, BintreeProver::round
//   /root/reference/src/protocol/sumcheck.rs:185-257  SumcheckPolyMapProver::round (+ FragmentedLincomb :67-151)
//   /root/reference/src/protocol/split.rs:37-82       Split::witness, SplitProver::round
// `gkr_msm_prove` only builds `Shape::full` polynomials (gkr_msm_simple.rs:150): a FragmentedPoly is then a plain vector,
// `split` is the even/odd de-interleave (fragmented.rs:676-732) and the EqPoly co-polynomial (copoly.rs:457-633) is the
// plain eq table, which folds like every other column.  The round polynomial of FragmentedLincomb::unipoly (evaluations at
// 0..deg+1 of sum_i (sum_o gamma^o f_o(p(i))) * eq(i)) is what the generic dense round object already computes
// (sumcheck.rs:283-327 has the same sums with P(0) = claim - P(1), the same field element for a consistent claim).
// The G1 commitments of the bit / point columns (binary_msm, G::msm over BLS12-381; gkr_msm_simple.rs:120-147) are SURVEY 8f-1.
#include <memory>
#include <vector>

#include "internal.hpp"

using namespace gm;

extern "C" {
struct gm_sc;
int32_t gm_sc_dense_deg2_create(const gm_fn* f, uint32_t num_vars, const uint64_t* const* d_cols, const uint64_t* h_point,
                                const uint64_t* h_gamma, const uint64_t* h_claims, gm_sc** out, void* stream);
int32_t gm_sc_dense_create(int32_t kind, const gm_fn* f, uint32_t num_vars, const uint64_t* const* d_cols,
                           const uint64_t* h_gamma, const uint64_t* h_claim, gm_sc** out, void* stream);
int32_t gm_sc_unipoly(gm_sc* so, uint64_t* h_coeffs, uint32_t* n_coeffs);
int32_t gm_sc_bind(gm_sc* so, const uint64_t* h_t);
int32_t gm_sc_final_evals(gm_sc* so, uint64_t* h_evals, uint32_t* n_evals);
int32_t gm_sc_destroy(gm_sc* so);
}

namespace gm {

// index = point * 2^lb + bit  (gkr_msm_simple.rs:120, 161-165)
__global__ void __launch_bounds__(256) k_gen1_base(const Fr* __restrict__ pts, const uint8_t* __restrict__ bits, uint32_t lb,
                                                    uint64_t n, Fr* __restrict__ ob, Fr* __restrict__ ox, Fr* __restrict__ oy) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = i >> lb;
    fr_store(ob + i, bits[i] ? fr_one() : fr_zero());
    fr_store(ox + i, fr_load(pts + 2 * p));
    fr_store(oy + i, fr_load(pts + 2 * p + 1));
}

}  // namespace gm

namespace {

#define TRY(x)                 \
    do {                       \
        int32_t rc__ = (x);    \
        if (rc__) return rc__; \
    } while (0)

struct Cols {
    std::vector<std::shared_ptr<DevBuf>> c;
    uint64_t len = 0;
};

struct L1 {
    bool is_map;
    gm_fn f;       // map
    int n_split;   // split: number of input polys
    uint32_t nv;   // variables of the layer input
};

gm_fn prim(int id) {
    gm_fn f;
    memset(&f, 0, sizeof(f));
    f.nseg = 1; f.prim[0] = id; f.count[0] = 1;
    return f;
}

int32_t alloc_cols(int k, uint64_t len, Cols* out) {
    out->c.clear();
    out->len = len;
    for (int i = 0; i < k; i++) {
        out->c.emplace_back(new DevBuf());
        TRY(out->c.back()->alloc((size_t)len * sizeof(Fr)));
    }
    return GM_OK;
}

struct ScHolder {
    gm_sc* so = nullptr;
    ~ScHolder() { if (so) gm_sc_destroy(so); }
};

}  // namespace

// d_points_xy: 2^lp affine points (64 B each); d_scalar_bits: 2^lp * 2^lb bytes (Vec<Vec<bool>> flattened, 0/1);
// h_tape: challenges in draw order, canonical field elements (4 x u64 each, < p: gen-1 reduces 64 bytes mod p);
// outputs: h_msgs = everything appended to the transcript in order (output polys, round polynomials as full coefficient
// vectors, final evaluations); h_output = the 3 output polys (3 * 2^lb elements); final claim (point of lp + lb, 3 evs).
static int32_t gkr_msm_prove_impl(const uint64_t* d_points_xy, const uint8_t* d_scalar_bits, uint32_t log_num_points,
                                  uint32_t log_num_scalar_bits, const uint64_t* h_tape, uint64_t n_tape, const gm_transcript* cb,
                                  uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_output,
                                  uint64_t* h_final_point, uint32_t* n_final_point, uint64_t* h_final_evs, uint64_t* tape_used,
                                  uint64_t* rounds, double* witness_ms, void* stream) {
    GM_REQUIRE(d_points_xy && d_scalar_bits && (h_tape || cb), "null argument");
    GM_REQUIRE(log_num_points >= 1 && log_num_scalar_bits >= 1 && log_num_points + log_num_scalar_bits <= 30, "bad sizes");
    hipStream_t s = as_stream(stream);
    const uint32_t lp = log_num_points, lb = log_num_scalar_bits, nv0 = lp + lb;
    const uint64_t n0 = 1ull << nv0;

    // layer list (gkr_msm_simple.rs:248-269) unrolled with its variable counts (bintree.rs:81-123)
    std::vector<L1> layers;
    {
        uint32_t nv = nv0;
        auto map = [&](int id) { layers.push_back(L1{true, prim(id), 0, nv}); };
        auto split = [&](int n) { layers.push_back(L1{false, prim(GM_FN_ID), n, nv}); nv--; };
        map(GM_FN_PT_BIT_CHOICE);
        split(2);
        map(GM_FN_AFF_L1); map(GM_FN_AFF_L2); map(GM_FN_AFF_L3);
        for (uint32_t i = 0; i + 1 < lp; i++) { split(3); map(GM_FN_PROJ_L1); map(GM_FN_PROJ_L2); map(GM_FN_PROJ_L3); }
    }

    hipEvent_t e0, e1;
    GM_HIP(hipEventCreate(&e0));
    GM_HIP(hipEventCreate(&e1));
    GM_HIP(hipEventRecord(e0, s));
    // base layer + witness.  trace[i] = input of layer i, kept only where the prover needs it:
    //   * Split layers prove without polynomial data (SplitProver::round folds claims, split.rs:66-82): nothing is kept;
    //   * the base polynomials (bit, px, py) are the 2^lb-fold expansion of the inputs: they are rebuilt from the points and bits
    //     right before the first layer is proven (the last step of the reverse pass) instead of sitting in HBM throughout;
    //   * every entry is released as soon as its layer is proven.
    // At log_num_points = 20, 2^8 bits this is 96 GiB of trace at its peak instead of 160 GiB (and a 26 GiB workspace
    // instead of 48): BASELINE.json configs[2] fits one MI355X with room to spare.
    std::vector<Cols> trace(layers.size());
    Cols cur;
    TRY(alloc_cols(3, n0, &cur));
    hipLaunchKernelGGL(k_gen1_base, dim3(ceil_div(n0, 256)), dim3(256), 0, s, reinterpret_cast<const Fr*>(d_points_xy),
                       d_scalar_bits, lb, n0, cur.c[0]->fr(), cur.c[1]->fr(), cur.c[2]->fr());
    GM_LAUNCH_CHECK();
    size_t arena_need = 0;
    for (size_t li = 0; li < layers.size(); li++) {
        const L1& L = layers[li];
        if (L.is_map && li > 0) trace[li] = cur;
        GmFn g;
        SegPlan sp;
        Cols nxt;
        std::vector<const Fr*> ci;
        std::vector<Fr*> co;
        for (auto& c : cur.c) ci.push_back(c->fr());
        if (L.is_map) {
            TRY(to_gmfn(&L.f, &g));
            seg_plan_build(g, &sp);
            TRY(alloc_cols(sp.n_outs, cur.len, &nxt));
            for (auto& c : nxt.c) co.push_back(c->fr());
            TRY(launch_dense_map(sp, ci.data(), co.data(), cur.len, s));
            // workspace of this layer's sumcheck object: fold buffers of 1/2 and 1/4 of every column, the eq levels (2^nv), slack
            const size_t need = (size_t)sp.n_ins * (cur.len / 2 + cur.len / 4 + 8) * sizeof(Fr) + (size_t)cur.len * sizeof(Fr);
            if (need > arena_need) arena_need = need;
        } else {
            gm_fn idn = prim(GM_FN_ID);
            idn.count[0] = L.n_split;
            TRY(to_gmfn(&idn, &g));
            seg_plan_build(g, &sp);
            TRY(alloc_cols(2 * L.n_split, cur.len / 2, &nxt));
            for (auto& c : nxt.c) co.push_back(c->fr());
            TRY(launch_dense_map_split(sp, ci.data(), co.data(), cur.len, 0, (uint32_t)L.n_split, s));
        }
        cur = nxt;   // the previous columns go back to the pool here unless the trace holds them (stream order keeps reuse safe)
    }
    GM_HIP(hipEventRecord(e1, s));
    // output polys (3 x 2^lb): transcript + claim
    const uint64_t nout = 1ull << lb;
    std::vector<Fr> msgs;
    std::vector<std::vector<Fr>> out(3, std::vector<Fr>(nout));
    for (int c = 0; c < 3; c++) GM_HIP(hipMemcpyAsync(out[c].data(), cur.c[c]->p, nout * sizeof(Fr), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    if (witness_ms) {
        float ms = 0;
        GM_HIP(hipEventElapsedTime(&ms, e0, e1));
        *witness_ms = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    int32_t cb_rc = 0;
    auto emit = [&](const Fr* v, size_t n) {  // transcript.append_scalars
        msgs.insert(msgs.end(), v, v + n);
        if (cb && cb->write_scalars && !cb_rc && n) cb_rc = cb->write_scalars(cb->ctx, reinterpret_cast<const uint64_t*>(v), n);
    };
    for (int c = 0; c < 3; c++) {
        emit(out[c].data(), out[c].size());
        if (h_output) memcpy(h_output + (size_t)c * nout * 4, out[c].data(), nout * sizeof(Fr));
    }
    uint64_t pos = 0, nrounds = 0;
    auto challenge = [&](Fr* c) -> int32_t {
        Fr v;
        if (cb) {
            if (cb_rc) return set_err(GM_ERR_STATE, "transcript write_scalars callback failed with %d", cb_rc);
            const int32_t rc = cb->challenge(cb->ctx, 1, 512, reinterpret_cast<uint64_t*>(&v));  // challenge_scalar: 64 bytes mod p
            if (rc) return set_err(GM_ERR_STATE, "transcript challenge callback failed with %d", rc);
        } else {
            if (pos >= n_tape) return set_err(GM_ERR_INVALID, "challenge tape exhausted after %llu challenges", (unsigned long long)pos);
            memcpy(&v, h_tape + 4 * pos, 32);
        }
        pos++;
        *c = fr_to_mont(v);
        return GM_OK;
    };
    std::vector<Fr> point(lb), evs(3);
    for (uint32_t i = 0; i < lb; i++) TRY(challenge(&point[i]));
    for (int c = 0; c < 3; c++) {  // FragmentedPoly::evaluate (fragmented.rs:748-761)
        std::vector<Fr> v = out[c];
        for (int k = (int)lb - 1; k >= 0; k--) {
            for (size_t i = 0; i < v.size() / 2; i++) v[i] = fr_add(v[2 * i], fr_mul(point[k], fr_sub(v[2 * i + 1], v[2 * i])));
            v.resize(v.size() / 2);
        }
        evs[c] = v[0];
    }

    // workspace for the sumcheck objects of one layer at a time
    cur = Cols();   // the output columns are on the host now
    Arena arena;
    TRY(arena.init(arena_need + ((size_t)64 << 20)));
    Fr* pinned = nullptr;
    TRY(thread_pinned_staging(&pinned));
    SharedPinnedScope pinned_scope(pinned);

    // BintreeProver::round loop (bintree.rs:213-288): layers in reverse, one challenge per call
    for (size_t li = layers.size(); li-- > 0;) {
        const L1& L = layers[li];
        Fr c0;
        TRY(challenge(&c0));
        if (!L.is_map) {
            // SplitProver::round (split.rs:66-82): fold the halves, fix_var_top
            const size_t h = evs.size() / 2;
            std::vector<Fr> nw(h);
            for (size_t i = 0; i < h; i++) nw[i] = fr_add(evs[i], fr_mul(c0, fr_sub(evs[h + i], evs[i])));
            evs = nw;
            point.push_back(c0);
            continue;
        }
        // SumcheckPolyMapProver::round (sumcheck.rs:197-257); first challenge = gamma
        arena.reset();
        ArenaScope scope(&arena);
        GmFn g;
        SegPlan sp;
        TRY(to_gmfn(&L.f, &g));
        seg_plan_build(g, &sp);
        const uint32_t nv = L.nv;
        GM_REQUIRE((int)evs.size() == sp.n_outs && point.size() == nv, "claim shape mismatch at layer %zu", li);
        // folded claim: sum_i ev_i gamma^i (make_folded_claim, sumcheck.rs:659-673)
        Fr claim = fr_zero(), gp = fr_one();
        for (size_t i = 0; i < evs.size(); i++) { claim = fr_add(claim, fr_mul(evs[i], gp)); gp = fr_mul(gp, c0); }
        // Round polynomial of sum_x eq(point, x) sum_i gamma^i f_i(p(x))  (FragmentedLincomb::unipoly, sumcheck.rs:99-151).
        // The reference evaluates it at deg + 2 points with the eq table as one more factor; the polynomial itself is what
        // goes on the transcript, so it is computed here with eq factored out (two evaluations of f per pair and no eq
        // column to fold -- the DenseDeg2 object, dense_eq.rs:98-173 + from12): the same coefficients, exactly.  from12
        // divides by 1 - point_j; if a coordinate equals 1 (probability 2^-255) the generic object is used instead.
        bool coord_is_one = false;
        for (const Fr& c : point) coord_is_one = coord_is_one || fr_eq(c, fr_one());
        if (li == 0) {   // the base polynomials, rebuilt from the inputs (outside the layer workspace)
            ArenaScope none(nullptr);
            TRY(alloc_cols(3, n0, &trace[0]));
            hipLaunchKernelGGL(k_gen1_base, dim3(ceil_div(n0, 256)), dim3(256), 0, s, reinterpret_cast<const Fr*>(d_points_xy),
                               d_scalar_bits, lb, n0, trace[0].c[0]->fr(), trace[0].c[1]->fr(), trace[0].c[2]->fr());
            GM_LAUNCH_CHECK();
        }
        std::vector<const uint64_t*> cols;
        for (int i = 0; i < sp.n_ins; i++) cols.push_back(reinterpret_cast<const uint64_t*>(trace[li].c[i]->p));
        ScHolder h;
        DevBuf eqbuf;
        Arena big;   // the generic object (a claim coordinate equal to 1: probability 2^-255) needs a larger workspace
        std::unique_ptr<ArenaScope> big_scope;
        if (coord_is_one || sp.deg != 2) {
            TRY(big.init((size_t)(sp.n_ins + 4) * ((size_t)1 << nv) * sizeof(Fr) + ((size_t)64 << 20)));
            big_scope.reset(new ArenaScope(&big));
        }
        if (!coord_is_one && sp.deg == 2) {
            TRY(gm_sc_dense_deg2_create(&L.f, nv, cols.data(), reinterpret_cast<const uint64_t*>(point.data()),
                                        reinterpret_cast<const uint64_t*>(&c0), reinterpret_cast<const uint64_t*>(evs.data()),
                                        &h.so, stream));
        } else {
            // EqPoly(point) on the full shape = the eq table; it becomes the last column
            TRY(eqbuf.alloc(((size_t)2 << nv) * sizeof(Fr)));
            std::vector<Fr*> lv(nv + 1);
            for (uint32_t i = 0; i < nv; i++) lv[i] = eqbuf.fr() + ((size_t)1 << nv) + (((size_t)1 << i) - 1);
            lv[nv] = eqbuf.fr();
            TRY(launch_eq_sequence(fr_one(), point.data(), nv, lv.data(), s));
            cols.push_back(reinterpret_cast<const uint64_t*>(eqbuf.p));
            TRY(gm_sc_dense_create(0, &L.f, nv, cols.data(), reinterpret_cast<const uint64_t*>(&c0),
                                   reinterpret_cast<const uint64_t*>(&claim), &h.so, stream));
        }
        std::vector<Fr> rs;
        for (uint32_t rd = 0; rd < nv; rd++) {
            Fr co[8];
            uint32_t nc = 0;
            TRY(gm_sc_unipoly(h.so, reinterpret_cast<uint64_t*>(co), &nc));
            emit(co, nc);  // transcript.append_scalars(b"poly", &round_uni_poly.as_vec())
            Fr r;
            TRY(challenge(&r));
            rs.insert(rs.begin(), r);              // fix_var_bot
            TRY(gm_sc_bind(h.so, reinterpret_cast<const uint64_t*>(&r)));
            nrounds++;
        }
        Fr fe[GM_MAX_COLS + 1];
        uint32_t ne = 0;
        TRY(gm_sc_final_evals(h.so, reinterpret_cast<uint64_t*>(fe), &ne));
        evs.assign(fe, fe + sp.n_ins);             // final_evaluations[0..num_i]
        emit(evs.data(), evs.size());
        point = rs;
        if (h.so) { gm_sc_destroy(h.so); h.so = nullptr; }   // before its workspace goes
        big_scope.reset();
        trace[li] = Cols();                        // this layer's input is not needed again: back to the pool
    }
    if (n_msgs) *n_msgs = msgs.size();
    if (h_msgs) {
        GM_REQUIRE(msgs.size() <= msgs_cap, "message buffer too small: %zu > %llu", msgs.size(), (unsigned long long)msgs_cap);
        memcpy(h_msgs, msgs.data(), msgs.size() * sizeof(Fr));
    }
    if (n_final_point) *n_final_point = (uint32_t)point.size();
    if (h_final_point) memcpy(h_final_point, point.data(), point.size() * sizeof(Fr));
    if (h_final_evs) memcpy(h_final_evs, evs.data(), evs.size() * sizeof(Fr));
    if (cb_rc) return set_err(GM_ERR_STATE, "transcript write_scalars callback failed with %d", cb_rc);
    if (tape_used) *tape_used = pos;
    if (rounds) *rounds = nrounds;
    return GM_OK;
}

extern "C" int32_t gm_gkr_msm_prove(const uint64_t* d_points_xy, const uint8_t* d_scalar_bits, uint32_t log_num_points,
                                    uint32_t log_num_scalar_bits, const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_msgs,
                                    uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_output, uint64_t* h_final_point,
                                    uint32_t* n_final_point, uint64_t* h_final_evs, uint64_t* tape_used, uint64_t* rounds,
                                    double* witness_ms, void* stream) {
    GM_REQUIRE(h_tape, "null argument");
    return gkr_msm_prove_impl(d_points_xy, d_scalar_bits, log_num_points, log_num_scalar_bits, h_tape, n_tape, nullptr, h_msgs,
                              msgs_cap, n_msgs, h_output, h_final_point, n_final_point, h_final_evs, tape_used, rounds,
                              witness_ms, stream);
}

// gkr_msm_prove against the caller's live transcript (gen-1 TranscriptReceiver/TranscriptSender, transcript.rs:70-101)
extern "C" int32_t gm_gkr_msm_prove_tr(const uint64_t* d_points_xy, const uint8_t* d_scalar_bits, uint32_t log_num_points,
                                       uint32_t log_num_scalar_bits, const gm_transcript* tr, uint64_t* h_output,
                                       uint64_t* h_final_point, uint32_t* n_final_point, uint64_t* h_final_evs,
                                       uint64_t* n_challenges, uint64_t* rounds, void* stream) {
    GM_REQUIRE(tr && tr->challenge, "null transcript");
    return gkr_msm_prove_impl(d_points_xy, d_scalar_bits, log_num_points, log_num_scalar_bits, nullptr, 0, tr, nullptr, 0,
                              nullptr, h_output, h_final_point, n_final_point, h_final_evs, n_challenges, rounds, nullptr,
                              stream);
}
