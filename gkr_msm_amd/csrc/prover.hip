// Host side above the kernels, mirroring the reference's prover for the "image part" of the Pippenger
// protocol (the reference is Rust; no Rust toolchain exists here, so the drivers are C++ with the same
// names and the same call order):
//
//   GlueSplit::witness                       cleanup/protocols/splits.rs:172-176
//   bintree_add::builder::witness::build     cleanup/protocols/gkrs/bintree_add.rs:137-239     (+ last_step :128-135)
//   triangle_add::builder::witness::build    cleanup/protocols/gkrs/triangle_add.rs:101-158    (+ last_step :88-99)
//   PippengerEndingWG::new                   cleanup/protocols/pippenger_ending.rs:32-95  (the reference builds the
//                                            bintree witness twice, :40-45 and :67-72; one build is kept)
//   bintree / triangle protocol layer lists  bintree_add.rs:247-375, triangle_add.rs:173-232
//   SimpleGKR::prove                         cleanup/protocols/gkrs/gkr.rs:45-50
//   GenericSumcheckProtocol::prove           cleanup/protocols/sumcheck.rs:101-123
//   DenseDeg2Sumcheck / VecVecDeg2Sumcheck   sumchecks/dense_eq.rs:198-229, sumchecks/vecvec_eq.rs:424-456
//   SplitAt / ZeroCheck / GlueSplit prove    splits.rs:121-143, zero_check.rs:24-28, splits.rs:185-197
//   PippengerBucketed::prove                 pippenger_ending.rs:142-149, Pippenger::prove "prove image part" pippenger.rs:138-141
//
// The Fiat-Shamir transcript (merlin) stays with the caller: challenges are taken from a tape
// the caller provides, prover messages are returned in order.
#include <chrono>
#include <memory>
#include <vector>

#include "internal.hpp"
#include "gkr_layers.hpp"
#include "msm_plan.hpp"
#include "vecvec.hpp"

using namespace gm;

// C ABI pieces of the other translation units used here
extern "C" {
int32_t gm_vv_from_msm(const gm_msm_plan* p, const uint64_t* d_points_xy, uint32_t y_logsize, gm_vv** out, void* stream);
int32_t gm_vv_slice(const gm_vv* in, uint32_t first, uint32_t count, gm_vv** out);
int32_t gm_vv_concat(const gm_vv* a, const gm_vv* b, gm_vv** out);
int32_t gm_vv_destroy(gm_vv* v);
struct gm_sc;
int32_t gm_sc_dense_deg2_create(const gm_fn* f, uint32_t num_vars, const uint64_t* const* d_cols, const uint64_t* h_point,
                                const uint64_t* h_gamma, const uint64_t* h_claims, gm_sc** out, void* stream);
int32_t gm_sc_vecvec_deg2_create(const gm_fn* f, const gm_vv* polys, const uint64_t* h_point, const uint64_t* h_gamma,
                                 const uint64_t* h_claims, gm_sc** out, void* stream);
int32_t gm_sc_unipoly(gm_sc* so, uint64_t* h_coeffs, uint32_t* n_coeffs);
int32_t gm_sc_bind(gm_sc* so, const uint64_t* h_t);
int32_t gm_sc_final_evals(gm_sc* so, uint64_t* h_evals, uint32_t* n_evals);
int32_t gm_sc_claim(const gm_sc* so, uint64_t* h_claim);
int32_t gm_sc_destroy(gm_sc* so);
}

namespace {

#define TRY(x)                      \
    do {                            \
        int32_t rc__ = (x);         \
        if (rc__) return rc__;      \
    } while (0)

// GM_PROVE_TIMING=1: wall time of the stages of the whole-protocol drivers on stderr (development aid; each mark
// synchronises the stream)
struct StageTimer {
    const char* who;
    hipStream_t s;
    bool on;
    std::chrono::steady_clock::time_point prev;
    StageTimer(const char* w, hipStream_t st) : who(w), s(st) {
        static const bool v = [] { const char* e = getenv("GM_PROVE_TIMING"); return e && e[0] == '1'; }();
        on = v;
        if (on) { (void)hipStreamSynchronize(s); prev = std::chrono::steady_clock::now(); }
    }
    void mark(const char* name) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gm %s] %-24s %8.2f ms   (pool misses so far: %llu, %.1f MiB)\n", who, name,
                std::chrono::duration<double, std::milli>(now - prev).count(), (unsigned long long)dev_pool().n_driver_allocs,
                dev_pool().driver_alloc_bytes / 1048576.0);
        prev = now;
    }
};

struct VVHolder {
    gm_vv* v = nullptr;
    VVHolder() = default;
    explicit VVHolder(gm_vv* p) : v(p) {}
    VVHolder(const VVHolder&) = delete;
    VVHolder& operator=(const VVHolder&) = delete;
    ~VVHolder() { if (v) gm_vv_destroy(v); }
};

// one layer input ("advice", split_map_gkr.rs:75-81)
struct Advice {
    enum Kind { VECVEC, DENSE, EMPTY } kind = EMPTY;
    std::shared_ptr<VVHolder> vv;
    std::vector<std::shared_ptr<DevBuf>> cols;
    uint64_t len = 0;
    std::vector<const uint64_t*> col_ptrs() const {
        std::vector<const uint64_t*> p;
        for (auto& c : cols) p.push_back(reinterpret_cast<const uint64_t*>(c->p));
        return p;
    }
};

int32_t dense_alloc(int n, uint64_t len, std::vector<std::shared_ptr<DevBuf>>* out) {
    out->clear();
    for (int i = 0; i < n; i++) {
        out->emplace_back(new DevBuf());
        TRY(out->back()->alloc((size_t)len * sizeof(Fr)));
    }
    return GM_OK;
}

int32_t dense_map_adv(const gm_fn& f, const Advice& in, Advice* out, hipStream_t s) {
    SegPlan sp = plan_of(f);
    out->kind = Advice::DENSE;
    out->len = in.len;
    TRY(dense_alloc(sp.n_outs, in.len, &out->cols));
    std::vector<const Fr*> ci;
    std::vector<Fr*> co;
    for (auto& c : in.cols) ci.push_back(c->fr());
    for (auto& c : out->cols) co.push_back(c->fr());
    return launch_dense_map(sp, ci.data(), co.data(), in.len, s);
}

int32_t dense_map_split_adv(const gm_fn& f, const Advice& in, uint32_t lo_bit, uint32_t bundle, Advice* out, hipStream_t s) {
    SegPlan sp = plan_of(f);
    out->kind = Advice::DENSE;
    out->len = in.len / 2;
    TRY(dense_alloc(2 * sp.n_outs, in.len / 2, &out->cols));
    std::vector<const Fr*> ci;
    std::vector<Fr*> co;
    for (auto& c : in.cols) ci.push_back(c->fr());
    for (auto& c : out->cols) co.push_back(c->fr());
    return launch_dense_map_split(sp, ci.data(), co.data(), in.len, lo_bit, bundle, s);
}

int32_t adv_map(const gm_fn& f, const Advice& in, Advice* out, hipStream_t s) {
    if (in.kind == Advice::DENSE) return dense_map_adv(f, in, out, s);
    SegPlan sp = plan_of(f);
    gm_vv* o = nullptr;
    TRY(vv_map(sp, in.vv->v, &o, s));
    out->kind = Advice::VECVEC;
    out->vv.reset(new VVHolder(o));
    return GM_OK;
}

// advice_map_split (bintree_add.rs:186-205): VecVec -> dense when layer_idx + 2 == row_logsize
int32_t adv_map_split_lo0(const gm_fn& f, const Advice& in, uint32_t layer_idx, uint32_t row_logsize, uint32_t bundle,
                          Advice* out, hipStream_t s) {
    if (in.kind == Advice::DENSE) return dense_map_split_adv(f, in, 0, bundle, out, s);
    SegPlan sp = plan_of(f);
    if (layer_idx + 2 == row_logsize) {
        out->kind = Advice::DENSE;
        out->len = in.vv->v->sharded ? in.vv->v->nrows : (1ull << in.vv->v->col_logsize);
        TRY(dense_alloc(2 * sp.n_outs, out->len, &out->cols));
        std::vector<Fr*> co;
        for (auto& c : out->cols) co.push_back(c->fr());
        return vv_map_split_to_dense(sp, in.vv->v, bundle, co.data(), s);
    }
    gm_vv* o = nullptr;
    TRY(vv_map_split(sp, in.vv->v, bundle, &o, s));
    out->kind = Advice::VECVEC;
    out->vv.reset(new VVHolder(o));
    return GM_OK;
}

struct Claims {
    std::vector<Fr> point, evs;
};

struct Tape {
    const uint64_t* tape;
    uint64_t n, pos = 0;
    std::vector<Fr>* msgs;
    uint64_t rounds = 0;
    const gm_transcript* cb = nullptr;  // the caller's live transcript; replaces the tape when set
    int32_t cb_rc = 0;
    int32_t challenge(Fr* out, uint32_t bits = 128) { return challenge_vec(out, 1, bits); }
    // challenge_vec(n, bits) (proof_transcript.rs:41-45): one squeeze; tape mode: the next n tape entries
    int32_t challenge_vec(Fr* out, uint32_t cnt, uint32_t bits) {
        const int32_t rc0 = challenge_raw(out, cnt, bits);
        if (rc0) return rc0;
        for (uint32_t i = 0; i < cnt; i++) out[i] = fr_to_mont(out[i]);
        return GM_OK;
    }
    int32_t challenge_raw(Fr* out, uint32_t cnt = 1, uint32_t bits = 128) {  // canonical limbs, as drawn
        if (cb) {
            if (cb_rc) return set_err(GM_ERR_STATE, "transcript write_scalars callback failed with %d", cb_rc);
            const int32_t rc = cb->challenge(cb->ctx, cnt, bits, reinterpret_cast<uint64_t*>(out));
            if (rc) return set_err(GM_ERR_STATE, "transcript challenge callback failed with %d", rc);
        } else {
            if (pos + cnt > n) return set_err(GM_ERR_INVALID, "challenge tape exhausted after %llu challenges", (unsigned long long)pos);
            memcpy(out, tape + 4 * pos, 32 * (size_t)cnt);  // canonical values as the transcript would yield them
        }
        pos += cnt;
        return GM_OK;
    }
    void write_scalars(const std::vector<Fr>& v) {
        msgs->insert(msgs->end(), v.begin(), v.end());
        if (cb && cb->write_scalars && !cb_rc && !v.empty())
            cb_rc = cb->write_scalars(cb->ctx, reinterpret_cast<const uint64_t*>(v.data()), v.size());
    }
    std::vector<uint64_t>* points = nullptr;  // G1 points written (affine wire form, 12 x u64 each), when the driver has any
    void write_points(const uint64_t* aff, uint64_t n) {
        if (points) points->insert(points->end(), aff, aff + 12 * n);
        if (cb && cb->write_points && !cb_rc && n) cb_rc = cb->write_points(cb->ctx, aff, n);
    }
};

// GM_PROVE_TIMING=2: where the host's wall time of the layer provers goes (development aid)
struct LayerClock {
    double create = 0, unipoly = 0, bind = 0, finals = 0, other = 0;
    uint64_t layers = 0, rounds = 0;
    static bool on() {
        static const bool v = [] { const char* e = getenv("GM_PROVE_TIMING"); return e && e[0] == '2'; }();
        return v;
    }
    static LayerClock& get() { static thread_local LayerClock c; return c; }
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void dump(const char* who) {
        if (!on()) return;
        fprintf(stderr, "[gm %s] %llu layers, %llu rounds: create %.1f ms, unipoly (wait for sums) %.1f ms, bind %.1f ms, finals %.1f ms\n", who,
                (unsigned long long)layers, (unsigned long long)rounds, create / 1e3, unipoly / 1e3, bind / 1e3, finals / 1e3);
        *this = LayerClock();
    }
};

// Layers of a sharded proof that every rank would prove alike (the bucket-reduction GKR runs on the gathered bucket sums): ONE rank
// -- the leader, rank 0 -- runs their sumcheck objects, the others take every round polynomial and the final evaluations from the
// exchange (they contribute zeros to the field sum) and only drive their transcript.  Same messages everywhere; 7 of 8 GPUs do not
// spend ~300 latency-bound rounds per proof on work whose result they can read.
inline Shard& current_lead() {
    static thread_local Shard s;
    return s;
}
struct LeadScope {
    Shard prev;
    explicit LeadScope(const Shard& s) : prev(current_lead()) { current_lead() = s; }
    ~LeadScope() { current_lead() = prev; }
};
inline bool lead_follower() { return current_lead().comm && current_lead().rank != 0; }

// GenericSumcheckProtocol::prove (sumcheck.rs:101-123); so == nullptr on a follower (see LeadScope), n_finals = the evaluations it expects
int32_t generic_sumcheck_prove(Tape* tr, gm_sc* so, uint32_t num_rounds, uint32_t degree, std::vector<Fr>* point,
                               std::vector<Fr>* final_evals, uint32_t n_finals = 0) {
    std::vector<Fr> r;
    const bool clk = LayerClock::on();
    LayerClock& lc = LayerClock::get();
    const Shard ld = current_lead();
    const bool follower = ld.comm && ld.rank != 0;
    if (follower != (so == nullptr)) return set_err(GM_ERR_STATE, "leader / follower mismatch");
    for (uint32_t rd = 0; rd < num_rounds; rd++) {
        Fr coeffs[8];
        uint32_t nc = degree + 1;
        const double t0 = clk ? LayerClock::now() : 0;
        if (!follower) TRY(gm_sc_unipoly(so, reinterpret_cast<uint64_t*>(coeffs), &nc));
        else for (uint32_t i = 0; i < nc; i++) coeffs[i] = fr_zero();
        if (nc != degree + 1) return set_err(GM_ERR_STATE, "round polynomial has %u coefficients, expected %u", nc, degree + 1);
        if (ld.comm) TRY(shard_sum_fr(ld, coeffs, (int)nc));
        if (clk) { lc.unipoly += LayerClock::now() - t0; lc.rounds++; }
        std::vector<Fr> msg;  // compress_coefficients: drop the linear term (sumcheck.rs:27-31)
        msg.push_back(coeffs[0]);
        for (uint32_t i = 2; i < nc; i++) msg.push_back(coeffs[i]);
        tr->write_scalars(msg);
        Fr x;
        TRY(tr->challenge(&x));
        r.push_back(x);
        const double t1 = clk ? LayerClock::now() : 0;
        if (!follower) TRY(gm_sc_bind(so, reinterpret_cast<const uint64_t*>(&x)));
        if (clk) lc.bind += LayerClock::now() - t1;
        tr->rounds++;
    }
    point->assign(r.rbegin(), r.rend());
    Fr ev[GM_MAX_COLS + 1];
    uint32_t ne = n_finals;
    const double t2 = clk ? LayerClock::now() : 0;
    if (!follower) TRY(gm_sc_final_evals(so, reinterpret_cast<uint64_t*>(ev), &ne));
    else for (uint32_t i = 0; i < ne; i++) ev[i] = fr_zero();
    if (ld.comm) {
        if (n_finals && ne != n_finals) return set_err(GM_ERR_STATE, "leader has %u final evaluations, the followers expect %u", ne, n_finals);
        TRY(shard_sum_fr(ld, ev, (int)ne));
    }
    if (clk) lc.finals += LayerClock::now() - t2;
    final_evals->assign(ev, ev + ne);
    return GM_OK;
}

struct ScHolder {
    gm_sc* so = nullptr;
    ~ScHolder() { if (so) gm_sc_destroy(so); }
};

// DenseDeg2Sumcheck::prove (dense_eq.rs:198-229)
int32_t dense_deg2_prove(Tape* tr, const gm_fn& f, uint32_t num_vars, Claims* claims, const Advice& adv, hipStream_t s) {
    if (adv.kind != Advice::DENSE) return set_err(GM_ERR_STATE, "dense layer got a non-dense advice");
    Fr gamma;
    TRY(tr->challenge(&gamma));
    ScHolder h;
    auto ptrs = adv.col_ptrs();
    const double t0 = LayerClock::on() ? LayerClock::now() : 0;
    if (!lead_follower())
        TRY(gm_sc_dense_deg2_create(&f, num_vars, ptrs.data(), reinterpret_cast<const uint64_t*>(claims->point.data()),
                                    reinterpret_cast<const uint64_t*>(&gamma), reinterpret_cast<const uint64_t*>(claims->evs.data()),
                                    &h.so, s));
    if (LayerClock::on()) { LayerClock::get().create += LayerClock::now() - t0; LayerClock::get().layers++; }
    std::vector<Fr> pt, evs;
    TRY(generic_sumcheck_prove(tr, h.so, num_vars, 3, &pt, &evs, (uint32_t)plan_of(f).n_ins));
    tr->write_scalars(evs);
    claims->point = pt;
    claims->evs = evs;
    return GM_OK;
}

// VecVecDeg2Sumcheck::prove (vecvec_eq.rs:424-456)
int32_t vecvec_deg2_prove(Tape* tr, const gm_fn& f, uint32_t num_vars, Claims* claims, const Advice& adv, hipStream_t s) {
    if (adv.kind != Advice::VECVEC) return set_err(GM_ERR_STATE, "vecvec layer got a non-vecvec advice");
    Fr gamma;
    TRY(tr->challenge(&gamma));
    ScHolder h;
    const double t0 = LayerClock::on() ? LayerClock::now() : 0;
    TRY(gm_sc_vecvec_deg2_create(&f, adv.vv->v, reinterpret_cast<const uint64_t*>(claims->point.data()),
                                 reinterpret_cast<const uint64_t*>(&gamma), reinterpret_cast<const uint64_t*>(claims->evs.data()),
                                 &h.so, s));
    if (LayerClock::on()) { LayerClock::get().create += LayerClock::now() - t0; LayerClock::get().layers++; }
    std::vector<Fr> pt, evs;
    TRY(generic_sumcheck_prove(tr, h.so, num_vars, 3, &pt, &evs));
    evs.pop_back();  // the eq column (vecvec_eq.rs:451)
    tr->write_scalars(evs);
    claims->point = pt;
    claims->evs = evs;
    return GM_OK;
}

// SplitAt::prove (splits.rs:121-143); HI(x): insert at x, LO(x): insert at len - x
int32_t split_at_prove(Tape* tr, Claims* c, bool hi, uint32_t idx, uint32_t bundle) {
    Fr r;
    TRY(tr->challenge(&r));
    std::vector<Fr> l, rr;
    for (size_t base = 0; base < c->evs.size(); base += bundle) {
        std::vector<Fr>& dst = ((base / bundle) % 2 == 0) ? l : rr;
        for (size_t i = base; i < base + bundle && i < c->evs.size(); i++) dst.push_back(c->evs[i]);
    }
    std::vector<Fr> nw;
    for (size_t i = 0; i < l.size() && i < rr.size(); i++) nw.push_back(fr_add(l[i], fr_mul(r, fr_sub(rr[i], l[i]))));
    const size_t pos = hi ? idx : c->point.size() - idx;
    c->point.insert(c->point.begin() + pos, r);
    c->evs = nw;
    return GM_OK;
}

// SimpleGKR::prove (gkr.rs:45-50)
int32_t simple_gkr_prove(Tape* tr, const std::vector<Layer>& layers, const std::vector<Advice>& advices, Claims* claims,
                         Arena* arena, hipStream_t s) {
    if (layers.size() != advices.size()) return set_err(GM_ERR_STATE, "%zu layers vs %zu advices (zip_eq)", layers.size(), advices.size());
    for (size_t k = layers.size(); k-- > 0;) {
        const Layer& L = layers[k];
        const Advice& a = advices[k];
        // every buffer of this layer's sumcheck object comes out of the arena; the object is gone when the layer
        // returns (its kernels are complete: final_evals synchronised), so the arena can be rewound
        arena->reset();
        ArenaScope scope(arena);
        switch (L.kind) {
            case Layer::VECVEC: TRY(vecvec_deg2_prove(tr, L.f, L.num_vars, claims, a, s)); break;
            case Layer::DENSE: TRY(dense_deg2_prove(tr, L.f, L.num_vars, claims, a, s)); break;
            case Layer::SPLIT: TRY(split_at_prove(tr, claims, L.split_hi, L.split_idx, L.bundle)); break;
            case Layer::ZEROCHECK:  // zero_check.rs:24-28
                claims->evs.push_back(fr_zero());
                claims->evs.push_back(fr_zero());
                break;
        }
    }
    return GM_OK;
}

}  // namespace

struct gm_pip_witness {
    uint32_t x_log, y_log, d_log;
    hipStream_t stream;
    Shard sh;               // sharded witness: this rank's windows only
    Arena arena;            // per-layer workspace of the sumcheck objects (reset after every layer)
    Fr* pinned = nullptr;   // host staging for the per-round results
    ~gm_pip_witness() { if (pinned) (void)hipHostFree(pinned); }
    std::vector<Advice> bintree_advices, triangle_advices;
    Advice bucket_sums;   // bintree last_step
    Advice dense_output;  // triangle last_step: 3*(d+1) columns of 2^y_log
};

// bintree_add::builder::witness::build (bintree_add.rs:137-184)
static int32_t bintree_witness_build(Advice advice, uint32_t row_logsize, uint32_t num_adds, bool do_bitcheck,
                                     std::vector<Advice>* advices, hipStream_t s) {
    GM_REQUIRE(num_adds > 0, "num_adds must be positive (bintree_add.rs:143)");
    for (uint32_t add = 0; add < num_adds; add++) {
        const bool last = add + 1 == num_adds;
        for (int step = 0; step < 3; step++) {
            Advice next;
            bool have_next = true;
            if (step == 0) TRY(adv_map(mkfn(add == 0 ? GM_FN_AFF_L1 : GM_FN_PROJ_L1, 1), advice, &next, s));
            else if (step == 1) TRY(adv_map(mkfn(add == 0 ? GM_FN_AFF_L2 : GM_FN_PROJ_L2, 1), advice, &next, s));
            else if (last) have_next = false;
            else TRY(adv_map_split_lo0(mkfn(add == 0 ? GM_FN_AFF_L3 : GM_FN_PROJ_L3, 1), advice, add, row_logsize, 3, &next, s));
            advices->push_back(advice);
            if (add == 0 && step == 0 && do_bitcheck) advices->push_back(Advice());
            if (have_next) advice = next;
        }
        if (!last) advices->push_back(Advice());
    }
    return GM_OK;
}

// triangle_add::builder::witness::build (triangle_add.rs:101-158); split index HI(hi_idx)
static int32_t triangle_witness_build(Advice advice, uint32_t num_vars, uint32_t hi_idx, std::vector<Advice>* advices,
                                      hipStream_t s) {
    const uint32_t num_layers = num_vars - hi_idx;
    for (uint32_t l = 0; l <= num_layers; l++) {
        for (int step = 0; step < 3; step++) {
            Advice next;
            bool have_next = true;
            if (step == 0) TRY(dense_map_adv(mkfn(GM_FN_TRI_L1, 1, GM_FN_PROJ_L1, (int)l), advice, &next, s));
            else if (step == 1) TRY(dense_map_adv(mkfn(GM_FN_PROJ_L2, (int)l + 3), advice, &next, s));
            else if (l == num_layers) have_next = false;
            else {
                // the arrays at layer l have num_vars - l variables; HI(hi_idx) = index bit (num_vars - l) - 1 - hi_idx
                const uint32_t lo_bit = (num_vars - l) - 1 - hi_idx;
                TRY(dense_map_split_adv(mkfn(GM_FN_PROJ_L3, (int)l + 3), advice, lo_bit, 3, &next, s));
            }
            advices->push_back(advice);
            if (have_next) advice = next;
        }
        if (l < num_layers) advices->push_back(Advice());
    }
    return GM_OK;
}

// ------------------------------------------------------------------------------------------- C ABI
// PippengerWG::new without the G1 commitments (pippenger.rs:37-70): image -> GlueSplit::witness -> PippengerEndingWG::new.
// comm != nullptr: the plan covers this rank's windows; everything up to the bucket sums is local to its bucket rows,
// the bucket sums are exchanged once and the bucket-reduction (triangle) witness is built replicated.
static int32_t pip_witness_create(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize, const gm_comm* comm,
                                  gm_pip_witness** out, void* stream) {
    GM_REQUIRE(plan && d_points_xy && out, "null argument");
    GM_REQUIRE(plan->x_log >= plan->d_log, "x_logsize >= d_logsize required (pippenger.rs:93)");
    GM_REQUIRE(plan->x_log >= 2, "x_logsize >= 2 required");
    hipStream_t s = as_stream(stream);
    std::unique_ptr<gm_pip_witness> w(new gm_pip_witness());
    w->x_log = plan->x_log; w->y_log = y_logsize; w->d_log = plan->d_log; w->stream = s;
    if (comm && comm->world > 1) {
        GM_REQUIRE(comm->all_gather && comm->rank < comm->world && (comm->world & (comm->world - 1)) == 0, "bad gm_comm");
        GM_REQUIRE(plan->y_size == (1u << y_logsize), "the sharded prover needs y_size = 2^y_logsize");
        GM_REQUIRE(plan->y_size % comm->world == 0 && plan->nwin == plan->y_size / comm->world &&
                       plan->y0 == comm->rank * plan->nwin,
                   "rank %u of %u must own windows [%u, %u)", comm->rank, comm->world, comm->rank * (plan->y_size / comm->world),
                   (comm->rank + 1) * (plan->y_size / comm->world));
        w->sh.comm = comm; w->sh.rank = comm->rank; w->sh.world = comm->world;
        while ((1u << w->sh.lg) < comm->world) w->sh.lg++;
    } else {
        GM_REQUIRE(plan->y0 == 0 && plan->y1 == plan->y_size, "the image needs a plan over all windows");
    }
    const bool sharded = w->sh.comm != nullptr;
    gm_vv* image = nullptr;
    TRY(vv_from_msm(plan, d_points_xy, y_logsize, sharded, &image, stream));
    VVHolder img(image);
    if (sharded) {  // EQPolyData::new looks at the longest row of the whole polynomial (vecvec.rs:86)
        std::vector<char> all;
        const uint32_t mine = image->max_row_len;
        TRY(shard_all_gather(w->sh, &mine, sizeof(uint32_t), &all));
        for (uint32_t r = 0; r < w->sh.world; r++) {
            uint32_t v;
            memcpy(&v, all.data() + 4 * (size_t)r, 4);
            if (v > image->max_row_len) image->max_row_len = v;
        }
    }
    // GlueSplit::witness (splits.rs:172-176)
    gm_vv *xy = nullptr, *z = nullptr, *xy_s = nullptr, *z_s = nullptr, *glued = nullptr;
    TRY(gm_vv_slice(image, 0, 2, &xy));
    VVHolder hxy(xy);
    TRY(gm_vv_slice(image, 2, 1, &z));
    VVHolder hz(z);
    TRY(vv_map_split(plan_of(mkfn(GM_FN_ID, 2)), xy, 2, &xy_s, s));
    VVHolder hxys(xy_s);
    TRY(vv_map_split(plan_of(mkfn(GM_FN_ID, 1)), z, 1, &z_s, s));
    VVHolder hzs(z_s);
    TRY(gm_vv_concat(xy_s, z_s, &glued));
    Advice in;
    in.kind = Advice::VECVEC;
    in.vv.reset(new VVHolder(glued));
    // PippengerEndingWG::new (pippenger_ending.rs:32-95)
    const uint32_t horizontal = plan->x_log, multirow = y_logsize, bucket = plan->d_log;
    TRY(bintree_witness_build(in, horizontal, horizontal, true, &w->bintree_advices, s));
    // last_step (bintree_add.rs:128-135)
    Advice local_sums;
    TRY(adv_map(mkfn(horizontal - 1 == 0 ? GM_FN_AFF_L3 : GM_FN_PROJ_L3, 1), w->bintree_advices.back(), &local_sums, s));
    GM_REQUIRE(local_sums.kind == Advice::DENSE, "bucket sums are not dense");
    const uint32_t nv = multirow + bucket;
    if (sharded) {
        // every rank gets all bucket sums: 3 columns of 2^(y_log + d) elements (0.75 MiB at config B)
        const uint64_t L = local_sums.len, full = (uint64_t)1 << nv;
        GM_REQUIRE(L * w->sh.world == full, "bucket sum slices do not tile the rows");
        std::vector<Fr> mine(3 * L);
        for (int c = 0; c < 3; c++)
            GM_HIP(hipMemcpyAsync(mine.data() + c * L, local_sums.cols[c]->p, L * sizeof(Fr), hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
        std::vector<char> all;
        TRY(shard_all_gather(w->sh, mine.data(), 3 * L * sizeof(Fr), &all));
        const Fr* a = reinterpret_cast<const Fr*>(all.data());
        w->bucket_sums.kind = Advice::DENSE;
        w->bucket_sums.len = full;
        TRY(dense_alloc(3, full, &w->bucket_sums.cols));
        std::vector<Fr> col(full);
        for (int c = 0; c < 3; c++) {
            for (uint32_t r = 0; r < w->sh.world; r++) memcpy(col.data() + r * L, a + ((size_t)r * 3 + c) * L, L * sizeof(Fr));
            GM_HIP(hipMemcpyAsync(w->bucket_sums.cols[c]->p, col.data(), full * sizeof(Fr), hipMemcpyHostToDevice, s));
            GM_HIP(hipStreamSynchronize(s));  // col is reused
        }
    } else {
        w->bucket_sums = local_sums;
    }
    Advice s1, s2;
    TRY(dense_map_split_adv(mkfn(GM_FN_ID, 3), w->bucket_sums, nv - 1 - multirow, 3, &s1, s));
    TRY(dense_map_split_adv(mkfn(GM_FN_ID, 6), s1, (nv - 1) - 1 - multirow, 3, &s2, s));
    TRY(triangle_witness_build(s2, nv - 2, multirow, &w->triangle_advices, s));
    // pippenger.rs:531-534: last_step(ending.last(), num_layers) with num_layers = d - 2
    TRY(dense_map_adv(mkfn(GM_FN_PROJ_L3, (int)(bucket - 2) + 3), w->triangle_advices.back(), &w->dense_output, s));
    // workspace for the largest layer (bintree level 0: 6 polys over the glue-split image):
    // fold buffers of 1/2 and 1/4 of the cells per polynomial + tables
    {
        const uint64_t T = glued->total, nr = glued->nrows;
        const size_t bytes = (size_t)6 * 32 * (T / 2 + T / 4 + 4 * nr + 64) + ((size_t)48 << 20) + ((size_t)96 << nv);
        TRY(w->arena.init(bytes));
        GM_HIP(hipHostMalloc((void**)&w->pinned, 16 * sizeof(Fr), hipHostMallocCoherent | hipHostMallocMapped));
        memset(w->pinned, 0, 16 * sizeof(Fr));
    }
    GM_HIP(hipStreamSynchronize(s));
    *out = w.release();
    return GM_OK;
}

extern "C" int32_t gm_pip_witness_create(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                         gm_pip_witness** out, void* stream) {
    return pip_witness_create(plan, d_points_xy, y_logsize, nullptr, out, stream);
}

extern "C" int32_t gm_pip_witness_create_sharded(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                                 const gm_comm* comm, gm_pip_witness** out, void* stream) {
    GM_REQUIRE(comm, "null gm_comm");
    return pip_witness_create(plan, d_points_xy, y_logsize, comm, out, stream);
}

extern "C" int32_t gm_comm_sum_fr(const gm_comm* comm, uint64_t* h_vals, uint32_t n) {
    GM_REQUIRE(comm && comm->all_gather && h_vals && comm->rank < comm->world, "bad argument");
    Shard sh;
    sh.comm = comm; sh.rank = comm->rank; sh.world = comm->world;
    return shard_sum_fr(sh, reinterpret_cast<Fr*>(h_vals), (int)n);
}

// Where the wall time of the calling thread's sharded calls went since the last reset (ShardClock, internal.hpp):
// out8 = {small all-gathers: microseconds, count; bulk all-gathers: microseconds, count, bytes; pull_dev: microseconds, count, bytes}
extern "C" int32_t gm_shard_clock(int32_t reset, double* out8) {
    ShardClock& c = ShardClock::get();
    if (out8) {
        out8[0] = c.small_us; out8[1] = (double)c.small_n; out8[2] = c.bulk_us; out8[3] = (double)c.bulk_n; out8[4] = (double)c.bulk_bytes;
        out8[5] = c.pull_us; out8[6] = (double)c.pull_n; out8[7] = (double)c.pull_bytes;
    }
    if (reset) c = ShardClock();
    return GM_OK;
}

extern "C" int32_t gm_pip_witness_destroy(gm_pip_witness* w) {
    delete w;
    return GM_OK;
}

// device columns of the dense output (3*(d+1) columns of 2^y_logsize) and of the bucket sums (3 x 2^(y_log+d))
extern "C" int32_t gm_pip_witness_outputs(const gm_pip_witness* w, const uint64_t** d_output_cols, uint32_t* n_output_cols,
                                          uint64_t* output_len, const uint64_t** d_bucket_sum_cols) {
    GM_REQUIRE(w, "null witness");
    if (d_output_cols)
        for (size_t i = 0; i < w->dense_output.cols.size(); i++)
            d_output_cols[i] = reinterpret_cast<const uint64_t*>(w->dense_output.cols[i]->p);
    if (n_output_cols) *n_output_cols = (uint32_t)w->dense_output.cols.size();
    if (output_len) *output_len = w->dense_output.len;
    if (d_bucket_sum_cols)
        for (int i = 0; i < 3; i++) d_bucket_sum_cols[i] = reinterpret_cast<const uint64_t*>(w->bucket_sums.cols[i]->p);
    return GM_OK;
}

// "claim computation" of run_pippenger (pippenger.rs:531-541): evaluate_poly(output, r) for every column of the dense output
extern "C" int32_t gm_pip_witness_claims(const gm_pip_witness* w, const uint64_t* h_point, uint64_t* h_evs, uint32_t* n_evs) {
    GM_REQUIRE(w && h_point && h_evs, "null argument");
    const uint64_t len = w->dense_output.len;
    GM_REQUIRE(len == (1ull << w->y_log), "dense output length is not 2^y_logsize");
    std::vector<Fr> r(w->y_log), col(len);
    memcpy(r.data(), h_point, r.size() * sizeof(Fr));
    Fr* out = reinterpret_cast<Fr*>(h_evs);
    for (size_t c = 0; c < w->dense_output.cols.size(); c++) {
        GM_HIP(hipMemcpyAsync(col.data(), w->dense_output.cols[c]->p, len * sizeof(Fr), hipMemcpyDeviceToHost, w->stream));
        GM_HIP(hipStreamSynchronize(w->stream));
        std::vector<Fr> cur = col;   // r[0] is the most significant variable (cleanup/utils/arith.rs:6-9)
        for (size_t k = r.size(); k-- > 0;) {
            for (size_t i = 0; i < cur.size() / 2; i++) cur[i] = fr_add(cur[2 * i], fr_mul(r[k], fr_sub(cur[2 * i + 1], cur[2 * i])));
            cur.resize(cur.size() / 2);
        }
        out[c] = cur[0];
    }
    if (n_evs) *n_evs = (uint32_t)w->dense_output.cols.size();
    return GM_OK;
}

// bytes of witness trace held on the device
extern "C" uint64_t gm_pip_witness_bytes(const gm_pip_witness* w) {
    if (!w) return 0;
    uint64_t b = 0;
    auto add = [&](const Advice& a) {
        if (a.kind == Advice::DENSE) for (auto& c : a.cols) b += c->bytes;
        if (a.kind == Advice::VECVEC) for (auto& c : a.vv->v->cols) b += c->bytes;
    };
    for (auto& a : w->bintree_advices) add(a);
    for (auto& a : w->triangle_advices) add(a);
    return b;
}

// PippengerBucketed::prove + GlueSplit::prove = "prove image part" (pippenger.rs:138-141).
//   h_claim_point: y_logsize elements; h_claim_evs: 3*(d+1) evaluations of the dense output at that point
//   h_tape: n_tape challenges, canonical 4 x u64 (values < 2^128)
//   outputs: prover messages in order (h_msgs, capacity msgs_cap elements), final claims (point of
//   y_log + d + x_log elements, 3 evaluations: x, y, z of the image), challenges consumed, sumcheck rounds run.
// PippengerBucketed::prove + GlueSplit::prove on an open transcript; c: claims in (r_y, dense-output evs) -> out (point, x/y/z evs)
static int32_t image_part_core(const gm_pip_witness* w, Tape* trp, Claims* cp) {
    Tape& tr = *trp;
    Claims& c = *cp;
    const uint32_t multirow = w->y_log, bucket = w->d_log, horizontal = w->x_log;
    hipStream_t s = w->stream;
    // PippengerBucketed::prove (pippenger_ending.rs:142-149)
    gm_pip_witness* wm = const_cast<gm_pip_witness*>(w);
    shared_pinned() = wm->pinned;
    struct PinnedReset { ~PinnedReset() { shared_pinned() = nullptr; } } pinned_reset;
    {   // sharded: the bucket reduction runs on the gathered bucket sums -- on the leader only (LeadScope)
        static const bool no_lead = [] { const char* e = getenv("GM_SHARD_NO_LEADER"); return e && e[0] == '1'; }();   // A/B: every rank proves it
        LeadScope lead((w->sh.comm && !no_lead) ? w->sh : Shard());
        TRY(simple_gkr_prove(&tr, triangle_layers(multirow + bucket - 2, multirow), w->triangle_advices, &c, &wm->arena, s));
    }
    TRY(split_at_prove(&tr, &c, true, multirow, 3));
    TRY(split_at_prove(&tr, &c, true, multirow, 3));
    {   // the bucket-sum tree runs over this rank's rows only when the witness is sharded
        ShardScope scope(w->sh);
        TRY(simple_gkr_prove(&tr, bintree_layers(multirow + bucket + horizontal, horizontal, horizontal, true),
                             w->bintree_advices, &c, &wm->arena, s));
    }
    // GlueSplit::prove (splits.rs:185-197)
    {
        Fr r;
        TRY(tr.challenge(&r));
        GM_REQUIRE(c.evs.size() == 6, "GlueSplit expects 6 evaluations, got %zu", c.evs.size());
        std::vector<Fr> nw = {fr_add(c.evs[0], fr_mul(r, fr_sub(c.evs[2], c.evs[0]))),
                              fr_add(c.evs[1], fr_mul(r, fr_sub(c.evs[3], c.evs[1]))),
                              fr_add(c.evs[4], fr_mul(r, fr_sub(c.evs[5], c.evs[4])))};
        c.point.push_back(r);
        c.evs = nw;
    }
    return GM_OK;
}

static int32_t prove_image_part(const gm_pip_witness* w, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                const uint64_t* h_tape, uint64_t n_tape, const gm_transcript* cb, uint64_t* h_msgs,
                                uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_final_point, uint32_t* n_final_point,
                                uint64_t* h_final_evs, uint64_t* tape_used, uint64_t* rounds) {
    const uint32_t multirow = w->y_log, bucket = w->d_log;
    std::vector<Fr> msgs;
    Tape tr{h_tape, n_tape, 0, &msgs, 0, cb, 0};
    Claims c;
    c.point.resize(multirow);
    memcpy(c.point.data(), h_claim_point, 32 * (size_t)multirow);
    c.evs.resize(3 * (bucket + 1));
    memcpy(c.evs.data(), h_claim_evs, 32 * c.evs.size());
    const double t_all = LayerClock::on() ? LayerClock::now() : 0;
    TRY(image_part_core(w, &tr, &c));
    if (LayerClock::on()) {
        fprintf(stderr, "[gm image part] total %.1f ms\n", (LayerClock::now() - t_all) / 1e3);
        LayerClock::get().dump("image part");
    }
    if (n_msgs) *n_msgs = msgs.size();
    if (h_msgs) {
        GM_REQUIRE(msgs.size() <= msgs_cap, "message buffer too small: %zu > %llu", msgs.size(), (unsigned long long)msgs_cap);
        memcpy(h_msgs, msgs.data(), msgs.size() * sizeof(Fr));
    }
    if (n_final_point) *n_final_point = (uint32_t)c.point.size();
    if (h_final_point) memcpy(h_final_point, c.point.data(), c.point.size() * sizeof(Fr));
    if (h_final_evs) memcpy(h_final_evs, c.evs.data(), c.evs.size() * sizeof(Fr));
    if (tr.cb_rc) return set_err(GM_ERR_STATE, "transcript write_scalars callback failed with %d", tr.cb_rc);
    if (tape_used) *tape_used = tr.pos;
    if (rounds) *rounds = tr.rounds;
    return GM_OK;
}

extern "C" int32_t gm_pip_prove_image_part(const gm_pip_witness* w, const uint64_t* h_claim_point,
                                           const uint64_t* h_claim_evs, const uint64_t* h_tape, uint64_t n_tape,
                                           uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_final_point,
                                           uint32_t* n_final_point, uint64_t* h_final_evs, uint64_t* tape_used,
                                           uint64_t* rounds) {
    GM_REQUIRE(w && h_claim_point && h_claim_evs && h_tape, "null argument");
    return prove_image_part(w, h_claim_point, h_claim_evs, h_tape, n_tape, nullptr, h_msgs, msgs_cap, n_msgs, h_final_point,
                            n_final_point, h_final_evs, tape_used, rounds);
}

// Same prover driven by the caller's live Fiat-Shamir transcript (ProofTranscript2, proof_transcript.rs:85-147): every
// write_scalars of the reference reaches tr->write_scalars in order, every challenge(128) is drawn through tr->challenge.
extern "C" int32_t gm_pip_prove_image_part_tr(const gm_pip_witness* w, const uint64_t* h_claim_point,
                                              const uint64_t* h_claim_evs, const gm_transcript* tr, uint64_t* h_final_point,
                                              uint32_t* n_final_point, uint64_t* h_final_evs, uint64_t* n_challenges,
                                              uint64_t* rounds) {
    GM_REQUIRE(w && h_claim_point && h_claim_evs && tr && tr->challenge, "null argument");
    return prove_image_part(w, h_claim_point, h_claim_evs, nullptr, 0, tr, nullptr, 0, nullptr, h_final_point, n_final_point,
                            h_final_evs, n_challenges, rounds);
}

// =================================================================================================================
// The two GKR circuits as protocols of their own: TriangleAddWG / TriangleAdd (gkrs/triangle_add.rs:160-250) and
// VecVecBintreeAddWG / VecVecBintreeAdd (gkrs/bintree_add.rs:85-126, 377-...).  The reference's tests and benches drive them
// directly (triangle_add.rs:277-393, bintree_add.rs:401-505); inside the Pippenger prover they are reached through
// gm_pip_witness.  Same builders, same layer lists, same SimpleGKR loop as above.
struct gm_gkr_witness {
    hipStream_t stream = nullptr;
    Arena arena;
    Fr* pinned = nullptr;
    std::vector<Advice> advices;
    std::vector<Layer> layers;
    Advice output;                      // last_step of the circuit, dense columns
    uint32_t out_vars = 0, n_claims = 0;
    ~gm_gkr_witness() { if (pinned) (void)hipHostFree(pinned); }
};

static int32_t gkr_witness_finish(gm_gkr_witness* w, size_t arena_bytes) {
    TRY(w->arena.init(arena_bytes));
    GM_HIP(hipHostMalloc((void**)&w->pinned, 16 * sizeof(Fr), hipHostMallocCoherent | hipHostMallocMapped));
    memset(w->pinned, 0, 16 * sizeof(Fr));
    GM_HIP(hipStreamSynchronize(w->stream));
    return GM_OK;
}

// TriangleAddWG::new(inputs, num_vars, split_var = HI(split_hi)) (triangle_add.rs:160-171) + last_step (:88-99).
// d_cols: the 12 columns (a, b, c, d) x (X, Y, Z) of 2^num_vars elements each, as produced by two HI splits.
extern "C" int32_t gm_triangle_witness_create(const uint64_t* const* d_cols, uint32_t num_vars, uint32_t split_hi,
                                              gm_gkr_witness** out, void* stream) {
    GM_REQUIRE(d_cols && out, "null argument");
    GM_REQUIRE(split_hi <= num_vars && num_vars < 40, "split index HI(%u) outside %u variables", split_hi, num_vars);
    std::unique_ptr<gm_gkr_witness> w(new gm_gkr_witness());
    hipStream_t s = w->stream = as_stream(stream);
    Advice in;
    in.kind = Advice::DENSE;
    in.len = (uint64_t)1 << num_vars;
    for (int i = 0; i < 12; i++) {
        in.cols.emplace_back(new DevBuf());
        in.cols.back()->p = const_cast<uint64_t*>(d_cols[i]);  // borrowed: the caller keeps the inputs alive
        in.cols.back()->owned = false;
        in.cols.back()->bytes = (size_t)in.len * sizeof(Fr);
    }
    TRY(triangle_witness_build(in, num_vars, split_hi, &w->advices, s));
    const uint32_t num_layers = num_vars - split_hi;
    TRY(dense_map_adv(mkfn(GM_FN_PROJ_L3, (int)num_layers + 3), w->advices.back(), &w->output, s));
    w->layers = triangle_layers(num_vars, split_hi);
    w->out_vars = split_hi;
    w->n_claims = 3 * (num_layers + 3);
    TRY(gkr_witness_finish(w.get(), ((size_t)12 * 96 << num_vars) + ((size_t)48 << 20)));
    *out = w.release();
    return GM_OK;
}

// VecVecBintreeAddWG::new_common(VecVecMAP(inputs), row_logsize, num_adds, do_bitcheck) (bintree_add.rs:98-125) + last_step.
// inputs: 4 polynomials (x_even, y_even, x_odd, y_odd), or 6 with (z_even, z_odd) appended when do_bitcheck.
extern "C" int32_t gm_bintree_witness_create(const gm_vv* inputs, uint32_t num_adds, int32_t do_bitcheck, gm_gkr_witness** out,
                                             void* stream) {
    GM_REQUIRE(inputs && out, "null argument");
    GM_REQUIRE(inputs->k == (do_bitcheck ? 6u : 4u), "%u input polynomials, expected %u", inputs->k, do_bitcheck ? 6u : 4u);
    GM_REQUIRE(num_adds >= 1 && num_adds <= inputs->row_logsize + inputs->col_logsize + 1, "bad num_adds %u", num_adds);
    GM_REQUIRE(!inputs->sharded, "standalone circuit over a sharded polynomial");
    std::unique_ptr<gm_gkr_witness> w(new gm_gkr_witness());
    hipStream_t s = w->stream = as_stream(stream);
    gm_vv* share = nullptr;
    TRY(gm_vv_slice(inputs, 0, inputs->k, &share));  // shares the caller's columns
    Advice in;
    in.kind = Advice::VECVEC;
    in.vv.reset(new VVHolder(share));
    // the inputs are the LO(0) split of the point columns: they carry row_logsize - 1 horizontal variables, the circuit is
    // parametrised by the row_logsize BEFORE the split (bintree_add.rs:411-413)
    const uint32_t row_logsize = inputs->row_logsize + 1, num_vars = row_logsize + inputs->col_logsize;
    TRY(bintree_witness_build(in, row_logsize, num_adds, do_bitcheck != 0, &w->advices, s));
    Advice last;
    TRY(adv_map(mkfn(num_adds - 1 == 0 ? GM_FN_AFF_L3 : GM_FN_PROJ_L3, 1), w->advices.back(), &last, s));
    w->out_vars = num_vars - num_adds;
    if (last.kind == Advice::VECVEC) {  // densify (the reference's tests call to_dense on it)
        w->output.kind = Advice::DENSE;
        w->output.len = (uint64_t)1 << w->out_vars;
        TRY(dense_alloc(3, w->output.len, &w->output.cols));
        std::vector<uint64_t*> co;
        for (auto& c : w->output.cols) co.push_back(reinterpret_cast<uint64_t*>(c->p));
        TRY(gm_vv_to_dense(last.vv->v, co.data(), s));
    } else {
        w->output = last;
    }
    w->layers = bintree_layers(num_vars, num_adds, row_logsize, do_bitcheck != 0);
    w->n_claims = 3;
    const uint64_t T = inputs->total, nr = inputs->nrows;
    const size_t bytes = (size_t)6 * 32 * (T + 4 * nr + 64) + ((size_t)48 << 20) + ((size_t)6 * 96 << (num_vars > 0 ? num_vars - 1 : 0));
    TRY(gkr_witness_finish(w.get(), bytes));
    *out = w.release();
    return GM_OK;
}

extern "C" int32_t gm_gkr_witness_destroy(gm_gkr_witness* w) {
    delete w;
    return GM_OK;
}

// the circuit's output (last_step): n_cols dense device columns of 2^num_vars elements
extern "C" int32_t gm_gkr_witness_output(const gm_gkr_witness* w, const uint64_t** d_cols, uint32_t cols_cap, uint32_t* n_cols,
                                         uint32_t* num_vars) {
    GM_REQUIRE(w, "null witness");
    if (n_cols) *n_cols = (uint32_t)w->output.cols.size();
    if (num_vars) *num_vars = w->out_vars;
    if (d_cols) {
        GM_REQUIRE(cols_cap >= w->output.cols.size(), "column pointer array too small");
        for (size_t i = 0; i < w->output.cols.size(); i++) d_cols[i] = reinterpret_cast<const uint64_t*>(w->output.cols[i]->p);
    }
    return GM_OK;
}

static int32_t gkr_prove(const gm_gkr_witness* w, const uint64_t* h_point, const uint64_t* h_evs, const uint64_t* h_tape,
                         uint64_t n_tape, const gm_transcript* cb, uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs,
                         uint64_t* h_final_point, uint32_t* n_final_point, uint64_t* h_final_evs, uint32_t* n_final_evs,
                         uint64_t* tape_used, uint64_t* rounds) {
    std::vector<Fr> msgs;
    Tape tr{h_tape, n_tape, 0, &msgs, 0, cb, 0};
    Claims c;
    c.point.resize(w->out_vars);
    if (w->out_vars) memcpy(c.point.data(), h_point, 32 * (size_t)w->out_vars);
    c.evs.resize(w->output.cols.size());
    memcpy(c.evs.data(), h_evs, 32 * c.evs.size());
    gm_gkr_witness* wm = const_cast<gm_gkr_witness*>(w);
    shared_pinned() = wm->pinned;
    struct PinnedReset { ~PinnedReset() { shared_pinned() = nullptr; } } pinned_reset;
    TRY(simple_gkr_prove(&tr, w->layers, w->advices, &c, &wm->arena, w->stream));
    if (n_msgs) *n_msgs = msgs.size();
    if (h_msgs) {
        GM_REQUIRE(msgs.size() <= msgs_cap, "message buffer too small: %zu > %llu", msgs.size(), (unsigned long long)msgs_cap);
        memcpy(h_msgs, msgs.data(), msgs.size() * sizeof(Fr));
    }
    if (n_final_point) *n_final_point = (uint32_t)c.point.size();
    if (h_final_point) memcpy(h_final_point, c.point.data(), c.point.size() * sizeof(Fr));
    if (n_final_evs) *n_final_evs = (uint32_t)c.evs.size();
    if (h_final_evs) memcpy(h_final_evs, c.evs.data(), c.evs.size() * sizeof(Fr));
    if (tr.cb_rc) return set_err(GM_ERR_STATE, "transcript write_scalars callback failed with %d", tr.cb_rc);
    if (tape_used) *tape_used = tr.pos;
    if (rounds) *rounds = tr.rounds;
    return GM_OK;
}

// TriangleAdd::prove / VecVecBintreeAdd::prove = SimpleGKR::prove (gkrs/gkr.rs:45-50) over the witness's advices.
// claims in: point (gm_gkr_witness_output's num_vars elements) + one evaluation per output column; claims out: the input
// polynomials' point and evaluations (12 for the triangle; 4 or 6 for the bintree).
extern "C" int32_t gm_gkr_prove(const gm_gkr_witness* w, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs,
                                uint64_t* h_final_point, uint32_t* n_final_point, uint64_t* h_final_evs, uint32_t* n_final_evs,
                                uint64_t* tape_used, uint64_t* rounds) {
    GM_REQUIRE(w && h_claim_evs && h_tape && (h_claim_point || !w->out_vars), "null argument");
    return gkr_prove(w, h_claim_point, h_claim_evs, h_tape, n_tape, nullptr, h_msgs, msgs_cap, n_msgs, h_final_point, n_final_point,
                     h_final_evs, n_final_evs, tape_used, rounds);
}

extern "C" int32_t gm_gkr_prove_tr(const gm_gkr_witness* w, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                   const gm_transcript* tr, uint64_t* h_final_point, uint32_t* n_final_point,
                                   uint64_t* h_final_evs, uint32_t* n_final_evs, uint64_t* n_challenges, uint64_t* rounds) {
    GM_REQUIRE(w && h_claim_evs && tr && tr->challenge && (h_claim_point || !w->out_vars), "null argument");
    return gkr_prove(w, h_claim_point, h_claim_evs, nullptr, 0, tr, nullptr, 0, nullptr, h_final_point, n_final_point, h_final_evs,
                     n_final_evs, n_challenges, rounds);
}

// =================================================================================================================
// "prove pushforward" (pippenger.rs:147-160): PushforwardProtocol::prove (pushforward/pushforward.rs:640-846) with the
// logup main phase (pushforward/logup_mainphase.rs:83-208).  The Fr columns come from the plan's last gm_msm_run
// (gm_msm_phase1_polys, gm_msm_second_phase); the G1 commitments of those columns are gm_msm_g1_outer / gm_g1_msm.
// Every sumcheck here is the eq-factored degree-2 object (the round polynomials of DenseEqSumcheckObject, sumcheck.rs:378-417)
// or the generic object with Prod3Fn; the layer maps are dense maps, `map_split_hi` (utils/algfn.rs:82-89) is a map followed
// by taking the two contiguous halves.
namespace gm {

// c_adj[i] = c_pull[i] + psi c[i] - tau   for i < n_active, the suppression term beyond (pushforward.rs:699-702)
__global__ void __launch_bounds__(256) k_pf_adj(const Fr* __restrict__ pull, const Fr* __restrict__ v, Fr psi, Fr tau, Fr supp,
                                                 uint64_t n_active, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(out + i, i < n_active ? fr_sub(fr_add(fr_load(pull + i), fr_mul(psi, fr_load(v + i))), tau) : supp);
}

// table[i] = eq[i] + psi * i - tau   (pushforward.rs:727-728)
__global__ void __launch_bounds__(256) k_pf_table(const Fr* __restrict__ eq, Fr psi, Fr tau, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(out + i, fr_sub(fr_add(fr_load(eq + i), fr_mul(psi, fr_from_u64(i))), tau));
}

// p_selector_prod[i] = eq_sel_y[i_y] * (p0[i_x] + gamma (p1[i_x] - 1) + gamma^2)   (pushforward.rs:751-759); points_xy = (x, y) pairs
__global__ void __launch_bounds__(256) k_pf_psel(const Fr* __restrict__ points_xy, const Fr* __restrict__ eq_sel_y, Fr g1, Fr g2,
                                                  uint32_t x_log, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t iy = i >> x_log, ix = i & ((1ull << x_log) - 1);
    const Fr p0 = fr_load(points_xy + 2 * ix), p1 = fr_load(points_xy + 2 * ix + 1);
    const Fr pf = fr_add(fr_add(p0, fr_mul(g1, fr_sub(p1, fr_one()))), g2);
    fr_store(out + i, fr_mul(fr_load(eq_sel_y + iy), pf));
}

}  // namespace gm

extern "C" {
int32_t gm_msm_phase1_polys(const gm_msm_plan* p, uint64_t* d_c, uint64_t* d_d, uint64_t* d_ac_c, uint64_t* d_ac_d, void* stream);
int32_t gm_msm_second_phase(const gm_msm_plan* p, const uint64_t* h_r, uint32_t y_logsize, uint64_t* d_c_pull, uint64_t* d_d_pull,
                            void* stream);
int32_t gm_sc_dense_create(int32_t kind, const gm_fn* f, uint32_t num_vars, const uint64_t* const* d_cols, const uint64_t* h_gamma,
                           const uint64_t* h_claim, gm_sc** out, void* stream);
}

namespace {

struct Frac {  // one (numerator, denominator) pair of the logup tree; the arrays may be halves of a parent's buffers
    const Fr* num = nullptr;
    const Fr* den = nullptr;
    uint64_t len = 0;
    std::shared_ptr<DevBuf> keep_n, keep_d;
};

int32_t read_fr(const Fr* d, Fr* h, hipStream_t s) {
    GM_HIP(hipMemcpyAsync(h, d, sizeof(Fr), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    return GM_OK;
}

struct PfCols {  // the Fr columns of the argument, handed to the opening phase (sharded: this rank's slices of c, d, c_pull, d_pull;
                 // the access counts whole on every rank, both in one buffer)
    std::shared_ptr<DevBuf> c, d, c_pull, d_pull, ac_c, ac_d;
    const Fr* ac_c_p = nullptr;
    const Fr* ac_d_p = nullptr;
};

int32_t pushforward_prove(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_log, const uint64_t* h_claim_point,
                          const uint64_t* h_claim_evs, Tape* tr, Fr* out_gamma, Claims* out_matrix, Claims* out_ac_c,
                          Claims* out_ac_d, hipStream_t s, PfCols* keep = nullptr) {
    GM_REQUIRE(plan->y0 == 0 && plan->y1 == plan->y_size, "the pushforward argument needs a plan over all windows");
    const uint32_t x_log = plan->x_log, d_log = plan->d_log, y_size = plan->y_size;
    GM_REQUIRE((1u << y_log) >= y_size, "y_logsize too small");
    const uint32_t mlog = x_log + y_log;
    GM_REQUIRE(mlog >= 1 && mlog <= 30, "matrix too large");
    const uint64_t M = 1ull << mlog, msize = (uint64_t)y_size << x_log, X = 1ull << x_log, D = 1ull << d_log;
    void* stream = reinterpret_cast<void*>(s);
    std::vector<Fr> r(y_log + d_log + x_log), evs(3);
    memcpy(r.data(), h_claim_point, r.size() * sizeof(Fr));
    memcpy(evs.data(), h_claim_evs, 3 * sizeof(Fr));
    StageTimer pf_timer("pushforward", s);
    evs[1] = fr_sub(evs[1], fr_one());  // claims.evs[1] -= 1 (pushforward.rs:641)

    // phase-1 / phase-2 columns, padded with zeros to 2^mlog (pushforward.rs:706-709)
    auto mk = [&](uint64_t n, std::shared_ptr<DevBuf>* b) -> int32_t {
        b->reset(new DevBuf());
        return (*b)->alloc(n * sizeof(Fr));
    };
    std::shared_ptr<DevBuf> c, d, ac_c, ac_d, c_pull, d_pull, c_adj, d_adj, num, den, table_c, table_d, p_sel, eqs;
    TRY(mk(M, &c)); TRY(mk(M, &d)); TRY(mk(X, &ac_c)); TRY(mk(D, &ac_d)); TRY(mk(M, &c_pull)); TRY(mk(M, &d_pull));
    if (M > msize) {
        for (auto* b : {&c, &d, &c_pull, &d_pull}) GM_HIP(hipMemsetAsync((*b)->fr() + msize, 0, (M - msize) * sizeof(Fr), s));
    }
    pf_timer.mark("alloc");
    TRY(gm_msm_phase1_polys(plan, (uint64_t*)c->p, (uint64_t*)d->p, (uint64_t*)ac_c->p, (uint64_t*)ac_d->p, stream));
    pf_timer.mark("phase1 polys");
    TRY(gm_msm_second_phase(plan, h_claim_point, y_log, (uint64_t*)c_pull->p, (uint64_t*)d_pull->p, stream));
    pf_timer.mark("second phase");

    // challenges (pushforward.rs:684-685)
    Fr psi, tau_c, tau_d, tau_s, gamma;
    {
        Fr four[4];
        TRY(tr->challenge_vec(four, 4, 512));   // challenge_vec::<F>(4, 512) (pushforward.rs:684)
        psi = four[0]; tau_c = four[1]; tau_d = four[2]; tau_s = four[3];
    }
    TRY(tr->challenge(&gamma));
    GM_REQUIRE(!fr_is_zero(tau_s) && !fr_is_zero(psi), "zero challenge (inverse().unwrap() in the reference)");

    TRY(mk(M, &c_adj)); TRY(mk(M, &d_adj)); TRY(mk(M, &num)); TRY(mk(M, &den));
    hipLaunchKernelGGL(k_pf_adj, dim3(ceil_div(M, 256)), dim3(256), 0, s, c_pull->fr(), c->fr(), psi, tau_c, tau_s, msize, M, c_adj->fr());
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pf_adj, dim3(ceil_div(M, 256)), dim3(256), 0, s, d_pull->fr(), d->fr(), psi, tau_d, tau_s, msize, M, d_adj->fr());
    GM_LAUNCH_CHECK();
    {   // [left, right] = f_addinv.map_split_hi(&[&c_adj, &d_adj]) (pushforward.rs:719)
        const Fr* in[2] = {c_adj->fr(), d_adj->fr()};
        Fr* outp[2] = {num->fr(), den->fr()};
        TRY(launch_dense_map(plan_of(mkfn(GM_FN_ADD_INVERSES, 1)), in, outp, M, s));
    }
    // tables (pushforward.rs:725-728): eq_c / eq_d with all their levels in scratch
    TRY(mk(2 * X + 2 * D, &eqs)); TRY(mk(X, &table_c)); TRY(mk(D, &table_d));
    {
        Fr* eq_c = eqs->fr();
        Fr* eq_d = eqs->fr() + 2 * X;
        std::vector<Fr*> lv(x_log + 1);
        for (uint32_t i = 0; i < x_log; i++) lv[i] = eq_c + X + ((1ull << i) - 1);
        lv[x_log] = eq_c;
        TRY(launch_eq_sequence(fr_one(), r.data() + y_log + d_log, x_log, lv.data(), s));
        lv.assign(d_log + 1, nullptr);
        for (uint32_t i = 0; i < d_log; i++) lv[i] = eq_d + D + ((1ull << i) - 1);
        lv[d_log] = eq_d;
        TRY(launch_eq_sequence(fr_one(), r.data() + y_log, d_log, lv.data(), s));
        hipLaunchKernelGGL(k_pf_table, dim3(ceil_div(X, 256)), dim3(256), 0, s, eq_c, psi, tau_c, X, table_c->fr());
        GM_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_pf_table, dim3(ceil_div(D, 256)), dim3(256), 0, s, eq_d, psi, tau_d, D, table_d->fr());
        GM_LAUNCH_CHECK();
    }
    pf_timer.mark("adj + tables");
    // suppression_term_total = 2 (2^mlog - matrix_size) / tau_suppression_term (pushforward.rs:730)
    const Fr supp_total = fr_mul(fr_from_u64(2 * (M - msize)), fr_inv(tau_s));

    // ---- LogupMainphaseProtocol::make_witness (logup_mainphase.rs:83-143) over logsizes [mlog-1, mlog-1, x_log, d_log]
    std::vector<Frac> inputs(4), layers;
    inputs[0].num = num->fr(); inputs[0].den = den->fr(); inputs[0].len = M / 2; inputs[0].keep_n = num; inputs[0].keep_d = den;
    inputs[1].num = num->fr() + M / 2; inputs[1].den = den->fr() + M / 2; inputs[1].len = M / 2; inputs[1].keep_n = num; inputs[1].keep_d = den;
    inputs[2].num = ac_c->fr(); inputs[2].den = table_c->fr(); inputs[2].len = X; inputs[2].keep_n = ac_c; inputs[2].keep_d = table_c;
    inputs[3].num = ac_d->fr(); inputs[3].den = table_d->fr(); inputs[3].len = D; inputs[3].keep_n = ac_d; inputs[3].keep_d = table_d;
    std::vector<uint32_t> logsizes = {mlog - 1, mlog - 1, x_log, d_log};
    GM_REQUIRE(mlog - 1 >= x_log && x_log >= d_log, "logsizes must be non-increasing (logup_mainphase.rs:75-77)");
    size_t next_in = 2;
    layers.push_back(inputs[0]);
    layers.push_back(inputs[1]);
    const SegPlan logup = plan_of(mkfn(GM_FN_LOGUP_LAYER, 1));
    for (size_t i = 0;; i += 2) {
        const uint64_t next_size = next_in < inputs.size() ? inputs[next_in].len : 1;
        const uint64_t curr = layers[i].len;
        Frac o;
        o.len = curr;
        TRY(mk(curr, &o.keep_n)); TRY(mk(curr, &o.keep_d));
        o.num = o.keep_n->fr(); o.den = o.keep_d->fr();
        const Fr* in[4] = {layers[i].num, layers[i].den, layers[i + 1].num, layers[i + 1].den};
        Fr* outp[2] = {o.keep_n->fr(), o.keep_d->fr()};
        TRY(launch_dense_map(logup, in, outp, curr, s));
        if (curr == next_size) {
            layers.push_back(o);
            if (next_in < inputs.size()) layers.push_back(inputs[next_in++]);
            else break;
        } else {
            GM_REQUIRE(curr > next_size, "logup witness: unreachable size order");
            Frac lo = o, hi = o;
            lo.len = hi.len = curr / 2;
            hi.num = o.num + curr / 2; hi.den = o.den + curr / 2;
            layers.push_back(lo);
            layers.push_back(hi);
        }
    }
    Frac top = layers.back();
    layers.pop_back();
    GM_REQUIRE(top.len == 1, "logup witness does not end in a single fraction");
    Fr nd[2];
    TRY(read_fr(top.num, &nd[0], s));
    TRY(read_fr(top.den, &nd[1], s));
    GM_REQUIRE(!fr_is_zero(nd[1]), "logup denominator is zero (logup_mainphase.rs:161)");
    GM_REQUIRE(fr_eq(nd[0], fr_mul(nd[1], supp_total)), "logup total does not match the suppression term (logup_mainphase.rs:162)");
    tr->write_scalars({nd[0], nd[1]});

    pf_timer.mark("logup witness");
    // workspace of the sumcheck objects: fold buffers of the widest layer + eq levels
    Arena arena;
    TRY(arena.init((size_t)32 * (5 * (M / 2 + M / 4) + 2 * M) + ((size_t)64 << 20)));
    pf_timer.mark("arena");
    Fr* pinned = nullptr;
    TRY(thread_pinned_staging(&pinned));
    SharedPinnedScope pinned_scope(pinned);

    // ---- LogupMainphaseProtocol::prove (logup_mainphase.rs:156-208)
    uint32_t curr_log = 0;
    Claims running;
    running.evs = {nd[0], nd[1]};
    std::vector<Claims> accumulated;
    Claims last;
    const gm_fn f_logup = mkfn(GM_FN_LOGUP_LAYER, 1);
    for (;;) {
        const uint32_t incoming = logsizes.back();
        GM_REQUIRE(layers.size() >= 2, "logup witness exhausted");
        const Frac rr = layers.back(); layers.pop_back();
        const Frac ll = layers.back(); layers.pop_back();
        GM_REQUIRE(ll.len == (1ull << curr_log) && rr.len == ll.len, "logup layer size mismatch");
        Claims c4 = running;
        if (curr_log == 0) {
            // DenseEqSumcheck over zero variables (sumcheck.rs:844-872): gamma is drawn, no rounds, the four values are sent
            Fr g0;
            TRY(tr->challenge(&g0));
            std::vector<Fr> v(4);
            TRY(read_fr(ll.num, &v[0], s)); TRY(read_fr(ll.den, &v[1], s)); TRY(read_fr(rr.num, &v[2], s)); TRY(read_fr(rr.den, &v[3], s));
            tr->write_scalars(v);
            c4.point.clear();
            c4.evs = v;
        } else {
            arena.reset();
            ArenaScope scope(&arena);
            Advice adv;
            adv.kind = Advice::DENSE;
            adv.len = ll.len;
            for (const Fr* ptr : {ll.num, ll.den, rr.num, rr.den}) {
                adv.cols.emplace_back(new DevBuf());
                adv.cols.back()->p = const_cast<Fr*>(ptr);   // borrowed view
                adv.cols.back()->owned = false;
                adv.cols.back()->bytes = ll.len * sizeof(Fr);
            }
            TRY(dense_deg2_prove(tr, f_logup, curr_log, &c4, adv, s));
        }
        if (incoming == curr_log) {
            if (logsizes.size() == 2) { last = c4; break; }
            running.point = c4.point;
            running.evs = {c4.evs[0], c4.evs[1]};
            Claims a;
            a.point = c4.point;
            a.evs = {c4.evs[2], c4.evs[3]};
            accumulated.push_back(a);
            logsizes.pop_back();
        } else {
            running = c4;
            TRY(split_at_prove(tr, &running, true, 0, 2));
            curr_log++;
        }
    }
    accumulated.push_back(last);
    std::reverse(accumulated.begin(), accumulated.end());
    GM_REQUIRE(accumulated.size() == 3, "logup main phase must end with 3 claims");
    Claims cd = accumulated[0];
    *out_ac_c = accumulated[1];
    *out_ac_d = accumulated[2];
    TRY(split_at_prove(tr, &cd, true, 0, 2));   // SplitAt(HI(0), 2) (pushforward.rs:744-746)
    GM_REQUIRE(cd.evs.size() == 2 && cd.point.size() == mlog, "cd claims have the wrong shape");

    pf_timer.mark("logup prove");
    // ---- combined sumcheck (pushforward.rs:748-801)
    const Fr g1 = gamma, g2 = fr_mul(gamma, gamma);
    TRY(mk(M, &p_sel));
    {
        // EqTruncPoly(y_log, y_size, r_y).evals() (verifier_polys.rs:98-106): tiny, built on the host
        std::vector<Fr> e(1ull << y_log, fr_zero());
        e[0] = fr_one();
        for (uint32_t i = 0; i < y_log; i++)
            for (uint64_t j = (1ull << i); j-- > 0;) {
                const Fr w = e[j], m = fr_mul(r[i], w);
                e[2 * j] = fr_sub(w, m);
                e[2 * j + 1] = m;
            }
        for (uint64_t i = y_size; i < (1ull << y_log); i++) e[i] = fr_zero();
        std::shared_ptr<DevBuf> d_e;
        TRY(mk(e.size(), &d_e));
        GM_HIP(hipMemcpyAsync(d_e->p, e.data(), e.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pf_psel, dim3(ceil_div(M, 256)), dim3(256), 0, s, reinterpret_cast<const Fr*>(d_points_xy), d_e->fr(), g1,
                           g2, x_log, M, p_sel->fr());
        GM_LAUNCH_CHECK();
        GM_HIP(hipStreamSynchronize(s));  // e / d_e go out of scope
    }
    const Fr ev_folded = fr_add(fr_add(evs[0], fr_mul(g1, evs[1])), fr_mul(g2, evs[2]));
    Fr claim = fr_add(fr_add(cd.evs[0], fr_mul(g1, cd.evs[1])), fr_mul(g2, ev_folded));
    arena.reset();
    ArenaScope scope(&arena);
    PinnedSharedScope two_objects;   // prod3 and frac run in lock-step and share the pinned staging
    ScHolder prod3, frac;
    {
        const uint64_t* pc[3] = {(const uint64_t*)p_sel->p, (const uint64_t*)c_pull->p, (const uint64_t*)d_pull->p};
        TRY(gm_sc_dense_create(1, nullptr, mlog, pc, nullptr, reinterpret_cast<const uint64_t*>(&ev_folded), &prod3.so, stream));
        const uint64_t* fc[2] = {(const uint64_t*)c_adj->p, (const uint64_t*)d_adj->p};
        const gm_fn f_inv = mkfn(GM_FN_ADD_INVERSES, 1);
        TRY(gm_sc_dense_deg2_create(&f_inv, mlog, fc, reinterpret_cast<const uint64_t*>(cd.point.data()),
                                    reinterpret_cast<const uint64_t*>(&gamma), reinterpret_cast<const uint64_t*>(cd.evs.data()),
                                    &frac.so, stream));
    }
    std::vector<Fr> out_pt;
    for (uint32_t i = 0; i < mlog; i++) {
        Fr pr[8], fq[8];
        uint32_t n1 = 0, n2 = 0;
        TRY(gm_sc_unipoly(prod3.so, reinterpret_cast<uint64_t*>(pr), &n1));
        TRY(gm_sc_unipoly(frac.so, reinterpret_cast<uint64_t*>(fq), &n2));
        GM_REQUIRE(n1 == 4 && n2 == 4, "combined sumcheck: responses must have 4 coefficients");
        std::vector<Fr> comb(4);
        for (int k = 0; k < 4; k++) comb[k] = fr_add(fq[k], fr_mul(g2, pr[k]));
        const Fr chk = fr_add(fr_add(fr_dbl(comb[0]), comb[1]), fr_add(comb[2], comb[3]));
        GM_REQUIRE(fr_eq(chk, claim), "combined sumcheck: round %u does not sum to the claim (pushforward.rs:789)", i);
        tr->write_scalars({comb[0], comb[2], comb[3]});   // compress_coefficients
        Fr t;
        TRY(tr->challenge(&t));
        claim = evaluate_univar(comb, t);
        out_pt.push_back(t);
        TRY(gm_sc_bind(prod3.so, reinterpret_cast<const uint64_t*>(&t)));
        TRY(gm_sc_bind(frac.so, reinterpret_cast<const uint64_t*>(&t)));
        tr->rounds++;
    }
    std::reverse(out_pt.begin(), out_pt.end());
    Fr pe[GM_MAX_COLS + 1], fe[GM_MAX_COLS + 1];
    uint32_t ne = 0;
    TRY(gm_sc_final_evals(prod3.so, reinterpret_cast<uint64_t*>(pe), &ne));
    GM_REQUIRE(ne == 3, "prod3 final evaluations");
    TRY(gm_sc_final_evals(frac.so, reinterpret_cast<uint64_t*>(fe), &ne));
    GM_REQUIRE(ne == 2, "frac final evaluations");
    const Fr p_sel_ev = pe[0], c_pull_ev = pe[1], d_pull_ev = pe[2], c_adj_ev = fe[0], d_adj_ev = fe[1];
    const Fr eqy = eq_trunc_evaluate(y_log, y_size, r.data(), out_pt.data());
    GM_REQUIRE(!fr_is_zero(eqy), "eq_sel_y evaluates to zero (inverse().unwrap(), pushforward.rs:808)");
    const Fr p_folded_ev = fr_add(fr_mul(p_sel_ev, fr_inv(eqy)), gamma);
    const Fr sel_ev = eq_sum_host(out_pt.data(), y_log, y_size);
    const Fr tmp = fr_mul(tau_s, fr_sub(fr_one(), sel_ev));
    const Fr psi_inv = fr_inv(psi);
    const Fr c_ev = fr_mul(psi_inv, fr_sub(fr_add(fr_sub(c_adj_ev, c_pull_ev), fr_mul(tau_c, sel_ev)), tmp));
    const Fr d_ev = fr_mul(psi_inv, fr_sub(fr_add(fr_sub(d_adj_ev, d_pull_ev), fr_mul(tau_d, sel_ev)), tmp));
    out_matrix->point = out_pt;
    out_matrix->evs = {p_folded_ev, c_pull_ev, d_pull_ev, c_ev, d_ev};
    tr->write_scalars(out_matrix->evs);
    pf_timer.mark("combined sumcheck");
    *out_gamma = gamma;
    if (keep) {
        keep->c = c; keep->d = d; keep->c_pull = c_pull; keep->d_pull = d_pull; keep->ac_c = ac_c; keep->ac_d = ac_d;
        keep->ac_c_p = ac_c->fr(); keep->ac_d_p = ac_d->fr();
    }
    return GM_OK;
}

}  // namespace

// =================================================================================================================
// The same argument with the matrix sharded by windows.  Every array of the argument is indexed (y, x) with the window
// in the high bits, so a rank's windows are a contiguous slice of each of them and
//   * c, d, c_pull, d_pull, c_adj, d_adj, p_selector_prod are built from the rank's own plan (its windows' digits and counters);
//   * every sumcheck -- the layers of the logup main phase and the combined Prod3 + fraction sumcheck -- binds the low variables
//     first: the sharded dense objects (round sums exchanged through gm_comm, the last log2(world) rounds replicated) serve them;
//   * the access counts are sums over the ranks' windows (one exchange of 2^x_logsize + 2^d_logsize elements);
//   * what does NOT stay local is the WITNESS of the logup tree: map_split_hi pairs element i with element i + len / 2, so after
//     every level the two halves are re-spread over the ranks (new rank j: l-slice from old rank j / 2, r-slice from old rank
//     world / 2 + j / 2): device to device through gm_comm::pull_dev when the communicator has it (the shared-memory one does:
//     HIP IPC handles, then copies between the devices -- over xGMI on a node, local between ranks that share a device), else
//     staged through the communicator's all-gather on host buffers; below `GM_PF_DIST_MIN` elements per rank (default 1024) a
//     level is gathered once and the tree goes on replicated.
// Requires y_size = 2^y_logsize (as the sharded image part) and world | y_size.  Same transcript on every rank, same messages and
// claims as the unsharded argument.
namespace gm {
__global__ void __launch_bounds__(256) k_pf_psel_at(const Fr* __restrict__ points_xy, const Fr* __restrict__ eq_sel_y, Fr g1, Fr g2,
                                                     uint32_t x_log, uint64_t base, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t gi = base + i, iy = gi >> x_log, ix = gi & ((1ull << x_log) - 1);
    const Fr p0 = fr_load(points_xy + 2 * ix), p1 = fr_load(points_xy + 2 * ix + 1);
    const Fr pf = fr_add(fr_add(p0, fr_mul(g1, fr_sub(p1, fr_one()))), g2);
    fr_store(out + i, fr_mul(fr_load(eq_sel_y + iy), pf));
}
// development aid (GM_PF_DEBUG_SUMS=1): XOR of the 64-bit words of an array -- an order-independent checksum of a buffer
__global__ void __launch_bounds__(256) k_pf_xor_words(const unsigned long long* __restrict__ p, uint64_t n_words, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) acc ^= p[i];
    for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicXor(out, acc);
}
// out[i] = sum over the parts of parts[r * n + i]
__global__ void __launch_bounds__(256) k_pf_sum_parts(const Fr* __restrict__ parts, uint32_t nparts, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr acc = fr_load(parts + i);
    for (uint32_t r = 1; r < nparts; r++) acc = fr_add(acc, fr_load(parts + (uint64_t)r * n + i));
    fr_store(out + i, acc);
}
}  // namespace gm

namespace {

// the whole array on the host, in global order, from every rank's slice of n_loc elements (host-staged: see the header comment)
int32_t pf_gather_slices(const Shard& sh, const Fr* d_slice, uint64_t n_loc, std::vector<Fr>* all, hipStream_t s) {
    all->resize((size_t)sh.world * n_loc);
    GM_HIP(hipMemcpyAsync(all->data() + (size_t)sh.rank * n_loc, d_slice, n_loc * sizeof(Fr), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    const int32_t rc = comm_all_gather(sh.comm, all->data(), n_loc * sizeof(Fr));
    if (rc) return set_err(GM_ERR_STATE, "gm_comm all_gather failed with %d", rc);
    return GM_OK;
}

// the global access counts (sums over the ranks' windows) on every rank: ac[0..X) = ac_c, ac[X..X+D) = ac_d; d_c / d_d: this rank's
// slices of the c and d columns (phase-1 polynomials of its windows)
int32_t sharded_access_counts(const gm_msm_plan* plan, const Shard& sh, Fr* d_c, Fr* d_d, Fr* d_ac, bool* host_staged, hipStream_t s) {
    const uint64_t X = 1ull << plan->x_log, D = 1ull << plan->d_log, L = X + D;
    const uint32_t G = sh.world;
    DevBuf part, parts;
    {
        ExportableScope exported;   // a pull source: a driver block of its own, not a cut of a reserved slab
        TRY(part.alloc(L * sizeof(Fr)));
    }
    TRY(gm_msm_phase1_polys(plan, (uint64_t*)d_c, (uint64_t*)d_d, (uint64_t*)part.p, (uint64_t*)(part.fr() + X), reinterpret_cast<void*>(s)));
    TRY(parts.alloc((uint64_t)G * L * sizeof(Fr)));
    std::vector<gm_pull> pc(G);
    for (uint32_t q = 0; q < G; q++) pc[q] = gm_pull{q, 0u, 0ull, L * sizeof(Fr), parts.fr() + (uint64_t)q * L};
    TRY(shard_pull(sh, part.fr(), L, pc, host_staged, s));
    hipLaunchKernelGGL(k_pf_sum_parts, dim3(ceil_div(L, 256)), dim3(256), 0, s, parts.fr(), G, L, d_ac);
    GM_LAUNCH_CHECK();
    GM_HIP(hipStreamSynchronize(s));   // part / parts go out of scope
    return GM_OK;
}

struct DFrac {   // a (numerator, denominator) pair of the sharded logup tree: this rank's slice (dist) or the whole arrays
    const Fr* num = nullptr;
    const Fr* den = nullptr;
    uint64_t len = 0;       // GLOBAL length
    bool dist = false;
    std::shared_ptr<DevBuf> keep_n, keep_d;
};

int32_t pushforward_prove_sharded(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_log, const Shard& sh,
                                  const uint64_t* h_claim_point, const uint64_t* h_claim_evs, Tape* tr, Fr* out_gamma,
                                  Claims* out_matrix, Claims* out_ac_c, Claims* out_ac_d, hipStream_t s, PfCols* keep = nullptr) {
    const uint32_t x_log = plan->x_log, d_log = plan->d_log, y_size = plan->y_size, G = sh.world;
    GM_REQUIRE(sh.comm && G >= 2 && (G & (G - 1)) == 0, "bad sharding context");
    GM_REQUIRE(y_size == (1u << y_log), "the sharded pushforward argument needs y_size = 2^y_logsize");
    GM_REQUIRE(y_size % G == 0 && plan->nwin == y_size / G && plan->y0 == sh.rank * plan->nwin, "rank %u of %u must own windows [%u, %u)",
               sh.rank, G, sh.rank * (y_size / G), (sh.rank + 1) * (y_size / G));
    const uint32_t mlog = x_log + y_log;
    GM_REQUIRE(mlog >= 1 && mlog <= 30 && sh.lg <= y_log, "matrix too large / more ranks than windows");
    const uint64_t M = 1ull << mlog, X = 1ull << x_log, D = 1ull << d_log, ML = M / G, base = (uint64_t)sh.rank * ML;
    void* stream = reinterpret_cast<void*>(s);
    static const uint64_t dist_min = [] { const char* e = getenv("GM_PF_DIST_MIN"); return (uint64_t)(e && atoll(e) >= 1 ? atoll(e) : 1024); }();
    static const bool host_staged_env = [] { const char* e = getenv("GM_PF_HOST_STAGED"); return e && e[0] == '1'; }();   // A/B and tests: never pull_dev
    bool host_staged = host_staged_env;   // also once the communicator reports that its device path is unavailable (100: every rank alike)
    std::vector<Fr> r(y_log + d_log + x_log), evs(3);
    memcpy(r.data(), h_claim_point, r.size() * sizeof(Fr));
    memcpy(evs.data(), h_claim_evs, 3 * sizeof(Fr));
    evs[1] = fr_sub(evs[1], fr_one());  // claims.evs[1] -= 1 (pushforward.rs:641)
    StageTimer pf_timer("pushforward (sharded)", s);

    auto mk = [&](uint64_t n, std::shared_ptr<DevBuf>* b) -> int32_t {
        b->reset(new DevBuf());
        return (*b)->alloc(n * sizeof(Fr));
    };
    // The SMALL distributed levels of the logup tree come out of one slab: gm_comm::pull_dev exports the allocation a source lies in
    // (one HIP IPC handle, ~0.7 ms to open on the other side; the mappings are cached, and the pool hands the same blocks back proof
    // after proof, so a process opens each of them once) -- a slab makes that one handle per peer for all the levels below 1/8 of
    // it instead of three per level.  The slab is capped at 1 GiB: exporting a 2.2 GiB slab with every level in it hung the
    // first pull at config B (four processes on one device, ROCm 7.2; not understood, not pursued -- GM_PF_NO_SLAB=1 switches the
    // slab off altogether).  Buffers that do not fit are pool blocks of their own, as before.
    static const bool no_slab = [] { const char* e = getenv("GM_PF_NO_SLAB"); return e && e[0] == '1'; }();
    Arena tree_arena;
    const size_t slab_want = (size_t)32 * (2 * ML + 16 * dist_min * G) + ((size_t)1 << 18), slab_cap = (size_t)1 << 30;
    const size_t slab_bytes = no_slab ? 0 : (slab_want < slab_cap ? slab_want : slab_cap);
    if (slab_bytes) {
        ExportableScope exported;
        TRY(tree_arena.init(slab_bytes));
    }
    auto mk_tree = [&](uint64_t n, std::shared_ptr<DevBuf>* b) -> int32_t {
        ExportableScope exported;   // every level of the tree is a pull source
        b->reset(new DevBuf());
        if (slab_bytes && n * sizeof(Fr) <= slab_bytes / 8) {
            if (void* p = tree_arena.carve(n * sizeof(Fr))) {
                (*b)->p = p; (*b)->bytes = n * sizeof(Fr); (*b)->owned = false;
                return GM_OK;
            }
        }
        return (*b)->alloc(n * sizeof(Fr));
    };
    std::shared_ptr<DevBuf> c, d, ac_c, ac_d, c_pull, d_pull, c_adj, d_adj, num, den, table_c, table_d, p_sel, eqs;
    TRY(mk(ML, &c)); TRY(mk(ML, &d)); TRY(mk(X + D, &ac_c)); TRY(mk(ML, &c_pull)); TRY(mk(ML, &d_pull));
    // this rank's windows; its access counts are its windows' share: the global ones are the sums over the ranks
    TRY(sharded_access_counts(plan, sh, c->fr(), d->fr(), ac_c->fr(), &host_staged, s));
    pf_timer.mark("phase-1 polys + access counts");
    const Fr* ac_c_p = ac_c->fr();
    const Fr* ac_d_p = ac_c->fr() + X;
    ac_d = ac_c;
    TRY(gm_msm_second_phase(plan, h_claim_point, y_log, (uint64_t*)c_pull->p, (uint64_t*)d_pull->p, stream));

    pf_timer.mark("second phase");
    // challenges (pushforward.rs:684-685)
    Fr psi, tau_c, tau_d, tau_s, gamma;
    {
        Fr four[4];
        TRY(tr->challenge_vec(four, 4, 512));
        psi = four[0]; tau_c = four[1]; tau_d = four[2]; tau_s = four[3];
    }
    TRY(tr->challenge(&gamma));
    GM_REQUIRE(!fr_is_zero(tau_s) && !fr_is_zero(psi), "zero challenge (inverse().unwrap() in the reference)");

    // (numerator, denominator) pairs of the tree live in ONE allocation each, den = num + local length: a level's re-spread is then
    // a single pull_dev (one source buffer per call) instead of two
    TRY(mk(ML, &c_adj)); TRY(mk(ML, &d_adj)); TRY(mk_tree(2 * ML, &num)); den = num;
    hipLaunchKernelGGL(k_pf_adj, dim3(ceil_div(ML, 256)), dim3(256), 0, s, c_pull->fr(), c->fr(), psi, tau_c, tau_s, ML, ML, c_adj->fr());
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pf_adj, dim3(ceil_div(ML, 256)), dim3(256), 0, s, d_pull->fr(), d->fr(), psi, tau_d, tau_s, ML, ML, d_adj->fr());
    GM_LAUNCH_CHECK();
    {
        const Fr* in[2] = {c_adj->fr(), d_adj->fr()};
        Fr* outp[2] = {num->fr(), num->fr() + ML};
        TRY(launch_dense_map(plan_of(mkfn(GM_FN_ADD_INVERSES, 1)), in, outp, ML, s));
    }
    // tables (pushforward.rs:725-728): small, replicated
    TRY(mk(2 * X + 2 * D, &eqs)); TRY(mk(X, &table_c)); TRY(mk(D, &table_d));
    {
        Fr* eq_c = eqs->fr();
        Fr* eq_d = eqs->fr() + 2 * X;
        std::vector<Fr*> lv(x_log + 1);
        for (uint32_t i = 0; i < x_log; i++) lv[i] = eq_c + X + ((1ull << i) - 1);
        lv[x_log] = eq_c;
        TRY(launch_eq_sequence(fr_one(), r.data() + y_log + d_log, x_log, lv.data(), s));
        lv.assign(d_log + 1, nullptr);
        for (uint32_t i = 0; i < d_log; i++) lv[i] = eq_d + D + ((1ull << i) - 1);
        lv[d_log] = eq_d;
        TRY(launch_eq_sequence(fr_one(), r.data() + y_log, d_log, lv.data(), s));
        hipLaunchKernelGGL(k_pf_table, dim3(ceil_div(X, 256)), dim3(256), 0, s, eq_c, psi, tau_c, X, table_c->fr());
        GM_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_pf_table, dim3(ceil_div(D, 256)), dim3(256), 0, s, eq_d, psi, tau_d, D, table_d->fr());
        GM_LAUNCH_CHECK();
    }
    pf_timer.mark("adj + tables");
    const Fr supp_total = fr_zero();   // 2 (2^mlog - matrix_size) / tau_suppression_term with matrix_size = 2^mlog
    static const bool dbg_xor = [] { const char* e = getenv("GM_PF_DEBUG_SUMS"); return e && e[0] == '1'; }();
    DevBuf dbg_word;
    if (dbg_xor) TRY(dbg_word.alloc(8));
    auto xor_of = [&](const Fr* d, uint64_t n) -> unsigned long long {   // GM_PF_DEBUG_SUMS: checksum of n elements at d
        unsigned long long h = 0;
        if (hipMemsetAsync(dbg_word.p, 0, 8, s) != hipSuccess) return 0;
        hipLaunchKernelGGL(k_pf_xor_words, dim3(1024), dim3(256), 0, s, reinterpret_cast<const unsigned long long*>(d), n * 4,
                           reinterpret_cast<unsigned long long*>(dbg_word.p));
        (void)hipMemcpyAsync(&h, dbg_word.p, 8, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        return h;
    };

    // ---- the logup tree.  A distributed array of global length L lives as slices of L / G; `split` re-spreads its halves.
    auto split_halves = [&](const DFrac& o, DFrac* lo, DFrac* hi) -> int32_t {
        const uint64_t L = o.len, S = L / G, H = L / 2;
        lo->len = hi->len = H;
        if (!o.dist) {   // replicated: the halves are views
            *lo = o; *hi = o;
            lo->len = hi->len = H;
            hi->num = o.num + H; hi->den = o.den + H;
            return GM_OK;
        }
        const bool stay = (H / G) >= dist_min && (H / G) >= 2;
        const uint64_t n_out = stay ? H / G : H, off = stay ? (uint64_t)sh.rank * (H / G) : 0;
        for (int side = 0; side < 2; side++) {
            DFrac* t = side ? hi : lo;
            t->dist = stay;
            TRY(mk_tree(2 * n_out, &t->keep_n));
            t->keep_d = t->keep_n;
            t->num = t->keep_n->fr(); t->den = t->keep_n->fr() + n_out;
        }
        if (sh.comm->pull_dev && !host_staged) {
            // device to device.  Staying distributed: new rank j takes its l-slice from old rank j / 2 and its r-slice from old rank
            // G / 2 + j / 2 (the first or the second half of that rank's slice); turning replicated: every rank's whole slice.
            // The source is the rank's [num | den] buffer of 2 S elements: both arrays in one call.
            GM_REQUIRE(o.den == o.num + S, "logup tree: numerator and denominator are not one buffer");
            std::vector<gm_pull> pc;
            for (int arr = 0; arr < 2; arr++) {
                Fr* dlo = const_cast<Fr*>(arr ? lo->den : lo->num);
                Fr* dhi = const_cast<Fr*>(arr ? hi->den : hi->num);
                const uint64_t ab = (uint64_t)arr * S * sizeof(Fr);
                if (stay) {
                    const uint64_t half_bytes = (S / 2) * sizeof(Fr), so = (uint64_t)(sh.rank & 1u) * half_bytes;
                    pc.push_back(gm_pull{sh.rank / 2, 0u, ab + so, half_bytes, dlo});
                    pc.push_back(gm_pull{G / 2 + sh.rank / 2, 0u, ab + so, half_bytes, dhi});
                } else {
                    for (uint32_t q = 0; q < G; q++)
                        pc.push_back(gm_pull{q, 0u, ab, S * sizeof(Fr), (q < G / 2 ? dlo : dhi) + (uint64_t)(q % (G / 2)) * S});
                }
            }
            const int32_t rc = comm_pull_dev(sh.comm, o.num, 2 * S * sizeof(Fr), (uint32_t)pc.size(), pc.data(), reinterpret_cast<void*>(s));
            if (rc == 0 && dbg_xor)
                fprintf(stderr, "[gm pf tree] rank %u split len %llu %s: src [num|den] %016llx -> lo %016llx %016llx hi %016llx %016llx\n", sh.rank,
                        (unsigned long long)L, stay ? "stays distributed" : "turns replicated", xor_of(o.num, 2 * S), xor_of(lo->num, n_out),
                        xor_of(lo->den, n_out), xor_of(hi->num, n_out), xor_of(hi->den, n_out));
            if (rc == 0) return GM_OK;
            if (rc != 100) return set_err(GM_ERR_STATE, "gm_comm pull_dev failed with %d", rc);
            host_staged = true;   // nothing was copied: this and the later levels through the host
        }
        std::vector<Fr> fn, fd;   // host-staged: the whole array through the communicator's all-gather
        TRY(pf_gather_slices(sh, o.num, S, &fn, s));
        TRY(pf_gather_slices(sh, o.den, S, &fd, s));
        for (int side = 0; side < 2; side++) {
            DFrac* t = side ? hi : lo;
            GM_HIP(hipMemcpyAsync(const_cast<Fr*>(t->num), fn.data() + (side ? H : 0) + off, n_out * sizeof(Fr), hipMemcpyHostToDevice, s));
            GM_HIP(hipMemcpyAsync(const_cast<Fr*>(t->den), fd.data() + (side ? H : 0) + off, n_out * sizeof(Fr), hipMemcpyHostToDevice, s));
        }
        GM_HIP(hipStreamSynchronize(s));   // fn / fd go out of scope
        return GM_OK;
    };
    std::vector<DFrac> layers;
    {
        DFrac root;
        root.num = num->fr(); root.den = num->fr() + ML; root.len = M; root.dist = true; root.keep_n = num; root.keep_d = num;
        DFrac lo, hi;
        TRY(split_halves(root, &lo, &hi));   // [left, right] of map_split_hi (pushforward.rs:719)
        layers.push_back(lo);
        layers.push_back(hi);
    }
    struct In { const Fr* num; const Fr* den; uint64_t len; std::shared_ptr<DevBuf> kn, kd; };
    std::vector<In> inputs = {{ac_c_p, table_c->fr(), X, ac_c, table_c}, {ac_d_p, table_d->fr(), D, ac_d, table_d}};
    std::vector<uint32_t> logsizes = {mlog - 1, mlog - 1, x_log, d_log};
    GM_REQUIRE(mlog - 1 >= x_log && x_log >= d_log, "logsizes must be non-increasing (logup_mainphase.rs:75-77)");
    size_t next_in = 0;
    const SegPlan logup = plan_of(mkfn(GM_FN_LOGUP_LAYER, 1));
    for (size_t i = 0;; i += 2) {
        const uint64_t next_size = next_in < inputs.size() ? inputs[next_in].len : 1;
        const uint64_t curr = layers[i].len;
        GM_REQUIRE(layers[i].dist == layers[i + 1].dist && layers[i + 1].len == curr, "logup witness: mismatched operands");
        const bool dist = layers[i].dist;
        const uint64_t n_loc = dist ? curr / G : curr;
        DFrac o;
        o.len = curr; o.dist = dist;
        TRY(mk_tree(2 * n_loc, &o.keep_n));
        o.keep_d = o.keep_n;
        o.num = o.keep_n->fr(); o.den = o.keep_n->fr() + n_loc;
        const Fr* in[4] = {layers[i].num, layers[i].den, layers[i + 1].num, layers[i + 1].den};
        Fr* outp[2] = {const_cast<Fr*>(o.num), const_cast<Fr*>(o.den)};
        TRY(launch_dense_map(logup, in, outp, n_loc, s));
        if (curr == next_size) {
            layers.push_back(o);
            if (next_in < inputs.size()) {
                const In& a = inputs[next_in++];
                DFrac v;
                v.len = a.len; v.dist = dist; v.keep_n = a.kn; v.keep_d = a.kd;
                v.num = a.num + (dist ? (uint64_t)sh.rank * n_loc : 0);   // a replicated input seen as this rank's slice
                v.den = a.den + (dist ? (uint64_t)sh.rank * n_loc : 0);
                layers.push_back(v);
            } else break;
        } else {
            GM_REQUIRE(curr > next_size, "logup witness: unreachable size order");
            DFrac lo, hi;
            TRY(split_halves(o, &lo, &hi));
            layers.push_back(lo);
            layers.push_back(hi);
        }
    }
    DFrac top = layers.back();
    layers.pop_back();
    GM_REQUIRE(top.len == 1 && !top.dist, "logup witness does not end in a single replicated fraction");
    Fr nd[2];
    TRY(read_fr(top.num, &nd[0], s));
    TRY(read_fr(top.den, &nd[1], s));
    {   // GM_PF_DEBUG_SUMS=1 (development aid): field sums of what the tree was built from, per rank -- the access counts (whole on every
        // rank: the sums must agree across the ranks), this rank's slice of the matrix fractions, and the root
        static const bool dbg_sums = [] { const char* e = getenv("GM_PF_DEBUG_SUMS"); return e && e[0] == '1'; }();
        if (dbg_sums || !fr_eq(nd[0], fr_mul(nd[1], supp_total))) {
            auto host_sum = [&](const Fr* d, uint64_t n) -> Fr {
                std::vector<Fr> h(n);
                Fr acc = fr_zero();
                if (hipMemcpyAsync(h.data(), d, n * sizeof(Fr), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return acc;
                for (uint64_t i = 0; i < n; i++) acc = fr_add(acc, h[i]);
                return acc;
            };
            const Fr s_c = host_sum(ac_c_p, X), s_d = host_sum(ac_d_p, D), s_n = host_sum(num->fr(), ML), s_dn = host_sum(num->fr() + ML, ML);
            fprintf(stderr, "[gm pushforward sharded] rank %u: sum ac_c %08x%08x sum ac_d %08x%08x sum num slice %08x%08x sum den slice %08x%08x root %08x%08x / %08x%08x\n",
                    sh.rank, s_c.l[1], s_c.l[0], s_d.l[1], s_d.l[0], s_n.l[1], s_n.l[0], s_dn.l[1], s_dn.l[0], nd[0].l[1], nd[0].l[0], nd[1].l[1], nd[1].l[0]);
        }
    }
    GM_REQUIRE(!fr_is_zero(nd[1]), "logup denominator is zero (logup_mainphase.rs:161)");
    GM_REQUIRE(fr_eq(nd[0], fr_mul(nd[1], supp_total)), "logup total does not match the suppression term (logup_mainphase.rs:162)");
    tr->write_scalars({nd[0], nd[1]});

    pf_timer.mark("logup witness + re-spreads");
    Arena arena;
    // workspace of the sumcheck objects: fold buffers and eq levels of this rank's slices (the sharded objects keep their slice of
    // the eq tables), the replicated top of the tree
    TRY(arena.init((size_t)32 * (5 * (ML / 2 + ML / 4) + 2 * ML + 8 * dist_min * G) + ((size_t)64 << 20)));
    Fr* pinned = nullptr;
    TRY(thread_pinned_staging(&pinned));
    SharedPinnedScope pinned_scope(pinned);

    // ---- LogupMainphaseProtocol::prove (logup_mainphase.rs:156-208)
    uint32_t curr_log = 0;
    Claims running;
    running.evs = {nd[0], nd[1]};
    std::vector<Claims> accumulated;
    Claims last;
    const gm_fn f_logup = mkfn(GM_FN_LOGUP_LAYER, 1);
    for (;;) {
        const uint32_t incoming = logsizes.back();
        GM_REQUIRE(layers.size() >= 2, "logup witness exhausted");
        const DFrac rr = layers.back(); layers.pop_back();
        const DFrac ll = layers.back(); layers.pop_back();
        GM_REQUIRE(ll.len == (1ull << curr_log) && rr.len == ll.len && ll.dist == rr.dist, "logup layer size mismatch");
        Claims c4 = running;
        if (curr_log == 0) {
            Fr g0;
            TRY(tr->challenge(&g0));
            std::vector<Fr> v(4);
            TRY(read_fr(ll.num, &v[0], s)); TRY(read_fr(ll.den, &v[1], s)); TRY(read_fr(rr.num, &v[2], s)); TRY(read_fr(rr.den, &v[3], s));
            tr->write_scalars(v);
            c4.point.clear();
            c4.evs = v;
        } else {
            arena.reset();
            ArenaScope scope(&arena);
            const uint64_t n_loc = ll.dist ? ll.len / G : ll.len;
            Advice adv;
            adv.kind = Advice::DENSE;
            adv.len = n_loc;
            for (const Fr* ptr : {ll.num, ll.den, rr.num, rr.den}) {
                adv.cols.emplace_back(new DevBuf());
                adv.cols.back()->p = const_cast<Fr*>(ptr);   // borrowed view
                adv.cols.back()->owned = false;
                adv.cols.back()->bytes = n_loc * sizeof(Fr);
            }
            ShardScope shard(ll.dist ? sh : Shard());   // a distributed layer: the columns are this rank's slice of 2^curr_log elements
            // a replicated layer (the top of the tree, below GM_PF_DIST_MIN elements per rank): every rank would prove it alike -- the
            // leader does, the others read its round polynomials from the exchange (LeadScope, as the bucket-reduction layers)
            static const bool no_lead = [] { const char* e = getenv("GM_SHARD_NO_LEADER"); return e && e[0] == '1'; }();
            LeadScope lead((!ll.dist && !no_lead) ? sh : Shard());
            TRY(dense_deg2_prove(tr, f_logup, curr_log, &c4, adv, s));
        }
        if (incoming == curr_log) {
            if (logsizes.size() == 2) { last = c4; break; }
            running.point = c4.point;
            running.evs = {c4.evs[0], c4.evs[1]};
            Claims a;
            a.point = c4.point;
            a.evs = {c4.evs[2], c4.evs[3]};
            accumulated.push_back(a);
            logsizes.pop_back();
        } else {
            running = c4;
            TRY(split_at_prove(tr, &running, true, 0, 2));
            curr_log++;
        }
    }
    accumulated.push_back(last);
    std::reverse(accumulated.begin(), accumulated.end());
    GM_REQUIRE(accumulated.size() == 3, "logup main phase must end with 3 claims");
    Claims cd = accumulated[0];
    *out_ac_c = accumulated[1];
    *out_ac_d = accumulated[2];
    TRY(split_at_prove(tr, &cd, true, 0, 2));   // SplitAt(HI(0), 2) (pushforward.rs:744-746)
    GM_REQUIRE(cd.evs.size() == 2 && cd.point.size() == mlog, "cd claims have the wrong shape");

    pf_timer.mark("logup prove");
    // ---- combined sumcheck (pushforward.rs:748-801) on this rank's slices
    const Fr g1 = gamma, g2 = fr_mul(gamma, gamma);
    TRY(mk(ML, &p_sel));
    {
        std::vector<Fr> e(1ull << y_log, fr_zero());
        e[0] = fr_one();
        for (uint32_t i = 0; i < y_log; i++)
            for (uint64_t j = (1ull << i); j-- > 0;) {
                const Fr w = e[j], m = fr_mul(r[i], w);
                e[2 * j] = fr_sub(w, m);
                e[2 * j + 1] = m;
            }
        std::shared_ptr<DevBuf> d_e;
        TRY(mk(e.size(), &d_e));
        GM_HIP(hipMemcpyAsync(d_e->p, e.data(), e.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pf_psel_at, dim3(ceil_div(ML, 256)), dim3(256), 0, s, reinterpret_cast<const Fr*>(d_points_xy), d_e->fr(), g1,
                           g2, x_log, base, ML, p_sel->fr());
        GM_LAUNCH_CHECK();
        GM_HIP(hipStreamSynchronize(s));
    }
    const Fr ev_folded = fr_add(fr_add(evs[0], fr_mul(g1, evs[1])), fr_mul(g2, evs[2]));
    Fr claim = fr_add(fr_add(cd.evs[0], fr_mul(g1, cd.evs[1])), fr_mul(g2, ev_folded));
    arena.reset();
    ArenaScope scope(&arena);
    PinnedSharedScope two_objects;
    ScHolder prod3, frac;
    {
        ShardScope shard(sh);
        const uint64_t* pc[3] = {(const uint64_t*)p_sel->p, (const uint64_t*)c_pull->p, (const uint64_t*)d_pull->p};
        TRY(gm_sc_dense_create(1, nullptr, mlog, pc, nullptr, reinterpret_cast<const uint64_t*>(&ev_folded), &prod3.so, stream));
        const uint64_t* fc[2] = {(const uint64_t*)c_adj->p, (const uint64_t*)d_adj->p};
        const gm_fn f_inv = mkfn(GM_FN_ADD_INVERSES, 1);
        TRY(gm_sc_dense_deg2_create(&f_inv, mlog, fc, reinterpret_cast<const uint64_t*>(cd.point.data()),
                                    reinterpret_cast<const uint64_t*>(&gamma), reinterpret_cast<const uint64_t*>(cd.evs.data()),
                                    &frac.so, stream));
    }
    std::vector<Fr> out_pt;
    for (uint32_t i = 0; i < mlog; i++) {
        Fr pr[8], fq[8];
        uint32_t n1 = 0, n2 = 0;
        TRY(gm_sc_unipoly(prod3.so, reinterpret_cast<uint64_t*>(pr), &n1));
        TRY(gm_sc_unipoly(frac.so, reinterpret_cast<uint64_t*>(fq), &n2));
        GM_REQUIRE(n1 == 4 && n2 == 4, "combined sumcheck: responses must have 4 coefficients");
        std::vector<Fr> comb(4);
        for (int k = 0; k < 4; k++) comb[k] = fr_add(fq[k], fr_mul(g2, pr[k]));
        const Fr chk = fr_add(fr_add(fr_dbl(comb[0]), comb[1]), fr_add(comb[2], comb[3]));
        GM_REQUIRE(fr_eq(chk, claim), "combined sumcheck: round %u does not sum to the claim (pushforward.rs:789)", i);
        tr->write_scalars({comb[0], comb[2], comb[3]});
        Fr t;
        TRY(tr->challenge(&t));
        claim = evaluate_univar(comb, t);
        out_pt.push_back(t);
        TRY(gm_sc_bind(prod3.so, reinterpret_cast<const uint64_t*>(&t)));
        TRY(gm_sc_bind(frac.so, reinterpret_cast<const uint64_t*>(&t)));
        tr->rounds++;
    }
    std::reverse(out_pt.begin(), out_pt.end());
    Fr pe[GM_MAX_COLS + 1], fe[GM_MAX_COLS + 1];
    uint32_t ne = 0;
    TRY(gm_sc_final_evals(prod3.so, reinterpret_cast<uint64_t*>(pe), &ne));
    GM_REQUIRE(ne == 3, "prod3 final evaluations");
    TRY(gm_sc_final_evals(frac.so, reinterpret_cast<uint64_t*>(fe), &ne));
    GM_REQUIRE(ne == 2, "frac final evaluations");
    const Fr p_sel_ev = pe[0], c_pull_ev = pe[1], d_pull_ev = pe[2], c_adj_ev = fe[0], d_adj_ev = fe[1];
    const Fr eqy = eq_trunc_evaluate(y_log, y_size, r.data(), out_pt.data());
    GM_REQUIRE(!fr_is_zero(eqy), "eq_sel_y evaluates to zero (inverse().unwrap(), pushforward.rs:808)");
    const Fr p_folded_ev = fr_add(fr_mul(p_sel_ev, fr_inv(eqy)), gamma);
    const Fr sel_ev = eq_sum_host(out_pt.data(), y_log, y_size);
    const Fr tmp = fr_mul(tau_s, fr_sub(fr_one(), sel_ev));
    const Fr psi_inv = fr_inv(psi);
    const Fr c_ev = fr_mul(psi_inv, fr_sub(fr_add(fr_sub(c_adj_ev, c_pull_ev), fr_mul(tau_c, sel_ev)), tmp));
    const Fr d_ev = fr_mul(psi_inv, fr_sub(fr_add(fr_sub(d_adj_ev, d_pull_ev), fr_mul(tau_d, sel_ev)), tmp));
    out_matrix->point = out_pt;
    out_matrix->evs = {p_folded_ev, c_pull_ev, d_pull_ev, c_ev, d_ev};
    tr->write_scalars(out_matrix->evs);
    pf_timer.mark("combined sumcheck");
    *out_gamma = gamma;
    if (keep) {
        keep->c = c; keep->d = d; keep->c_pull = c_pull; keep->d_pull = d_pull; keep->ac_c = ac_c; keep->ac_d = ac_c;
        keep->ac_c_p = ac_c_p; keep->ac_d_p = ac_d_p;
    }
    return GM_OK;
}

int32_t put_claims(const Claims& c, uint64_t* h_point, uint64_t* h_evs) {
    if (h_point) memcpy(h_point, c.point.data(), c.point.size() * sizeof(Fr));
    if (h_evs) memcpy(h_evs, c.evs.data(), c.evs.size() * sizeof(Fr));
    return GM_OK;
}

int32_t pushforward_entry(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize, const uint64_t* h_claim_point,
                          const uint64_t* h_claim_evs, const uint64_t* h_tape, uint64_t n_tape, const gm_transcript* cb,
                          uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_gamma, uint64_t* h_matrix_point,
                          uint64_t* h_matrix_evs, uint64_t* h_ac_c_point, uint64_t* h_ac_c_evs, uint64_t* h_ac_d_point,
                          uint64_t* h_ac_d_evs, uint64_t* tape_used, uint64_t* rounds, void* stream, const gm_comm* comm = nullptr) {
    GM_REQUIRE(plan && d_points_xy && h_claim_point && h_claim_evs, "null argument");
    std::vector<Fr> msgs;
    Tape tr{h_tape, n_tape, 0, &msgs, 0, cb, 0};
    Fr gamma;
    Claims mx, acc, acd;
    if (comm && comm->world > 1) {
        GM_REQUIRE(comm->all_gather && comm->rank < comm->world, "bad gm_comm");
        Shard sh;
        sh.comm = comm; sh.rank = comm->rank; sh.world = comm->world;
        while ((1u << sh.lg) < comm->world) sh.lg++;
        TRY(pushforward_prove_sharded(plan, d_points_xy, y_logsize, sh, h_claim_point, h_claim_evs, &tr, &gamma, &mx, &acc, &acd, as_stream(stream)));
    } else
    TRY(pushforward_prove(plan, d_points_xy, y_logsize, h_claim_point, h_claim_evs, &tr, &gamma, &mx, &acc, &acd, as_stream(stream)));
    if (tr.cb_rc) return set_err(GM_ERR_STATE, "transcript write_scalars callback failed with %d", tr.cb_rc);
    if (n_msgs) *n_msgs = msgs.size();
    if (h_msgs) {
        GM_REQUIRE(msgs.size() <= msgs_cap, "message buffer too small: %zu > %llu", msgs.size(), (unsigned long long)msgs_cap);
        memcpy(h_msgs, msgs.data(), msgs.size() * sizeof(Fr));
    }
    if (h_gamma) memcpy(h_gamma, &gamma, sizeof(Fr));
    put_claims(mx, h_matrix_point, h_matrix_evs);
    put_claims(acc, h_ac_c_point, h_ac_c_evs);
    put_claims(acd, h_ac_d_point, h_ac_d_evs);
    if (tape_used) *tape_used = tr.pos;
    if (rounds) *rounds = tr.rounds;
    return GM_OK;
}

}  // namespace

extern "C" int32_t gm_pushforward_prove(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                        const uint64_t* h_claim_point, const uint64_t* h_claim_evs, const uint64_t* h_tape,
                                        uint64_t n_tape, uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_gamma,
                                        uint64_t* h_matrix_point, uint64_t* h_matrix_evs, uint64_t* h_ac_c_point,
                                        uint64_t* h_ac_c_evs, uint64_t* h_ac_d_point, uint64_t* h_ac_d_evs, uint64_t* tape_used,
                                        uint64_t* rounds, void* stream) {
    GM_REQUIRE(h_tape, "null tape");
    return pushforward_entry(plan, d_points_xy, y_logsize, h_claim_point, h_claim_evs, h_tape, n_tape, nullptr, h_msgs, msgs_cap,
                             n_msgs, h_gamma, h_matrix_point, h_matrix_evs, h_ac_c_point, h_ac_c_evs, h_ac_d_point, h_ac_d_evs,
                             tape_used, rounds, stream);
}

// The argument with the matrix sharded by windows: `plan` covers this rank's windows (gm_msm_plan_create(.., y_begin, y_end)), y_size =
// the global window count = 2^y_logsize, comm->world | y_size.  Every rank runs the same transcript and obtains the messages and
// claims of the unsharded argument.  `comm` must outlive the call.
extern "C" int32_t gm_pushforward_prove_sharded(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                                const gm_comm* comm, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                                const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_msgs, uint64_t msgs_cap,
                                                uint64_t* n_msgs, uint64_t* h_gamma, uint64_t* h_matrix_point, uint64_t* h_matrix_evs,
                                                uint64_t* h_ac_c_point, uint64_t* h_ac_c_evs, uint64_t* h_ac_d_point,
                                                uint64_t* h_ac_d_evs, uint64_t* tape_used, uint64_t* rounds, void* stream) {
    GM_REQUIRE(h_tape && comm, "null tape / gm_comm");
    return pushforward_entry(plan, d_points_xy, y_logsize, h_claim_point, h_claim_evs, h_tape, n_tape, nullptr, h_msgs, msgs_cap,
                             n_msgs, h_gamma, h_matrix_point, h_matrix_evs, h_ac_c_point, h_ac_c_evs, h_ac_d_point, h_ac_d_evs,
                             tape_used, rounds, stream, comm);
}

extern "C" int32_t gm_pushforward_prove_tr(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                           const uint64_t* h_claim_point, const uint64_t* h_claim_evs, const gm_transcript* tr,
                                           uint64_t* h_gamma, uint64_t* h_matrix_point, uint64_t* h_matrix_evs,
                                           uint64_t* h_ac_c_point, uint64_t* h_ac_c_evs, uint64_t* h_ac_d_point,
                                           uint64_t* h_ac_d_evs, uint64_t* n_challenges, uint64_t* rounds, void* stream) {
    GM_REQUIRE(tr && tr->challenge, "null transcript");
    return pushforward_entry(plan, d_points_xy, y_logsize, h_claim_point, h_claim_evs, nullptr, 0, tr, nullptr, 0, nullptr, h_gamma,
                             h_matrix_point, h_matrix_evs, h_ac_c_point, h_ac_c_evs, h_ac_d_point, h_ac_d_evs, n_challenges, rounds,
                             stream);
}

// =================================================================================================================
// MultiOpenReduction::prove (cleanup/protocols/multiopen_reduction.rs:65-93): nargs polynomials of nvars variables with one
// evaluation claim each (at different points) are reduced to claims at one common point by the sumcheck of
// sum_i gamma^i p_i(x) eq(point_i, x).  d_polys: nargs device columns of 2^nvars elements (the caller zero-pads, pippenger.rs:233);
// h_points: nargs x nvars coordinates; h_evs: nargs evaluations.
namespace {
// sh.comm != nullptr: d_polys are this rank's contiguous slices of 2^nvars / world elements; the eq tables are built for the slice
// (eq(point)[rank * n_loc + j] = eq(point[0..lg), rank) * eq(point[lg..), j)) and the sumcheck runs on the sharded dense object
int32_t multiopen_core(Tape* trp, uint32_t nvars, uint32_t nargs, const uint64_t* const* d_polys, const uint64_t* h_points,
                       const uint64_t* h_evs, std::vector<Fr>* out_pt, std::vector<Fr>* out_evs, hipStream_t s, const Shard& sh = Shard()) {
    Tape& tr = *trp;
    void* stream = reinterpret_cast<void*>(s);
    Fr gamma;
    TRY(tr.challenge(&gamma));
    // folded_claim = gamma_rlc(gamma, evs) (sumcheck.rs:591-602)
    std::vector<Fr> evs(nargs);
    memcpy(evs.data(), h_evs, nargs * sizeof(Fr));
    Fr claim = evs[nargs - 1];
    for (uint32_t i = 1; i < nargs; i++) claim = fr_add(fr_mul(claim, gamma), evs[nargs - 1 - i]);
    // advice.extend(EqPoly(point_i).evals())
    const uint32_t lg = sh.comm ? sh.lg : 0, lv_n = nvars - lg;
    GM_REQUIRE(lg <= nvars, "more ranks than elements");
    const uint64_t n = 1ull << lv_n;   // elements per column on this rank
    std::vector<std::shared_ptr<DevBuf>> eqs(nargs);
    std::vector<const uint64_t*> cols(d_polys, d_polys + nargs);
    for (uint32_t i = 0; i < nargs; i++) {
        eqs[i].reset(new DevBuf());
        TRY(eqs[i]->alloc(2 * n * sizeof(Fr)));
        std::vector<Fr> pt(nvars);
        memcpy(pt.data(), h_points + 4 * (size_t)i * nvars, nvars * sizeof(Fr));
        Fr mult = fr_one();   // eq(point[0..lg), rank): point[0] pairs with the most significant index bit
        for (uint32_t b = 0; b < lg; b++) mult = fr_mul(mult, ((sh.rank >> (lg - 1 - b)) & 1u) ? pt[b] : fr_sub(fr_one(), pt[b]));
        if (lv_n == 0) {
            GM_HIP(hipMemcpyAsync(eqs[i]->p, &mult, sizeof(Fr), hipMemcpyHostToDevice, s));
            GM_HIP(hipStreamSynchronize(s));
        } else {
            std::vector<Fr*> lv(lv_n + 1);
            for (uint32_t l = 0; l < lv_n; l++) lv[l] = eqs[i]->fr() + n + ((1ull << l) - 1);
            lv[lv_n] = eqs[i]->fr();
            TRY(launch_eq_sequence(mult, pt.data() + lg, lv_n, lv.data(), s));
        }
        cols.push_back(reinterpret_cast<const uint64_t*>(eqs[i]->p));
    }
    Fr* pinned = nullptr;
    TRY(thread_pinned_staging(&pinned));
    SharedPinnedScope pinned_scope(pinned);
    ScHolder h;
    gm_fn f = mkfn(GM_FN_ID, (int)nargs);
    {
        ShardScope scope(sh);
        TRY(gm_sc_dense_create(2, &f, nvars, cols.data(), reinterpret_cast<const uint64_t*>(&gamma), reinterpret_cast<const uint64_t*>(&claim),
                               &h.so, stream));
    }
    std::vector<Fr> fin;
    TRY(generic_sumcheck_prove(&tr, h.so, nvars, 2, out_pt, &fin));
    fin.resize(nargs);   // poly_evs[..nargs] (multiopen_reduction.rs:84)
    tr.write_scalars(fin);
    *out_evs = fin;
    return GM_OK;
}

int32_t multiopen_entry(uint32_t nvars, uint32_t nargs, const uint64_t* const* d_polys, const uint64_t* h_points,
                        const uint64_t* h_evs, const uint64_t* h_tape, uint64_t n_tape, const gm_transcript* cb, uint64_t* h_msgs,
                        uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_out_point, uint64_t* h_out_evs, uint64_t* tape_used,
                        uint64_t* rounds, void* stream) {
    GM_REQUIRE(d_polys && h_points && h_evs && nvars >= 1 && nvars <= 28 && nargs >= 1 && nargs <= 8, "bad argument");
    std::vector<Fr> msgs;
    Tape tr{h_tape, n_tape, 0, &msgs, 0, cb, 0};
    std::vector<Fr> pt, fin;
    TRY(multiopen_core(&tr, nvars, nargs, d_polys, h_points, h_evs, &pt, &fin, as_stream(stream)));
    if (tr.cb_rc) return set_err(GM_ERR_STATE, "transcript write_scalars callback failed with %d", tr.cb_rc);
    if (n_msgs) *n_msgs = msgs.size();
    if (h_msgs) {
        GM_REQUIRE(msgs.size() <= msgs_cap, "message buffer too small");
        memcpy(h_msgs, msgs.data(), msgs.size() * sizeof(Fr));
    }
    if (h_out_point) memcpy(h_out_point, pt.data(), pt.size() * sizeof(Fr));
    if (h_out_evs) memcpy(h_out_evs, fin.data(), fin.size() * sizeof(Fr));
    if (tape_used) *tape_used = tr.pos;
    if (rounds) *rounds = tr.rounds;
    return GM_OK;
}
}  // namespace

extern "C" int32_t gm_multiopen_prove(uint32_t nvars, uint32_t nargs, const uint64_t* const* d_polys, const uint64_t* h_points,
                                      const uint64_t* h_evs, const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_msgs,
                                      uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_out_point, uint64_t* h_out_evs,
                                      uint64_t* tape_used, uint64_t* rounds, void* stream) {
    GM_REQUIRE(h_tape, "null tape");
    return multiopen_entry(nvars, nargs, d_polys, h_points, h_evs, h_tape, n_tape, nullptr, h_msgs, msgs_cap, n_msgs, h_out_point,
                           h_out_evs, tape_used, rounds, stream);
}

extern "C" int32_t gm_multiopen_prove_tr(uint32_t nvars, uint32_t nargs, const uint64_t* const* d_polys, const uint64_t* h_points,
                                         const uint64_t* h_evs, const gm_transcript* tr, uint64_t* h_out_point, uint64_t* h_out_evs,
                                         uint64_t* n_challenges, uint64_t* rounds, void* stream) {
    GM_REQUIRE(tr && tr->challenge, "null transcript");
    return multiopen_entry(nvars, nargs, d_polys, h_points, h_evs, nullptr, 0, tr, nullptr, 0, nullptr, h_out_point, h_out_evs,
                           n_challenges, rounds, stream);
}

// =================================================================================================================
// The whole gen-2 prover: PippengerWG::new (pippenger.rs:37-70) and Pippenger::prove (pippenger.rs:118-290) behind two calls.
// Everything below is orchestration of entry points that exist on their own (and are tested on their own): bucketing
// (gm_msm_run, done by the caller), phase-1 commitments (gm_msm_g1_outer, gm_g1_msm), the image part, second_phase and its
// commitments (gm_g1_msm_nonaff over the outer buckets), the pushforward argument, MultiOpenReduction and the Knuckles
// opening; plus the host-side scalar / G1 glue of the "open" span.
#include "g1.hip.h"

extern "C" {
int32_t gm_msm_g1_outer(const gm_msm_plan* plan, const uint64_t* d_basis_aff, uint32_t clm, uint64_t* d_d_outer, uint64_t* d_c_outer,
                        uint64_t c_outer_cap, uint32_t* c_stride, uint64_t* h_d_comm, uint64_t* h_c_comm, void* stream);
int32_t gm_g1_msm(const uint64_t* d_bases_aff, const uint64_t* d_scalars, uint64_t n, int32_t scalars_mont, uint32_t nbits,
                  uint64_t* h_out_aff, void* stream);
int32_t gm_g1_msm_nonaff(const uint64_t* d_bases_jac, const uint64_t* d_scalars, uint64_t n, int32_t scalars_mont, uint32_t nbits,
                         uint64_t* h_out_aff, void* stream);
int32_t gm_g1_msm_nonaff_grouped(const uint64_t* d_bases_jac, uint64_t stride, const uint32_t* h_n, uint32_t n_groups,
                                 const uint64_t* d_scalars, int32_t scalars_mont, uint32_t nbits, uint64_t* h_out_aff, void* stream);
int32_t gm_msm_g1_outer_part(const gm_msm_plan* plan, const uint64_t* d_basis_local, const int32_t* h_slot, uint32_t clm,
                             uint64_t* d_d_outer, uint64_t* d_c_outer, uint64_t c_outer_cap, uint32_t* c_stride, uint32_t* first_matrix,
                             uint32_t* n_matrices, uint64_t* h_d_part_jac, uint64_t* h_c_part_jac, void* stream);
int32_t gm_g1_combine_parts(const gm_comm* comm, const uint64_t* h_parts_jac, uint32_t n, uint64_t* h_out_aff);
int32_t gm_knuckles_open_sharded_tr(const gm_comm* comm, const gm_key_view* key, const uint64_t* d_inverses_slice, const uint64_t* h_k,
                                    uint32_t num_vars, const uint64_t* d_poly_slice, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                                    const uint64_t* h_commitment_aff, const gm_transcript* tr, uint64_t* h_proof, uint64_t* h_pair,
                                    void* stream);
int32_t gm_knuckles_open_tr(const uint64_t* d_basis_aff, const uint64_t* d_inverses, const uint64_t* h_k, uint32_t num_vars,
                            const uint64_t* d_poly, uint64_t poly_len, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                            const uint64_t* h_commitment_aff, const gm_transcript* tr, uint64_t* h_proof, uint64_t* h_pair,
                            void* stream);
}

namespace gm {

__global__ void __launch_bounds__(256) k_pp_split_xy(const Fr* __restrict__ pts, uint64_t n, Fr* __restrict__ p0, Fr* __restrict__ p1) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(p0 + i, fr_load(pts + 2 * i));
    fr_store(p1 + i, fr_load(pts + 2 * i + 1));
}

// out[i] = a[i] + g * b[i]
__global__ void __launch_bounds__(256) k_pp_axpy(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr g, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(out + i, fr_add(fr_load(a + i), fr_mul(g, fr_load(b + i))));
}

// combined_witness (pippenger.rs:208-222): out[i] = sum_{y : y % cm == i >> x_log} multirow[y / cm] *
//   (c[idx] + d[idx] u + c_pull[idx] u^2 + d_pull[idx] u^3),  idx = (i mod 2^x_log) + 2^x_log * y
struct Us { Fr u[4]; };
__global__ void __launch_bounds__(256) k_pp_combined(const Fr* __restrict__ c, const Fr* __restrict__ d, const Fr* __restrict__ cp,
                                                      const Fr* __restrict__ dp, const Fr* __restrict__ multirow, Us us,
                                                      uint32_t x_log, uint32_t y_size, uint32_t clm, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = i & ((1ull << x_log) - 1);
    const uint32_t y_rem = (uint32_t)(i >> x_log), cm = 1u << clm;
    Fr acc = fr_zero();
    for (uint32_t y = y_rem; y < y_size; y += cm) {
        const uint64_t idx = x + ((uint64_t)y << x_log);
        Fr v = fr_load(c + idx);
        v = fr_add(v, fr_mul(fr_load(d + idx), us.u[1]));
        v = fr_add(v, fr_mul(fr_load(cp + idx), us.u[2]));
        v = fr_add(v, fr_mul(fr_load(dp + idx), us.u[3]));
        acc = fr_add(acc, fr_mul(fr_load(multirow + (y >> clm)), v));
    }
    fr_store(out + i, acc);
}

// One rank's share of combined_witness from the windows [y0, y1) it holds (c, d, cp, dp: its slices, window-major from y0):
// part[i] covers witness index a0 + i, a0 = rem0 << x_log -- rem0 = y0 mod 2^clm when the rank's windows are fewer than 2^clm
// (each remainder then has exactly one of them), 0 otherwise.
__global__ void __launch_bounds__(256) k_pp_combined_part(const Fr* __restrict__ c, const Fr* __restrict__ d, const Fr* __restrict__ cp,
                                                           const Fr* __restrict__ dp, const Fr* __restrict__ multirow, Us us, uint32_t x_log,
                                                           uint32_t y0, uint32_t y1, uint32_t clm, uint32_t rem0, uint64_t n,
                                                           Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = i & ((1ull << x_log) - 1);
    const uint32_t cm = 1u << clm, y_rem = rem0 + (uint32_t)(i >> x_log);
    Fr acc = fr_zero();
    for (uint32_t y = y0 + ((y_rem - y0) & (cm - 1)); y < y1; y += cm) {
        const uint64_t idx = x + ((uint64_t)(y - y0) << x_log);
        Fr v = fr_load(c + idx);
        v = fr_add(v, fr_mul(fr_load(d + idx), us.u[1]));
        v = fr_add(v, fr_mul(fr_load(cp + idx), us.u[2]));
        v = fr_add(v, fr_mul(fr_load(dp + idx), us.u[3]));
        acc = fr_add(acc, fr_mul(fr_load(multirow + (y >> clm)), v));
    }
    fr_store(out + i, acc);
}

// this rank's slice [base, base + n) of a short replicated column zero-padded to the witness length: out[i] = src[base + i] below len
__global__ void __launch_bounds__(256) k_pp_pad_slice(const Fr* __restrict__ src, uint64_t len, uint64_t base, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(out + i, base + i < len ? fr_load(src + base + i) : fr_zero());
}
// the same for p_0 + gamma p_1
__global__ void __launch_bounds__(256) k_pp_axpy_slice(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr g, uint64_t len, uint64_t base,
                                                        uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(out + i, base + i < len ? fr_add(fr_load(a + base + i), fr_mul(g, fr_load(b + base + i))) : fr_zero());
}

// out[i] = sum_j q[j] w_j[i]  (the folded opening witness, pippenger.rs:269-275)
struct Cols4 { const Fr* p[4]; };
__global__ void __launch_bounds__(256) k_pp_fold4(Cols4 w, Us q, uint64_t n, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr acc = fr_mul(fr_load(w.p[0] + i), q.u[0]);
#pragma unroll
    for (int j = 1; j < 4; j++) acc = fr_add(acc, fr_mul(fr_load(w.p[j] + i), q.u[j]));
    fr_store(out + i, acc);
}

}  // namespace gm

struct gm_pippenger_wg {
    const gm_msm_plan* plan = nullptr;
    const uint64_t* d_points_xy = nullptr;
    const uint64_t* d_basis = nullptr;
    uint32_t y_log = 0, clm = 0, n_mat = 0, c_stride = 0;
    gm_pip_witness* w = nullptr;
    std::shared_ptr<DevBuf> d_outer, c_outer, p0, p1;
    std::vector<uint64_t> comm_c, comm_d;          // n_mat affine points each
    std::vector<uint32_t> c_upper;                 // c_upper_bound of every matrix (pushforward.rs:431): its longest bucket row
    uint64_t comm_p0[12], comm_p1[12], comm_ac_c[12], comm_ac_d[12];
    hipStream_t stream = nullptr;
    // sharded (gm_pippenger_wg_create_sharded): this rank's windows; d_outer / c_outer are its PARTIAL outer buckets of the matrices
    // m0 .. m0 + n_mat_loc - 1 its windows touch; the commitments above are the combined (global) ones on every rank
    Shard sh;
    uint32_t m0 = 0, n_mat_loc = 0;
    std::vector<const uint64_t*> key_seg;
    std::vector<uint64_t> key_first, key_count;
    gm_key_view key{0, 0, nullptr, nullptr, nullptr};
    ~gm_pippenger_wg() { delete w; }
};

namespace {

G1Jac pp_aff_in(const uint64_t* h) {
    G1Aff a;
    memcpy(&a, h, sizeof(G1Aff));
    return g1_from_aff(a);
}
void pp_aff_out(uint64_t* h, const G1Jac& p) {
    const G1Aff a = g1_to_aff(p);
    memcpy(h, &a, sizeof(G1Aff));
}
G1Jac pp_mul(const G1Jac& p, const Fr& k_mont) {
    const Fr k = fr_from_mont(k_mont);
    G1Jac acc = g1_inf();
    for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            acc = g1_dbl(acc);
            if ((k.l[i] >> b) & 1) acc = g1_add(acc, p);
        }
    return acc;
}

// gm_transcript adapter over the driver's Tape (the Knuckles opening is driven through its public entry point)
int32_t tape_ws(void* ctx, const uint64_t* e, uint64_t n) {
    Tape* t = static_cast<Tape*>(ctx);
    t->write_scalars(std::vector<Fr>(reinterpret_cast<const Fr*>(e), reinterpret_cast<const Fr*>(e) + n));
    return t->cb_rc;
}
int32_t tape_ch(void* ctx, uint32_t cnt, uint32_t bits, uint64_t* out) {
    Tape* t = static_cast<Tape*>(ctx);
    return t->challenge_raw(reinterpret_cast<Fr*>(out), cnt, bits);
}
int32_t tape_wp(void* ctx, const uint64_t* aff, uint64_t n) {
    Tape* t = static_cast<Tape*>(ctx);
    t->write_points(aff, n);
    return t->cb_rc;
}

// The reference's tracing spans of Pippenger::prove (pippenger.rs:121-159: "prove image part", the phase-2 commitments, "prove
// pushforward", "open") of the calling thread's last proof, in milliseconds of host wall time; every span ends with results on the
// host, so no extra synchronisation is needed to attribute it.  [0] image part, [1] phase-2 commitments, [2] pushforward,
// [3] opening witnesses + commitment combinations, [4] MultiOpenReduction, [5] Knuckles opening.
struct ProveSpans {
    double ms[8] = {0};
    double t = 0;
    static ProveSpans& get() { static thread_local ProveSpans s; return s; }
    void start() { for (double& v : ms) v = 0; t = LayerClock::now(); }
    void end(int i) { const double n = LayerClock::now(); ms[i] += (n - t) / 1e3; t = n; }
};

int32_t pippenger_prove(const gm_pippenger_wg* st, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                        const uint64_t* d_kn_inverses, const uint64_t* h_k, Tape* tr, uint64_t* h_pair) {
    const gm_msm_plan* plan = st->plan;
    const uint32_t x_log = plan->x_log, d_log = plan->d_log, y_log = st->y_log, y_size = plan->y_size, clm = st->clm;
    const uint32_t n_mat = st->n_mat;
    hipStream_t s = st->stream;
    void* stream = reinterpret_cast<void*>(s);
    GM_REQUIRE(y_log >= clm, "commitment_log_multiplicity exceeds y_logsize");
    StageTimer timer("prove", s);
    auto stage = [&](const char* name) { timer.mark(name); };
    ProveSpans::get().start();
    // phase-1 commitments onto the transcript (pippenger.rs:131-136)
    tr->write_points(st->comm_c.data(), n_mat);
    tr->write_points(st->comm_d.data(), n_mat);
    tr->write_points(st->comm_p0, 1);
    tr->write_points(st->comm_p1, 1);
    tr->write_points(st->comm_ac_c, 1);
    tr->write_points(st->comm_ac_d, 1);
    // prove image part
    Claims c;
    c.point.resize(y_log);
    memcpy(c.point.data(), h_claim_point, 32 * (size_t)y_log);
    c.evs.resize(3 * (d_log + 1));
    memcpy(c.evs.data(), h_claim_evs, 32 * c.evs.size());
    TRY(image_part_core(st->w, tr, &c));
    stage("image part");
    ProveSpans::get().end(0);
    // commit phase 2 (second_phase, pushforward.rs:596-605): msm_nonaff of the outer buckets with the eq tables
    std::vector<uint64_t> comm_cp(12 * (size_t)n_mat), comm_dp(12 * (size_t)n_mat);
    {
        const uint64_t X = 1ull << x_log, D = 1ull << d_log;
        DevBuf eqs;
        TRY(eqs.alloc((2 * X + 2 * D) * sizeof(Fr)));
        Fr* eq_c = eqs.fr();
        Fr* eq_d = eqs.fr() + 2 * X;
        std::vector<Fr*> lv(x_log + 1);
        for (uint32_t i = 0; i < x_log; i++) lv[i] = eq_c + X + ((1ull << i) - 1);
        lv[x_log] = eq_c;
        TRY(launch_eq_sequence(fr_one(), c.point.data() + y_log + d_log, x_log, lv.data(), s));
        lv.assign(d_log + 1, nullptr);
        for (uint32_t i = 0; i < d_log; i++) lv[i] = eq_d + D + ((1ull << i) - 1);
        lv[d_log] = eq_d;
        TRY(launch_eq_sequence(fr_one(), c.point.data() + y_log, d_log, lv.data(), s));
        // every outer-bucket array against the same eq table: one grouped MSM per kind
        std::vector<uint32_t> nd_all(n_mat, (uint32_t)D);
        TRY(gm_g1_msm_nonaff_grouped(reinterpret_cast<const uint64_t*>(st->d_outer->p), D, nd_all.data(), n_mat,
                                     reinterpret_cast<const uint64_t*>(eq_d), 1, 255, comm_dp.data(), stream));
        // eq_c[..basis.len()] with basis = c_outer_buckets[m] of length c_upper_bound[m] (pushforward.rs:601-604)
        TRY(gm_g1_msm_nonaff_grouped(reinterpret_cast<const uint64_t*>(st->c_outer->p), st->c_stride, st->c_upper.data(), n_mat,
                                     reinterpret_cast<const uint64_t*>(eq_c), 1, 255, comm_cp.data(), stream));
    }
    tr->write_points(comm_cp.data(), n_mat);
    tr->write_points(comm_dp.data(), n_mat);
    stage("phase-2 commitments");
    ProveSpans::get().end(1);
    // prove pushforward
    Fr gamma;
    Claims mx, acc, acd;
    PfCols cols;
    TRY(pushforward_prove(plan, st->d_points_xy, y_log, reinterpret_cast<const uint64_t*>(c.point.data()),
                          reinterpret_cast<const uint64_t*>(c.evs.data()), tr, &gamma, &mx, &acc, &acd, s, &cols));
    stage("pushforward");
    ProveSpans::get().end(2);
    // ---- open (pippenger.rs:162-286)
    const Fr p_folded_ev = mx.evs[0], c_pull_ev = mx.evs[1], d_pull_ev = mx.evs[2], c_ev = mx.evs[3], d_ev = mx.evs[4];
    const uint32_t nv = x_log + clm;
    const uint64_t n = 1ull << nv, X = 1ull << x_log;
    std::vector<Fr> pts(4 * (size_t)nv, fr_zero());   // the four claim points, nv coordinates each
    for (uint32_t i = 0; i < x_log; i++) pts[0 * nv + clm + i] = mx.point[y_log + i];            // p_folded_point
    for (uint32_t i = 0; i < x_log; i++) pts[1 * nv + clm + i] = acc.point[i];                    // ac_c_point
    for (uint32_t i = 0; i < d_log; i++) pts[2 * nv + (nv - d_log) + i] = acd.point[i];           // ac_d_point
    for (uint32_t i = 0; i < nv; i++) pts[3 * nv + i] = mx.point[y_log - clm + i];                // combined_opening_point
    // multirow_evs = EqPoly(y_log - clm, matrix_pt[..y_log - clm]).evals()
    std::vector<Fr> multirow(1ull << (y_log - clm), fr_zero());
    multirow[0] = fr_one();
    for (uint32_t i = 0; i < y_log - clm; i++)
        for (uint64_t j = (1ull << i); j-- > 0;) {
            const Fr w = multirow[j], m = fr_mul(mx.point[i], w);
            multirow[2 * j] = fr_sub(w, m);
            multirow[2 * j + 1] = m;
        }
    // sum_m multirow_evs[m] * commitment[m] (pippenger.rs:191-194): small MSMs on the device
    const uint32_t n_comb = n_mat < multirow.size() ? n_mat : (uint32_t)multirow.size();
    DevBuf d_mr, d_cs;
    TRY(d_mr.alloc(n_comb * sizeof(Fr)));
    TRY(d_cs.alloc((size_t)n_comb * sizeof(G1Aff)));
    GM_HIP(hipMemcpyAsync(d_mr.p, multirow.data(), n_comb * sizeof(Fr), hipMemcpyHostToDevice, s));
    auto comb = [&](const uint64_t* cs, G1Jac* out) -> int32_t {
        uint64_t r12[12];
        GM_HIP(hipMemcpyAsync(d_cs.p, cs, (size_t)n_comb * sizeof(G1Aff), hipMemcpyHostToDevice, s));
        const int32_t rc = gm_g1_msm((const uint64_t*)d_cs.p, (const uint64_t*)d_mr.p, n_comb, 1, 255, r12, stream);
        if (rc) return rc;
        *out = pp_aff_in(r12);
        return GM_OK;
    };
    G1Jac c_comb, d_comb, cp_comb, dp_comb;
    TRY(comb(st->comm_c.data(), &c_comb));
    TRY(comb(st->comm_d.data(), &d_comb));
    TRY(comb(comm_cp.data(), &cp_comb));
    TRY(comb(comm_dp.data(), &dp_comb));
    Fr u;
    TRY(tr->challenge(&u, 512));   // transcript.challenge(512) (pippenger.rs:197)
    Us us;
    us.u[0] = fr_one(); us.u[1] = u; us.u[2] = fr_mul(u, u); us.u[3] = fr_mul(us.u[2], u);
    const G1Jac combined_comm = g1_add(g1_add(c_comb, pp_mul(d_comb, us.u[1])), g1_add(pp_mul(cp_comb, us.u[2]), pp_mul(dp_comb, us.u[3])));
    const Fr combined_ev = fr_add(fr_add(c_ev, fr_mul(d_ev, us.u[1])), fr_add(fr_mul(c_pull_ev, us.u[2]), fr_mul(d_pull_ev, us.u[3])));
    // the four opening witnesses, zero-padded to 2^nv
    DevBuf w0, w1, w2, w3, d_multirow, folded;
    TRY(w0.alloc(n * sizeof(Fr))); TRY(w1.alloc(n * sizeof(Fr))); TRY(w2.alloc(n * sizeof(Fr))); TRY(w3.alloc(n * sizeof(Fr)));
    TRY(folded.alloc(n * sizeof(Fr)));
    for (DevBuf* b : {&w0, &w1, &w2}) GM_HIP(hipMemsetAsync(b->p, 0, n * sizeof(Fr), s));
    hipLaunchKernelGGL(k_pp_axpy, dim3(ceil_div(X, 256)), dim3(256), 0, s, st->p0->fr(), st->p1->fr(), gamma, X, w0.fr());
    GM_LAUNCH_CHECK();
    GM_HIP(hipMemcpyAsync(w1.p, cols.ac_c->p, X * sizeof(Fr), hipMemcpyDeviceToDevice, s));
    GM_HIP(hipMemcpyAsync(w2.p, cols.ac_d->p, ((size_t)1 << d_log) * sizeof(Fr), hipMemcpyDeviceToDevice, s));
    TRY(d_multirow.alloc(multirow.size() * sizeof(Fr)));
    GM_HIP(hipMemcpyAsync(d_multirow.p, multirow.data(), multirow.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pp_combined, dim3(ceil_div(n, 256)), dim3(256), 0, s, cols.c->fr(), cols.d->fr(), cols.c_pull->fr(),
                       cols.d_pull->fr(), d_multirow.fr(), us, x_log, y_size, clm, n, w3.fr());
    GM_LAUNCH_CHECK();
    GM_HIP(hipStreamSynchronize(s));  // multirow (host) is consumed
    stage("opening witnesses");
    ProveSpans::get().end(3);
    // MultiOpenReduction (pippenger.rs:224-258)
    std::vector<Fr> mo_evs = {fr_sub(p_folded_ev, fr_mul(gamma, gamma)), acc.evs[0], acd.evs[0], combined_ev};
    const uint64_t* wcols[4] = {(const uint64_t*)w0.p, (const uint64_t*)w1.p, (const uint64_t*)w2.p, (const uint64_t*)w3.p};
    std::vector<Fr> mo_pt, mo_out;
    TRY(multiopen_core(tr, nv, 4, wcols, reinterpret_cast<const uint64_t*>(pts.data()), reinterpret_cast<const uint64_t*>(mo_evs.data()),
                       &mo_pt, &mo_out, s));
    stage("multi-open reduction");
    ProveSpans::get().end(4);
    Fr q;
    TRY(tr->challenge(&q));
    Us qs;
    qs.u[0] = fr_one(); qs.u[1] = q; qs.u[2] = fr_mul(q, q); qs.u[3] = fr_mul(qs.u[2], q);
    const G1Jac parts[4] = {g1_add(pp_aff_in(st->comm_p0), pp_mul(pp_aff_in(st->comm_p1), gamma)), pp_aff_in(st->comm_ac_c),
                            pp_aff_in(st->comm_ac_d), combined_comm};
    G1Jac folded_comm = g1_inf();
    for (int j = 0; j < 4; j++) folded_comm = g1_add(folded_comm, pp_mul(parts[j], qs.u[j]));
    Cols4 c4;
    c4.p[0] = w0.fr(); c4.p[1] = w1.fr(); c4.p[2] = w2.fr(); c4.p[3] = w3.fr();
    hipLaunchKernelGGL(k_pp_fold4, dim3(ceil_div(n, 256)), dim3(256), 0, s, c4, qs, n, folded.fr());
    GM_LAUNCH_CHECK();
    // ev = gamma_rlc(q, multiopen_claims.evs)
    Fr open_ev = mo_out[3];
    for (int j = 2; j >= 0; j--) open_ev = fr_add(fr_mul(open_ev, q), mo_out[j]);
    uint64_t fc_aff[12], proof[48];
    pp_aff_out(fc_aff, folded_comm);
    gm_transcript adapter{tr, tape_ws, tape_ch, tape_wp};
    TRY(gm_knuckles_open_tr(st->d_basis, d_kn_inverses, h_k, nv, reinterpret_cast<const uint64_t*>(folded.p), n,
                            reinterpret_cast<const uint64_t*>(mo_pt.data()), reinterpret_cast<const uint64_t*>(&open_ev), fc_aff,
                            &adapter, proof, h_pair, stream));
    stage("knuckles open");
    ProveSpans::get().end(5);
    if (tr->cb_rc) return set_err(GM_ERR_STATE, "transcript callback failed with %d", tr->cb_rc);
    return GM_OK;
}

}  // namespace

extern "C" int32_t gm_pippenger_wg_create(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                          uint32_t commitment_log_multiplicity, const uint64_t* d_kzg_basis_aff,
                                          gm_pippenger_wg** out, void* stream) {
    GM_REQUIRE(plan && d_points_xy && d_kzg_basis_aff && out, "null argument");
    GM_REQUIRE(commitment_log_multiplicity <= y_logsize, "commitment_log_multiplicity exceeds y_logsize");
    hipStream_t s = as_stream(stream);
    std::unique_ptr<gm_pippenger_wg> st(new gm_pippenger_wg());
    st->plan = plan; st->d_points_xy = d_points_xy; st->d_basis = d_kzg_basis_aff; st->y_log = y_logsize;
    st->clm = commitment_log_multiplicity; st->stream = s;
    const uint32_t cm = 1u << st->clm;
    st->n_mat = (plan->y_size + cm - 1) / cm;
    StageTimer wt("wg::new", s);
    TRY(pip_witness_create(plan, d_points_xy, y_logsize, nullptr, &st->w, stream));
    wt.mark("witness");
    // outer buckets + c / d commitments
    {
        std::vector<uint32_t> rl(plan->nrows);
        GM_HIP(hipMemcpyAsync(rl.data(), plan->row_len, (size_t)plan->nrows * 4, hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
        uint32_t cmax = 1;
        for (uint32_t v : rl) cmax = v > cmax ? v : cmax;
        st->c_upper.assign(st->n_mat, 1);
        for (uint32_t r = 0; r < plan->nrows; r++) {
            const uint32_t m = (r >> plan->d_log) >> st->clm;
            if (rl[r] > st->c_upper[m]) st->c_upper[m] = rl[r];
        }
        st->d_outer.reset(new DevBuf());
        st->c_outer.reset(new DevBuf());
        TRY(st->d_outer->alloc(((size_t)st->n_mat << plan->d_log) * sizeof(G1Jac)));
        TRY(st->c_outer->alloc((size_t)st->n_mat * cmax * sizeof(G1Jac)));
        st->comm_c.assign(12 * (size_t)st->n_mat, 0);
        st->comm_d.assign(12 * (size_t)st->n_mat, 0);
        TRY(gm_msm_g1_outer(plan, d_kzg_basis_aff, st->clm, (uint64_t*)st->d_outer->p, (uint64_t*)st->c_outer->p,
                            (uint64_t)st->n_mat * cmax, &st->c_stride, st->comm_d.data(), st->comm_c.data(), stream));
    }
    wt.mark("outer buckets + c/d comm");
    // p_0, p_1, ac_c, ac_d commitments (pushforward.rs:533-536)
    const uint64_t X = plan->N, D = 1ull << plan->d_log;
    st->p0.reset(new DevBuf());
    st->p1.reset(new DevBuf());
    TRY(st->p0->alloc(X * sizeof(Fr)));
    TRY(st->p1->alloc(X * sizeof(Fr)));
    hipLaunchKernelGGL(k_pp_split_xy, dim3(ceil_div(X, 256)), dim3(256), 0, s, reinterpret_cast<const Fr*>(d_points_xy), X, st->p0->fr(),
                       st->p1->fr());
    GM_LAUNCH_CHECK();
    {
        DevBuf c, d, ac_c, ac_d;
        const uint64_t msize = (uint64_t)plan->y_size * X;
        TRY(c.alloc(msize * sizeof(Fr))); TRY(d.alloc(msize * sizeof(Fr))); TRY(ac_c.alloc(X * sizeof(Fr))); TRY(ac_d.alloc(D * sizeof(Fr)));
        TRY(gm_msm_phase1_polys(plan, (uint64_t*)c.p, (uint64_t*)d.p, (uint64_t*)ac_c.p, (uint64_t*)ac_d.p, stream));
        wt.mark("phase-1 polys");
        TRY(gm_g1_msm(d_kzg_basis_aff, (const uint64_t*)st->p0->p, X, 1, 255, st->comm_p0, stream));
        TRY(gm_g1_msm(d_kzg_basis_aff, (const uint64_t*)st->p1->p, X, 1, 255, st->comm_p1, stream));
        wt.mark("p_0, p_1 commitments");
        // ac_c / ac_d are negated access counts (pushforward.rs:507-508): commit(-v) = -commit(v) with v < 2^32, so the MSM
        // runs over 32-bit scalars and the result is negated -- the same group element at a sixth of the work
        auto commit_negated_counts = [&](DevBuf& col, uint64_t len, uint64_t* out12) -> int32_t {
            TRY(gm_fr_batch(3, (const uint64_t*)col.p, nullptr, (uint64_t*)col.p, len, stream));   // in place: -(-count) = count
            TRY(gm_g1_msm(d_kzg_basis_aff, (const uint64_t*)col.p, len, 1, 32, out12, stream));
            G1Aff a;
            memcpy(&a, out12, sizeof(G1Aff));
            if (!g1_aff_is_inf(a)) a = g1_aff_neg(a);
            memcpy(out12, &a, sizeof(G1Aff));
            return GM_OK;
        };
        TRY(commit_negated_counts(ac_c, X, st->comm_ac_c));
        TRY(commit_negated_counts(ac_d, D, st->comm_ac_d));
        wt.mark("ac_c, ac_d commitments");
    }
    *out = st.release();
    return GM_OK;
}


// =================================================================================================================
// The whole gen-2 prover with the matrix sharded by windows (BASELINE.json configs[4]).  What the unsharded calls above
// do with the whole KZG key, a rank does with the key ranges it holds (gm_key_view) -- every G1 step of the protocol is linear in
// the committed column, so a commitment is the ranks' partial MSMs combined (gm_g1_combine_parts: one group element per rank):
//   c / d commitments, phase-2 commitments   partial outer buckets of the rank's windows (gm_msm_g1_outer_part)
//   p_0, p_1, ac_c, ac_d                     columns every rank holds whole: rank r commits entries [r L / G, (r + 1) L / G)
//   the opening                              the four witnesses as contiguous slices of 2^(x + clm) / G, the multi-open reduction on the
//                                            sharded dense object, the Knuckles opening on slices (knuckles.hip)
// Every rank runs the same transcript and ends with the same proof and pairing pair as the unsharded prover.
namespace {

// this rank's share of the commitment of a column every rank holds whole
int32_t commit_replicated_part(const gm_pippenger_wg* st, const Fr* d_col, uint64_t L, uint32_t nbits, G1Jac* part, const char* what, void* stream) {
    const uint32_t G = st->sh.world;
    uint64_t lo = 0, cnt = st->sh.rank == 0 ? L : 0;
    if (L >= G) { cnt = L / G; lo = (uint64_t)st->sh.rank * cnt; }
    *part = g1_inf();
    if (!cnt) return GM_OK;
    GM_KEY_RANGE(kp, &st->key, lo, cnt, what);
    uint64_t a12[12];
    TRY(gm_g1_msm(kp, reinterpret_cast<const uint64_t*>(d_col + lo), cnt, 1, nbits, a12, stream));
    *part = pp_aff_in(a12);
    return GM_OK;
}

int32_t pippenger_prove_sharded(const gm_pippenger_wg* st, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                const uint64_t* d_kn_inverses_slice, const uint64_t* h_k, Tape* tr, uint64_t* h_pair) {
    const gm_msm_plan* plan = st->plan;
    const Shard& sh = st->sh;
    const uint32_t x_log = plan->x_log, d_log = plan->d_log, y_log = st->y_log, clm = st->clm, G = sh.world;
    const uint32_t n_mat = st->n_mat, cm = 1u << clm;
    hipStream_t s = st->stream;
    void* stream = reinterpret_cast<void*>(s);
    GM_REQUIRE(y_log >= clm, "commitment_log_multiplicity exceeds y_logsize");
    StageTimer timer("prove (sharded)", s);
    auto stage = [&](const char* name) { timer.mark(name); };
    ProveSpans::get().start();
    tr->write_points(st->comm_c.data(), n_mat);
    tr->write_points(st->comm_d.data(), n_mat);
    tr->write_points(st->comm_p0, 1);
    tr->write_points(st->comm_p1, 1);
    tr->write_points(st->comm_ac_c, 1);
    tr->write_points(st->comm_ac_d, 1);
    Claims c;
    c.point.resize(y_log);
    memcpy(c.point.data(), h_claim_point, 32 * (size_t)y_log);
    c.evs.resize(3 * (d_log + 1));
    memcpy(c.evs.data(), h_claim_evs, 32 * c.evs.size());
    TRY(image_part_core(st->w, tr, &c));
    stage("image part");
    ProveSpans::get().end(0);
    // commit phase 2: the eq-weighted MSMs over this rank's PARTIAL outer buckets, combined
    std::vector<uint64_t> comm_cp(12 * (size_t)n_mat), comm_dp(12 * (size_t)n_mat);
    {
        const uint64_t X = 1ull << x_log, D = 1ull << d_log;
        DevBuf eqs;
        TRY(eqs.alloc((2 * X + 2 * D) * sizeof(Fr)));
        Fr* eq_c = eqs.fr();
        Fr* eq_d = eqs.fr() + 2 * X;
        std::vector<Fr*> lv(x_log + 1);
        for (uint32_t i = 0; i < x_log; i++) lv[i] = eq_c + X + ((1ull << i) - 1);
        lv[x_log] = eq_c;
        TRY(launch_eq_sequence(fr_one(), c.point.data() + y_log + d_log, x_log, lv.data(), s));
        lv.assign(d_log + 1, nullptr);
        for (uint32_t i = 0; i < d_log; i++) lv[i] = eq_d + D + ((1ull << i) - 1);
        lv[d_log] = eq_d;
        TRY(launch_eq_sequence(fr_one(), c.point.data() + y_log, d_log, lv.data(), s));
        const uint32_t nl = st->n_mat_loc;
        std::vector<uint32_t> nd_all(nl, (uint32_t)D);
        std::vector<uint64_t> loc_d(12 * (size_t)nl), loc_c(12 * (size_t)nl);
        TRY(gm_g1_msm_nonaff_grouped(reinterpret_cast<const uint64_t*>(st->d_outer->p), D, nd_all.data(), nl,
                                     reinterpret_cast<const uint64_t*>(eq_d), 1, 255, loc_d.data(), stream));
        TRY(gm_g1_msm_nonaff_grouped(reinterpret_cast<const uint64_t*>(st->c_outer->p), st->c_stride, st->c_upper.data(), nl,
                                     reinterpret_cast<const uint64_t*>(eq_c), 1, 255, loc_c.data(), stream));
        std::vector<G1Jac> parts(2 * (size_t)n_mat, g1_inf());
        for (uint32_t i = 0; i < nl; i++) {
            parts[st->m0 + i] = pp_aff_in(loc_c.data() + 12 * (size_t)i);
            parts[n_mat + st->m0 + i] = pp_aff_in(loc_d.data() + 12 * (size_t)i);
        }
        std::vector<uint64_t> both(24 * (size_t)n_mat);
        TRY(gm_g1_combine_parts(sh.comm, reinterpret_cast<const uint64_t*>(parts.data()), 2 * n_mat, both.data()));
        memcpy(comm_cp.data(), both.data(), 96 * (size_t)n_mat);
        memcpy(comm_dp.data(), both.data() + 12 * (size_t)n_mat, 96 * (size_t)n_mat);
    }
    tr->write_points(comm_cp.data(), n_mat);
    tr->write_points(comm_dp.data(), n_mat);
    stage("phase-2 commitments");
    ProveSpans::get().end(1);
    Fr gamma;
    Claims mx, acc, acd;
    PfCols cols;
    TRY(pushforward_prove_sharded(plan, st->d_points_xy, y_log, sh, reinterpret_cast<const uint64_t*>(c.point.data()),
                                  reinterpret_cast<const uint64_t*>(c.evs.data()), tr, &gamma, &mx, &acc, &acd, s, &cols));
    stage("pushforward");
    ProveSpans::get().end(2);
    // ---- open (pippenger.rs:162-286) on slices
    const Fr p_folded_ev = mx.evs[0], c_pull_ev = mx.evs[1], d_pull_ev = mx.evs[2], c_ev = mx.evs[3], d_ev = mx.evs[4];
    const uint32_t nv = x_log + clm;
    GM_REQUIRE(nv >= sh.lg, "more ranks than opening-witness entries");
    const uint64_t n = 1ull << nv, X = 1ull << x_log, D = 1ull << d_log, SL = n / G, base = (uint64_t)sh.rank * SL;
    std::vector<Fr> pts(4 * (size_t)nv, fr_zero());
    for (uint32_t i = 0; i < x_log; i++) pts[0 * nv + clm + i] = mx.point[y_log + i];
    for (uint32_t i = 0; i < x_log; i++) pts[1 * nv + clm + i] = acc.point[i];
    for (uint32_t i = 0; i < d_log; i++) pts[2 * nv + (nv - d_log) + i] = acd.point[i];
    for (uint32_t i = 0; i < nv; i++) pts[3 * nv + i] = mx.point[y_log - clm + i];
    std::vector<Fr> multirow(1ull << (y_log - clm), fr_zero());
    multirow[0] = fr_one();
    for (uint32_t i = 0; i < y_log - clm; i++)
        for (uint64_t j = (1ull << i); j-- > 0;) {
            const Fr w = multirow[j], m = fr_mul(mx.point[i], w);
            multirow[2 * j] = fr_sub(w, m);
            multirow[2 * j + 1] = m;
        }
    // sum_m multirow_evs[m] * commitment[m]: n_mat <= 2^(y_log - clm) points, the same small MSM on every rank
    const uint32_t n_comb = n_mat < multirow.size() ? n_mat : (uint32_t)multirow.size();
    DevBuf d_mr, d_cs;
    TRY(d_mr.alloc(n_comb * sizeof(Fr)));
    TRY(d_cs.alloc((size_t)n_comb * sizeof(G1Aff)));
    GM_HIP(hipMemcpyAsync(d_mr.p, multirow.data(), n_comb * sizeof(Fr), hipMemcpyHostToDevice, s));
    auto comb = [&](const uint64_t* cs, G1Jac* out) -> int32_t {
        uint64_t r12[12];
        GM_HIP(hipMemcpyAsync(d_cs.p, cs, (size_t)n_comb * sizeof(G1Aff), hipMemcpyHostToDevice, s));
        const int32_t rc = gm_g1_msm((const uint64_t*)d_cs.p, (const uint64_t*)d_mr.p, n_comb, 1, 255, r12, stream);
        if (rc) return rc;
        *out = pp_aff_in(r12);
        return GM_OK;
    };
    G1Jac c_comb, d_comb, cp_comb, dp_comb;
    TRY(comb(st->comm_c.data(), &c_comb));
    TRY(comb(st->comm_d.data(), &d_comb));
    TRY(comb(comm_cp.data(), &cp_comb));
    TRY(comb(comm_dp.data(), &dp_comb));
    Fr u;
    TRY(tr->challenge(&u, 512));
    Us us;
    us.u[0] = fr_one(); us.u[1] = u; us.u[2] = fr_mul(u, u); us.u[3] = fr_mul(us.u[2], u);
    const G1Jac combined_comm = g1_add(g1_add(c_comb, pp_mul(d_comb, us.u[1])), g1_add(pp_mul(cp_comb, us.u[2]), pp_mul(dp_comb, us.u[3])));
    const Fr combined_ev = fr_add(fr_add(c_ev, fr_mul(d_ev, us.u[1])), fr_add(fr_mul(c_pull_ev, us.u[2]), fr_mul(d_pull_ev, us.u[3])));
    // the four opening witnesses: this rank's slice [base, base + SL) of each (zero-padded to 2^nv)
    DevBuf w0, w1, w2, w3, d_multirow, folded;
    TRY(w0.alloc(SL * sizeof(Fr))); TRY(w1.alloc(SL * sizeof(Fr))); TRY(w2.alloc(SL * sizeof(Fr))); TRY(w3.alloc(SL * sizeof(Fr)));
    {
        ExportableScope exported;   // the Knuckles opening re-spreads it over the ranks (dist_read)
        TRY(folded.alloc(SL * sizeof(Fr)));
    }
    hipLaunchKernelGGL(k_pp_axpy_slice, dim3(ceil_div(SL, 256)), dim3(256), 0, s, st->p0->fr(), st->p1->fr(), gamma, X, base, SL, w0.fr());
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pp_pad_slice, dim3(ceil_div(SL, 256)), dim3(256), 0, s, cols.ac_c_p, X, base, SL, w1.fr());
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pp_pad_slice, dim3(ceil_div(SL, 256)), dim3(256), 0, s, cols.ac_d_p, D, base, SL, w2.fr());
    GM_LAUNCH_CHECK();
    TRY(d_multirow.alloc(multirow.size() * sizeof(Fr)));
    GM_HIP(hipMemcpyAsync(d_multirow.p, multirow.data(), multirow.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
    bool host_staged = false;
    {
        // combined_witness (pippenger.rs:208-222): a rank's windows give it a share over a contiguous index range of the witness --
        // [ (y0 mod cm) X, + W X ) for W = y_size / G < cm windows per rank, the whole of it otherwise -- and the slice of rank r
        // is the sum of the shares that cover it (y_size / cm of them, or all G): pulled side by side, then added.
        const uint32_t W = plan->nwin;
        const bool narrow = W < cm;
        const uint32_t rem0 = narrow ? (plan->y0 & (cm - 1)) : 0;
        const uint64_t len_g = (uint64_t)(narrow ? W : cm) << x_log;
        DevBuf share, stage_buf;
        {
            ExportableScope exported;
            TRY(share.alloc(len_g * sizeof(Fr)));
        }
        hipLaunchKernelGGL(k_pp_combined_part, dim3(ceil_div(len_g, 256)), dim3(256), 0, s, cols.c->fr(), cols.d->fr(), cols.c_pull->fr(),
                           cols.d_pull->fr(), d_multirow.fr(), us, x_log, plan->y0, plan->y1, clm, rem0, len_g, share.fr());
        GM_LAUNCH_CHECK();
        std::vector<gm_pull> pc;
        std::vector<uint32_t> from;
        for (uint32_t g = 0; g < G; g++) {
            const uint64_t a_g = narrow ? ((uint64_t)((g * W) & (cm - 1)) << x_log) : 0;
            if (a_g <= base && base + SL <= a_g + len_g) from.push_back(g);
        }
        GM_REQUIRE(!from.empty(), "no rank covers this rank's slice of the combined witness");
        TRY(stage_buf.alloc((uint64_t)from.size() * SL * sizeof(Fr)));
        for (size_t k = 0; k < from.size(); k++) {
            const uint32_t g = from[k];
            const uint64_t a_g = narrow ? ((uint64_t)((g * W) & (cm - 1)) << x_log) : 0;
            pc.push_back(gm_pull{g, 0u, (base - a_g) * sizeof(Fr), SL * sizeof(Fr), stage_buf.fr() + k * SL});
        }
        TRY(shard_pull(sh, share.fr(), len_g, pc, &host_staged, s));
        hipLaunchKernelGGL(k_pf_sum_parts, dim3(ceil_div(SL, 256)), dim3(256), 0, s, stage_buf.fr(), (uint32_t)from.size(), SL, w3.fr());
        GM_LAUNCH_CHECK();
        GM_HIP(hipStreamSynchronize(s));   // share / stage_buf go out of scope; multirow (host) is consumed
    }
    stage("opening witnesses");
    ProveSpans::get().end(3);
    std::vector<Fr> mo_evs = {fr_sub(p_folded_ev, fr_mul(gamma, gamma)), acc.evs[0], acd.evs[0], combined_ev};
    const uint64_t* wcols[4] = {(const uint64_t*)w0.p, (const uint64_t*)w1.p, (const uint64_t*)w2.p, (const uint64_t*)w3.p};
    std::vector<Fr> mo_pt, mo_out;
    TRY(multiopen_core(tr, nv, 4, wcols, reinterpret_cast<const uint64_t*>(pts.data()), reinterpret_cast<const uint64_t*>(mo_evs.data()),
                       &mo_pt, &mo_out, s, sh));
    stage("multi-open reduction");
    ProveSpans::get().end(4);
    Fr q;
    TRY(tr->challenge(&q));
    Us qs;
    qs.u[0] = fr_one(); qs.u[1] = q; qs.u[2] = fr_mul(q, q); qs.u[3] = fr_mul(qs.u[2], q);
    const G1Jac parts[4] = {g1_add(pp_aff_in(st->comm_p0), pp_mul(pp_aff_in(st->comm_p1), gamma)), pp_aff_in(st->comm_ac_c),
                            pp_aff_in(st->comm_ac_d), combined_comm};
    G1Jac folded_comm = g1_inf();
    for (int j = 0; j < 4; j++) folded_comm = g1_add(folded_comm, pp_mul(parts[j], qs.u[j]));
    Cols4 c4;
    c4.p[0] = w0.fr(); c4.p[1] = w1.fr(); c4.p[2] = w2.fr(); c4.p[3] = w3.fr();
    hipLaunchKernelGGL(k_pp_fold4, dim3(ceil_div(SL, 256)), dim3(256), 0, s, c4, qs, SL, folded.fr());
    GM_LAUNCH_CHECK();
    Fr open_ev = mo_out[3];
    for (int j = 2; j >= 0; j--) open_ev = fr_add(fr_mul(open_ev, q), mo_out[j]);
    uint64_t fc_aff[12], proof[48];
    pp_aff_out(fc_aff, folded_comm);
    gm_transcript adapter{tr, tape_ws, tape_ch, tape_wp};
    TRY(gm_knuckles_open_sharded_tr(sh.comm, &st->key, d_kn_inverses_slice, h_k, nv, reinterpret_cast<const uint64_t*>(folded.p),
                                    reinterpret_cast<const uint64_t*>(mo_pt.data()), reinterpret_cast<const uint64_t*>(&open_ev), fc_aff,
                                    &adapter, proof, h_pair, stream));
    stage("knuckles open");
    ProveSpans::get().end(5);
    if (tr->cb_rc) return set_err(GM_ERR_STATE, "transcript callback failed with %d", tr->cb_rc);
    return GM_OK;
}

}  // namespace

// PippengerWG::new for one rank of a window-sharded proof (collective).  plan: this rank's windows (gm_msm_plan_create(.., y_begin,
// y_end), after gm_msm_run); key: the ranges of kzg_basis() resident on this rank -- gm_pippenger_sharded_key_ranges lists what it
// must cover; comm must outlive the handle.  The commitments the handle carries are the combined ones, equal on every rank to those
// of gm_pippenger_wg_create over the whole key.
extern "C" int32_t gm_pippenger_wg_create_sharded(const gm_msm_plan* plan, const uint64_t* d_points_xy, uint32_t y_logsize,
                                                  uint32_t commitment_log_multiplicity, const gm_key_view* key, const gm_comm* comm,
                                                  gm_pippenger_wg** out, void* stream) {
    GM_REQUIRE(plan && d_points_xy && key && comm && out, "null argument");
    GM_REQUIRE(commitment_log_multiplicity <= y_logsize && commitment_log_multiplicity <= 8, "bad commitment_log_multiplicity");
    GM_REQUIRE(comm->world >= 2, "a sharded proof needs at least two ranks (gm_pippenger_wg_create otherwise)");
    GM_REQUIRE(key->n_segments >= 1 && key->d_segment && key->first && key->count, "empty gm_key_view");
    hipStream_t s = as_stream(stream);
    std::unique_ptr<gm_pippenger_wg> st(new gm_pippenger_wg());
    st->plan = plan; st->d_points_xy = d_points_xy; st->d_basis = nullptr; st->y_log = y_logsize;
    st->clm = commitment_log_multiplicity; st->stream = s;
    for (uint32_t i = 0; i < key->n_segments; i++) {
        st->key_seg.push_back(key->d_segment[i]); st->key_first.push_back(key->first[i]); st->key_count.push_back(key->count[i]);
    }
    st->key = gm_key_view{key->n_segments, 0, st->key_seg.data(), st->key_first.data(), st->key_count.data()};
    const uint32_t clm = st->clm, cm = 1u << clm, d_log = plan->d_log;
    st->n_mat = (plan->y_size + cm - 1) / cm;
    StageTimer wt("wg::new (sharded)", s);
    TRY(pip_witness_create(plan, d_points_xy, y_logsize, comm, &st->w, stream));
    st->sh = st->w->sh;
    const Shard& sh = st->sh;
    const uint32_t G = sh.world;
    const uint64_t X = plan->N, D = 1ull << d_log;
    GM_REQUIRE(X >= G, "fewer points than ranks");
    wt.mark("witness");
    {   // partial outer buckets of this rank's windows + its share of the c / d commitments
        st->m0 = plan->y0 >> clm;
        st->n_mat_loc = ((plan->y1 - 1) >> clm) - st->m0 + 1;
        std::vector<uint32_t> rl(plan->nrows);
        GM_HIP(hipMemcpyAsync(rl.data(), plan->row_len, (size_t)plan->nrows * 4, hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
        uint32_t cmax = 1;
        for (uint32_t v : rl) cmax = v > cmax ? v : cmax;
        st->c_upper.assign(st->n_mat_loc, 1);
        for (uint32_t r = 0; r < plan->nrows; r++) {
            const uint32_t m = ((plan->y0 + (r >> d_log)) >> clm) - st->m0;
            if (rl[r] > st->c_upper[m]) st->c_upper[m] = rl[r];
        }
        // the key slices of the rank's windows: slice s = kzg_basis[s X, (s + 1) X) for s = y mod 2^clm; all in ONE resident segment,
        // at whole-slice distances from its start (gm_msm_g1_outer_part addresses them as slots of one buffer)
        int32_t h_slot[256];
        for (int i = 0; i < 256; i++) h_slot[i] = -1;
        const uint64_t* seg_ptr = nullptr;
        uint64_t seg_first = 0;
        for (uint32_t y = plan->y0; y < plan->y1; y++) {
            const uint32_t sl = y & (cm - 1);
            if (h_slot[sl] >= 0) continue;
            uint32_t found = key->n_segments;
            for (uint32_t i = 0; i < key->n_segments; i++)
                if ((uint64_t)sl * X >= key->first[i] && ((uint64_t)sl + 1) * X <= key->first[i] + key->count[i]) { found = i; break; }
            GM_REQUIRE(found < key->n_segments, "outer buckets: key slice %u (points [%llu, %llu)) is not resident on rank %u", sl,
                       (unsigned long long)((uint64_t)sl * X), (unsigned long long)(((uint64_t)sl + 1) * X), sh.rank);
            if (!seg_ptr) { seg_ptr = key->d_segment[found]; seg_first = key->first[found]; }
            GM_REQUIRE(key->d_segment[found] == seg_ptr, "outer buckets: the key slices of a rank's windows must lie in one segment");
            GM_REQUIRE(((uint64_t)sl * X - seg_first) % X == 0 && ((uint64_t)sl * X - seg_first) / X < 256,
                       "outer buckets: the segment must start at a multiple of 2^x_logsize points from its slices");
            h_slot[sl] = (int32_t)(((uint64_t)sl * X - seg_first) / X);
        }
        st->d_outer.reset(new DevBuf());
        st->c_outer.reset(new DevBuf());
        TRY(st->d_outer->alloc(((size_t)st->n_mat_loc << d_log) * sizeof(G1Jac)));
        TRY(st->c_outer->alloc((size_t)st->n_mat_loc * cmax * sizeof(G1Jac)));
        std::vector<G1Jac> dpart(st->n_mat_loc), cpart(st->n_mat_loc);
        uint32_t m0 = 0, nml = 0;
        TRY(gm_msm_g1_outer_part(plan, seg_ptr, h_slot, clm, (uint64_t*)st->d_outer->p, (uint64_t*)st->c_outer->p,
                                 (uint64_t)st->n_mat_loc * cmax, &st->c_stride, &m0, &nml, reinterpret_cast<uint64_t*>(dpart.data()),
                                 reinterpret_cast<uint64_t*>(cpart.data()), stream));
        GM_REQUIRE(m0 == st->m0 && nml == st->n_mat_loc, "outer buckets: unexpected matrix range");
        std::vector<G1Jac> parts(2 * (size_t)st->n_mat, g1_inf());
        for (uint32_t i = 0; i < nml; i++) {
            parts[st->m0 + i] = cpart[i];
            parts[st->n_mat + st->m0 + i] = dpart[i];
        }
        std::vector<uint64_t> both(24 * (size_t)st->n_mat);
        TRY(gm_g1_combine_parts(sh.comm, reinterpret_cast<const uint64_t*>(parts.data()), 2 * st->n_mat, both.data()));
        st->comm_c.assign(both.begin(), both.begin() + 12 * (size_t)st->n_mat);
        st->comm_d.assign(both.begin() + 12 * (size_t)st->n_mat, both.end());
    }
    wt.mark("outer buckets + c/d comm");
    st->p0.reset(new DevBuf());
    st->p1.reset(new DevBuf());
    TRY(st->p0->alloc(X * sizeof(Fr)));
    TRY(st->p1->alloc(X * sizeof(Fr)));
    hipLaunchKernelGGL(k_pp_split_xy, dim3(ceil_div(X, 256)), dim3(256), 0, s, reinterpret_cast<const Fr*>(d_points_xy), X, st->p0->fr(),
                       st->p1->fr());
    GM_LAUNCH_CHECK();
    {
        DevBuf c, d, ac;
        const uint64_t ML = (uint64_t)plan->nwin * X;
        TRY(c.alloc(ML * sizeof(Fr))); TRY(d.alloc(ML * sizeof(Fr))); TRY(ac.alloc((X + D) * sizeof(Fr)));
        bool host_staged = false;
        TRY(sharded_access_counts(plan, sh, c.fr(), d.fr(), ac.fr(), &host_staged, s));
        wt.mark("access counts");
        // ac_c / ac_d are negated access counts (pushforward.rs:507-508): the MSMs run over the 32-bit counts, the results are negated
        TRY(gm_fr_batch(3, (const uint64_t*)ac.p, nullptr, (uint64_t*)ac.p, X + D, stream));
        G1Jac four[4];
        TRY(commit_replicated_part(st.get(), st->p0->fr(), X, 255, &four[0], "p_0 commitment", stream));
        TRY(commit_replicated_part(st.get(), st->p1->fr(), X, 255, &four[1], "p_1 commitment", stream));
        TRY(commit_replicated_part(st.get(), ac.fr(), X, 32, &four[2], "ac_c commitment", stream));
        TRY(commit_replicated_part(st.get(), ac.fr() + X, D, 32, &four[3], "ac_d commitment", stream));
        uint64_t out4[48];
        TRY(gm_g1_combine_parts(sh.comm, reinterpret_cast<const uint64_t*>(four), 4, out4));
        for (int j = 2; j < 4; j++) {
            G1Aff a;
            memcpy(&a, out4 + 12 * j, sizeof(G1Aff));
            if (!g1_aff_is_inf(a)) a = g1_aff_neg(a);
            memcpy(out4 + 12 * j, &a, sizeof(G1Aff));
        }
        memcpy(st->comm_p0, out4, 96); memcpy(st->comm_p1, out4 + 12, 96); memcpy(st->comm_ac_c, out4 + 24, 96); memcpy(st->comm_ac_d, out4 + 36, 96);
        wt.mark("p_0, p_1, ac_c, ac_d commitments");
    }
    *out = st.release();
    return GM_OK;
}

// What rank `rank` of `world` reads of kzg_basis() in a sharded proof of this shape: up to 4 ranges (first, count), for the caller to
// size its gm_key_view.  [0] the key slices of its windows' outer buckets (one contiguous run when its windows are), [1] its share of
// the short commitments (p_0, p_1, ac_c), [2] of ac_d, [3] its range of the opening (t and the two quotients).
extern "C" int32_t gm_pippenger_sharded_key_ranges(uint32_t x_logsize, uint32_t d_logsize, uint32_t y_logsize,
                                                   uint32_t commitment_log_multiplicity, uint32_t rank, uint32_t world, uint64_t* first4,
                                                   uint64_t* count4) {
    GM_REQUIRE(first4 && count4 && world >= 2 && (world & (world - 1)) == 0 && rank < world, "bad argument");
    GM_REQUIRE(commitment_log_multiplicity <= y_logsize && (1u << y_logsize) >= world, "bad shape");
    const uint64_t X = 1ull << x_logsize, D = 1ull << d_logsize, cm = 1ull << commitment_log_multiplicity;
    const uint64_t W = (1ull << y_logsize) / world, y0 = rank * W;
    if (W >= cm) { first4[0] = 0; count4[0] = cm * X; }
    else { first4[0] = (y0 & (cm - 1)) * X; count4[0] = W * X; }
    if (X >= world) { first4[1] = rank * (X / world); count4[1] = X / world; } else { first4[1] = 0; count4[1] = rank == 0 ? X : 0; }
    if (D >= world) { first4[2] = rank * (D / world); count4[2] = D / world; } else { first4[2] = 0; count4[2] = rank == 0 ? D : 0; }
    const uint64_t N = cm * X, total = 2 * N - 1, S = 2 * N / world, b = rank * S;
    first4[3] = b;
    count4[3] = b >= total ? 0 : (total - b < S ? total - b : S);
    return GM_OK;
}

extern "C" int32_t gm_pippenger_last_spans(double* out8) {
    GM_REQUIRE(out8, "null argument");
    memcpy(out8, ProveSpans::get().ms, 8 * sizeof(double));
    return GM_OK;
}

extern "C" int32_t gm_pippenger_wg_destroy(gm_pippenger_wg* st) {
    delete st;
    return GM_OK;
}

extern "C" int32_t gm_pippenger_wg_witness(const gm_pippenger_wg* st, const gm_pip_witness** w) {
    GM_REQUIRE(st && w, "null argument");
    *w = st->w;
    return GM_OK;
}

extern "C" int32_t gm_pippenger_prove(const gm_pippenger_wg* st, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                      const uint64_t* d_knuckles_inverses, const uint64_t* h_k, const uint64_t* h_tape,
                                      uint64_t n_tape, uint64_t* h_msgs, uint64_t msgs_cap, uint64_t* n_msgs, uint64_t* h_points,
                                      uint64_t points_cap, uint64_t* n_points, uint64_t* h_pair, uint64_t* tape_used,
                                      uint64_t* rounds) {
    GM_REQUIRE(st && h_claim_point && h_claim_evs && d_knuckles_inverses && h_k && h_tape && h_pair, "null argument");
    std::vector<Fr> msgs;
    std::vector<uint64_t> points;
    Tape tr{h_tape, n_tape, 0, &msgs, 0, nullptr, 0};
    tr.points = &points;
    if (st->sh.comm) TRY(pippenger_prove_sharded(st, h_claim_point, h_claim_evs, d_knuckles_inverses, h_k, &tr, h_pair));
    else TRY(pippenger_prove(st, h_claim_point, h_claim_evs, d_knuckles_inverses, h_k, &tr, h_pair));
    if (n_msgs) *n_msgs = msgs.size();
    if (h_msgs) {
        GM_REQUIRE(msgs.size() <= msgs_cap, "message buffer too small: %zu > %llu", msgs.size(), (unsigned long long)msgs_cap);
        memcpy(h_msgs, msgs.data(), msgs.size() * sizeof(Fr));
    }
    if (n_points) *n_points = points.size() / 12;
    if (h_points) {
        GM_REQUIRE(points.size() / 12 <= points_cap, "point buffer too small");
        memcpy(h_points, points.data(), points.size() * 8);
    }
    if (tape_used) *tape_used = tr.pos;
    if (rounds) *rounds = tr.rounds;
    return GM_OK;
}

extern "C" int32_t gm_pippenger_prove_tr(const gm_pippenger_wg* st, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                         const uint64_t* d_knuckles_inverses, const uint64_t* h_k, const gm_transcript* tr,
                                         uint64_t* h_pair, uint64_t* n_challenges, uint64_t* rounds) {
    GM_REQUIRE(st && h_claim_point && h_claim_evs && d_knuckles_inverses && h_k && tr && tr->challenge && h_pair, "null argument");
    std::vector<Fr> msgs;
    Tape t{nullptr, 0, 0, &msgs, 0, tr, 0};
    if (st->sh.comm) TRY(pippenger_prove_sharded(st, h_claim_point, h_claim_evs, d_knuckles_inverses, h_k, &t, h_pair));
    else TRY(pippenger_prove(st, h_claim_point, h_claim_evs, d_knuckles_inverses, h_k, &t, h_pair));
    if (n_challenges) *n_challenges = t.pos;
    if (rounds) *rounds = t.rounds;
    return GM_OK;
}
