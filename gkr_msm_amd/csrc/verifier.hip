// The gen-2 verifier, Pippenger::verify (/root/reference/src/cleanup/protocols/pippenger.rs:296-406), on the host.
// The reference's verifier is CPU code that reads the proof through the transcript; so is this: no device work, no oracle.
// Written from the reference's verify functions (cited at each step), not from this library's provers, so the order and the
// sizes of everything the provers write are checked against what the reference verifier reads.
//
// Transcript: a gm_transcript_reader (read_scalars / challenge / read_points; the Rust shim forwards to ProofTranscript2 in
// verifier mode, cleanup/proof_transcript.rs:46-49,59-62,120-131), or recorded messages plus a challenge tape (tests).
// A failed check returns GM_ERR_VERIFY with the failing assertion in gm_last_error() -- the reference panics there.
#include <cstring>
#include <vector>

#include "common.hpp"
#include "gkr_layers.hpp"
#include "pairing.hpp"

using namespace gm;

namespace {

#define TRY(x)                      \
    do {                            \
        int32_t rc__ = (x);         \
        if (rc__) return rc__;      \
    } while (0)

#define VERIFY(cond, ...)                                          \
    do {                                                           \
        if (!(cond)) return set_err(GM_ERR_VERIFY, __VA_ARGS__);   \
    } while (0)

struct Reader {
    const gm_transcript_reader* cb = nullptr;
    const Fr* scalars = nullptr;     // recorded mode: Montgomery scalars, affine points, canonical tape
    const G1Aff* points = nullptr;
    const uint64_t* tape = nullptr;
    uint64_t n_scalars = 0, n_points = 0, n_tape = 0, si = 0, pi = 0, pos = 0;

    int32_t read_scalars(uint64_t n, Fr* out) {
        if (cb) {
            const int32_t rc = cb->read_scalars(cb->ctx, n, reinterpret_cast<uint64_t*>(out));
            if (rc) return set_err(GM_ERR_VERIFY, "transcript read_scalars failed with %d (proof too short or malformed)", rc);
        } else {
            VERIFY(si + n <= n_scalars, "proof ran out of scalars (proof_transcript.rs:123 \"Out of bounds\")");
            memcpy(out, scalars + si, n * sizeof(Fr));
        }
        si += n;
        return GM_OK;
    }
    int32_t read_points(uint64_t n, G1Aff* out) {
        if (cb) {
            const int32_t rc = cb->read_points(cb->ctx, n, reinterpret_cast<uint64_t*>(out));
            if (rc) return set_err(GM_ERR_VERIFY, "transcript read_points failed with %d (proof too short or malformed)", rc);
        } else {
            VERIFY(pi + n <= n_points, "proof ran out of points (proof_transcript.rs:123 \"Out of bounds\")");
            memcpy(out, points + pi, n * sizeof(G1Aff));
        }
        for (uint64_t i = 0; i < n; i++) {
            VERIFY(g1_aff_on_curve(out[i]), "a proof point is not on the curve (deserialize_compressed)");
            // recorded mode holds raw affine points; a reader callback skips the ~70 us membership test only when it says its
            // deserializer has done it (gm_transcript_reader::points_validated; the built-in merlin reader does: g1_decompress)
            if (!cb || !cb->points_validated) VERIFY(g1_aff_in_subgroup_host(out[i]), "a proof point is outside the prime-order subgroup (deserialize_compressed, Validate::Yes)");
        }
        pi += n;
        return GM_OK;
    }
    // Montgomery challenges; challenge_vec(n, bits) is one squeeze (proof_transcript.rs:41-45)
    int32_t challenge(Fr* out, uint32_t cnt = 1, uint32_t bits = 128) {
        if (cb) {
            const int32_t rc = cb->challenge(cb->ctx, cnt, bits, reinterpret_cast<uint64_t*>(out));
            if (rc) return set_err(GM_ERR_STATE, "transcript challenge callback failed with %d", rc);
        } else {
            if (pos + cnt > n_tape) return set_err(GM_ERR_INVALID, "challenge tape exhausted after %llu challenges", (unsigned long long)pos);
            memcpy(out, tape + 4 * pos, 32 * (size_t)cnt);
        }
        for (uint32_t i = 0; i < cnt; i++) out[i] = fr_to_mont(out[i]);
        pos += cnt;
        return GM_OK;
    }
};

struct VClaims {
    std::vector<Fr> point, evs;
};

// gamma_rlc == zip_with_gamma (sumcheck.rs:591-602, utils.rs:137-148)
Fr gamma_rlc(const Fr& gamma, const std::vector<Fr>& v) {
    if (v.empty()) return fr_zero();
    Fr r = v.back();
    for (size_t i = v.size() - 1; i-- > 0;) r = fr_add(fr_mul(r, gamma), v[i]);
    return r;
}

// eq_eval (utils.rs:150-156)
Fr eq_eval(const std::vector<Fr>& a, const std::vector<Fr>& b) {
    Fr r = fr_one();
    for (size_t i = 0; i < a.size(); i++) r = fr_mul(r, eq_bind_factor(a[i], b[i]));
    return r;
}

// decompress_coefficients (sumcheck.rs:14-25)
std::vector<Fr> decompress(const std::vector<Fr>& msg, const Fr& claim) {
    Fr sm = fr_dbl(msg[0]);
    for (size_t i = 1; i < msg.size(); i++) sm = fr_add(sm, msg[i]);
    std::vector<Fr> c;
    c.push_back(msg[0]);
    c.push_back(fr_sub(claim, sm));
    c.insert(c.end(), msg.begin() + 1, msg.end());
    return c;
}

// main_cycle_sumcheck_verifier (sumcheck.rs:63-77)
int32_t sumcheck_verify(Reader* tr, uint32_t degree, uint32_t num_vars, Fr* claim, std::vector<Fr>* point) {
    std::vector<Fr> r;
    for (uint32_t i = 0; i < num_vars; i++) {
        std::vector<Fr> msg(degree);
        TRY(tr->read_scalars(degree, msg.data()));
        const std::vector<Fr> poly = decompress(msg, *claim);
        Fr x;
        TRY(tr->challenge(&x));
        r.push_back(x);
        *claim = evaluate_univar(poly, x);
    }
    point->assign(r.rbegin(), r.rend());
    return GM_OK;
}

// DenseDeg2Sumcheck::verify (dense_eq.rs:223-237) = VecVecDeg2Sumcheck::verify (vecvec_eq.rs:452-467) = DenseEqSumcheck::verify
// (sumcheck.rs:874-889): the three differ only in the prover object
int32_t layer_verify(Reader* tr, const gm_fn& f, uint32_t num_vars, VClaims* c, const char* what) {
    const SegPlan sp = plan_of(f);
    Fr gamma;
    TRY(tr->challenge(&gamma));
    VERIFY((int)c->evs.size() == sp.n_outs, "%s: %zu claims for a function with %d outputs", what, c->evs.size(), sp.n_outs);
    Fr ev = gamma_rlc(gamma, c->evs);
    std::vector<Fr> out_pt;
    TRY(sumcheck_verify(tr, 3, num_vars, &ev, &out_pt));   // degrees = f.deg() + 1 with f.deg() = 2 for every layer function
    std::vector<Fr> poly_evs(sp.n_ins), fo(sp.n_outs);
    TRY(tr->read_scalars(sp.n_ins, poly_evs.data()));
    seg_plan_exec_host(sp, poly_evs.data(), fo.data());
    VERIFY(c->point.size() == out_pt.size(), "%s: claim point has %zu coordinates, the layer %u variables", what, c->point.size(), num_vars);
    VERIFY(fr_eq(fr_mul(gamma_rlc(gamma, fo), eq_eval(c->point, out_pt)), ev), "Final combinator check has failed (%s, %u variables)",
           what, num_vars);
    c->point = out_pt;
    c->evs = poly_evs;
    return GM_OK;
}

// SplitAt::verify = prove (splits.rs:121-147)
int32_t split_verify(Reader* tr, VClaims* c, bool hi, uint32_t idx, uint32_t bundle) {
    Fr r;
    TRY(tr->challenge(&r));
    std::vector<Fr> l, rr;
    for (size_t base = 0; base < c->evs.size(); base += bundle) {
        std::vector<Fr>& dst = ((base / bundle) % 2 == 0) ? l : rr;
        for (size_t i = base; i < base + bundle && i < c->evs.size(); i++) dst.push_back(c->evs[i]);
    }
    VERIFY(l.size() == rr.size(), "SplitAt: unbalanced bundles");
    std::vector<Fr> nw;
    for (size_t i = 0; i < l.size(); i++) nw.push_back(fr_add(l[i], fr_mul(r, fr_sub(rr[i], l[i]))));
    const size_t pos = hi ? idx : c->point.size() - idx;
    VERIFY(pos <= c->point.size(), "SplitAt: index past the point");
    c->point.insert(c->point.begin() + pos, r);
    c->evs = nw;
    return GM_OK;
}

// SimpleGKR::verify (gkr.rs:52-58)
int32_t gkr_verify(Reader* tr, const std::vector<Layer>& layers, VClaims* c, const char* what) {
    for (size_t k = layers.size(); k-- > 0;) {
        const Layer& L = layers[k];
        switch (L.kind) {
            case Layer::VECVEC:
            case Layer::DENSE: TRY(layer_verify(tr, L.f, L.num_vars, c, what)); break;
            case Layer::SPLIT: TRY(split_verify(tr, c, L.split_hi, L.split_idx, L.bundle)); break;
            case Layer::ZEROCHECK:  // zero_check.rs:24-33
                c->evs.push_back(fr_zero());
                c->evs.push_back(fr_zero());
                break;
        }
    }
    return GM_OK;
}

// LogupMainphaseProtocol::verify (logup_mainphase.rs:202-240)
int32_t logup_verify(Reader* tr, std::vector<uint32_t> logsizes, const Fr& claim, std::vector<VClaims>* out) {
    const gm_fn f = mkfn(GM_FN_LOGUP_LAYER, 1);
    Fr nd[2];
    TRY(tr->read_scalars(2, nd));
    VERIFY(!fr_is_zero(nd[1]), "logup: zero denominator (logup_mainphase.rs:206)");
    VERIFY(fr_eq(nd[0], fr_mul(nd[1], claim)), "logup: num != denom * claim (logup_mainphase.rs:207)");
    uint32_t curr = 0;
    VClaims running;
    running.evs = {nd[0], nd[1]};
    std::vector<VClaims> acc;
    for (;;) {
        VERIFY(!logsizes.empty(), "logup: ran out of inputs");
        const uint32_t incoming = logsizes.back();
        VClaims c4 = running;
        TRY(layer_verify(tr, f, curr, &c4, "logup layer"));
        if (incoming == curr) {
            if (logsizes.size() == 2) {
                acc.push_back(c4);
                break;
            }
            running.point = c4.point;
            running.evs = {c4.evs[0], c4.evs[1]};
            VClaims side;
            side.point = c4.point;
            side.evs = {c4.evs[2], c4.evs[3]};
            acc.push_back(side);
            logsizes.pop_back();
        } else {
            TRY(split_verify(tr, &c4, true, 0, 2));
            running = c4;
            curr++;
        }
    }
    out->assign(acc.rbegin(), acc.rend());
    return GM_OK;
}

struct PfFinal {
    Fr gamma;
    VClaims matrix, ac_c, ac_d;
};

// PushforwardProtocol::verify (pushforward.rs:849-968)
int32_t pushforward_verify(Reader* tr, uint32_t x_log, uint32_t y_log, uint32_t y_size, uint32_t d_log, VClaims claims, PfFinal* out) {
    VERIFY(claims.evs.size() == 3, "pushforward: expected 3 evaluations");
    VERIFY(claims.point.size() == (size_t)y_log + d_log + x_log, "pushforward: claim point length (pushforward.rs:859)");
    claims.evs[1] = fr_sub(claims.evs[1], fr_one());
    const std::vector<Fr> r_y(claims.point.begin(), claims.point.begin() + y_log);
    const uint32_t mlog = x_log + y_log;
    const uint64_t msize = (uint64_t)y_size << x_log;
    Fr ch[4], gamma;
    TRY(tr->challenge(ch, 4, 512));
    const Fr psi = ch[0], tau_c = ch[1], tau_d = ch[2], tau_s = ch[3];
    TRY(tr->challenge(&gamma));
    VERIFY(!fr_is_zero(tau_s), "tau_suppression_term is zero (inverse().unwrap(), pushforward.rs:895)");
    const Fr supp = fr_mul(fr_from_u64(2 * (((uint64_t)1 << mlog) - msize)), fr_inv(tau_s));
    std::vector<VClaims> mp;
    TRY(logup_verify(tr, {mlog - 1, mlog - 1, x_log, d_log}, supp, &mp));
    VERIFY(mp.size() == 3, "logup: three claim groups (pushforward.rs:904)");
    VClaims cd = mp[0];
    TRY(split_verify(tr, &cd, true, 0, 2));
    VERIFY(cd.evs.size() == 2, "pushforward: cd claims (pushforward.rs:923)");
    const Fr g1 = gamma, g2 = fr_mul(gamma, gamma);
    const Fr ev_folded = fr_add(fr_add(claims.evs[0], fr_mul(g1, claims.evs[1])), fr_mul(g2, claims.evs[2]));
    Fr claim = fr_add(fr_add(cd.evs[0], fr_mul(g1, cd.evs[1])), fr_mul(g2, ev_folded));
    std::vector<Fr> out_pt;
    TRY(sumcheck_verify(tr, 3, mlog, &claim, &out_pt));
    Fr fe[5];
    TRY(tr->read_scalars(5, fe));
    const Fr p_folded_ev = fe[0], c_pull_ev = fe[1], d_pull_ev = fe[2], c_ev = fe[3], d_ev = fe[4];
    const Fr adj_p = fr_sub(p_folded_ev, gamma);
    const Fr p_sel = fr_mul(adj_p, eq_trunc_evaluate(y_log, y_size, r_y.data(), out_pt.data()));
    const Fr sel_ev = eq_sum_host(out_pt.data(), y_log, y_size);   // SelectorPoly::evaluate (verifier_polys.rs:69-71)
    const Fr tmp = fr_mul(tau_s, fr_sub(fr_one(), sel_ev));
    const Fr c_adj = fr_add(fr_sub(fr_add(c_pull_ev, fr_mul(psi, c_ev)), fr_mul(tau_c, sel_ev)), tmp);
    const Fr d_adj = fr_add(fr_sub(fr_add(d_pull_ev, fr_mul(psi, d_ev)), fr_mul(tau_d, sel_ev)), tmp);
    const Fr lhs = fr_add(fr_mul(eq_eval(cd.point, out_pt), fr_add(fr_add(c_adj, d_adj), fr_mul(g1, fr_mul(c_adj, d_adj)))),
                          fr_mul(g2, fr_mul(fr_mul(c_pull_ev, d_pull_ev), p_sel)));
    VERIFY(fr_eq(lhs, claim), "pushforward: combined sumcheck final check (pushforward.rs:955-960)");
    out->gamma = gamma;
    out->matrix.point = out_pt;
    out->matrix.evs.assign(fe, fe + 5);
    out->ac_c = mp[1];
    out->ac_d = mp[2];
    return GM_OK;
}

// MultiOpenReduction::verify (multiopen_reduction.rs:95-117); FoldedProdAlgFn (:13-42): sum_i gamma^i a_i a_{i + n}
int32_t multiopen_verify(Reader* tr, uint32_t nvars, const std::vector<std::vector<Fr>>& pts, const std::vector<Fr>& evs_in,
                         std::vector<Fr>* out_pt, std::vector<Fr>* out_evs) {
    const size_t nargs = pts.size();
    Fr gamma;
    TRY(tr->challenge(&gamma));
    Fr claim = gamma_rlc(gamma, evs_in);
    TRY(sumcheck_verify(tr, 2, nvars, &claim, out_pt));
    out_evs->resize(nargs);
    TRY(tr->read_scalars(nargs, out_evs->data()));
    Fr acc = fr_zero(), gp = fr_one();
    for (size_t i = 0; i < nargs; i++) {
        VERIFY(pts[i].size() == nvars, "multiopen: claim point %zu has %zu coordinates", i, pts[i].size());
        acc = fr_add(acc, fr_mul(gp, fr_mul((*out_evs)[i], eq_eval(pts[i], *out_pt))));
        gp = fr_mul(gp, gamma);
    }
    VERIFY(fr_eq(claim, acc), "multiopen: final combinator check (multiopen_reduction.rs:110)");
    return GM_OK;
}

G1Jac mul_fr(const G1Jac& p, const Fr& k_mont) {
    const Fr k = fr_from_mont(k_mont);
    G1Jac acc = g1_inf();
    for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            acc = g1_dbl(acc);
            if ((k.l[i] >> b) & 1) acc = g1_add(acc, p);
        }
    return acc;
}
G1Jac jac(const G1Aff& a) { return g1_from_aff(a); }

// KzgVerifyingKey::verify_reduce_to_pair (kzg.rs:46-59): ([Q] at - g0 opening + [P], [Q])
void reduce_to_pair(const G1Jac& g0, const G1Jac& poly_comm, const G1Jac& quot_comm, const Fr& at, const Fr& opening, G1Jac* a, G1Jac* b) {
    *a = g1_add(g1_add(mul_fr(quot_comm, at), g1_neg(mul_fr(g0, opening))), poly_comm);
    *b = quot_comm;
}

// KnucklesOpeningProtocol::verify (opening.rs:100-143)
int32_t knuckles_verify(Reader* tr, const G1Jac& g0, const Fr& k, uint32_t num_vars, const G1Jac& commitment, const std::vector<Fr>& point,
                        const Fr& ev, G1Jac* pa, G1Jac* pb) {
    VERIFY(point.size() == num_vars, "knuckles: opening point has %zu coordinates, expected %u", point.size(), num_vars);
    G1Aff t_comm_a, p_lt_x_proof_a, t_kx_proof_a;
    TRY(tr->read_points(1, &t_comm_a));
    Fr x, lambda, fin, two[2], t_kx;
    TRY(tr->challenge(&x));
    const Fr kx = fr_mul(x, k);
    TRY(tr->read_scalars(2, two));
    const Fr t_x = two[0], p_x = two[1];
    TRY(tr->challenge(&lambda));
    const G1Jac t_comm = jac(t_comm_a);
    const G1Jac p_lt_comm = g1_add(mul_fr(t_comm, lambda), commitment);
    const Fr p_lt_open = fr_add(fr_mul(t_x, lambda), p_x);
    TRY(tr->read_points(1, &p_lt_x_proof_a));
    G1Jac a0, b0, a1, b1;
    reduce_to_pair(g0, p_lt_comm, jac(p_lt_x_proof_a), x, p_lt_open, &a0, &b0);
    TRY(tr->read_scalars(1, &t_kx));
    TRY(tr->read_points(1, &t_kx_proof_a));
    reduce_to_pair(g0, t_comm, jac(t_kx_proof_a), kx, t_kx, &a1, &b1);
    // k^(N - 1), N = 2^num_vars
    Fr k_pow = fr_one(), sq = k;
    for (uint32_t i = 0; i < num_vars; i++) {  // N - 1 = 2^num_vars - 1: all ones
        k_pow = fr_mul(k_pow, sq);
        sq = fr_mul(sq, sq);
    }
    Fr xpow = x, eq_ev = fr_one();
    for (uint32_t i = 0; i < num_vars; i++) {
        const Fr r = point[num_vars - i - 1];
        eq_ev = fr_mul(eq_ev, fr_add(r, fr_mul(fr_sub(fr_one(), r), xpow)));
        xpow = fr_mul(xpow, xpow);
    }
    const Fr lhs = fr_add(fr_mul(x, fr_sub(t_kx, fr_mul(k_pow, t_x))), fr_mul(xpow, ev));
    const Fr rhs = fr_mul(fr_mul(x, p_x), eq_ev);
    VERIFY(fr_eq(lhs, rhs), "knuckles: x (T(kx) - k^(N-1) T(x)) + x^N claim != x P(x) Eq(x) (opening.rs:138-140)");
    TRY(tr->challenge(&fin));
    *pa = g1_add(a0, mul_fr(a1, fin));
    *pb = g1_add(b0, mul_fr(b1, fin));
    return GM_OK;
}

// Pippenger::verify (pippenger.rs:296-406)
int32_t pippenger_verify(Reader* tr, uint32_t x_log, uint32_t d_log, uint32_t y_size, uint32_t y_log, uint32_t clm, const Fr* claim_point,
                         const Fr* claim_evs, const G1Aff& g0_aff, const Fr& k, uint64_t* h_pair) {
    GM_REQUIRE(x_log >= d_log && x_log >= 2 && d_log >= 2, "x_logsize >= d_logsize >= 2 required (pippenger.rs:93)");
    // size bounds first: the shapes below are allocated and shifted by these (nothing crosses the C ABI as an exception)
    GM_REQUIRE(x_log <= 30 && d_log <= 16 && y_log <= 16 && clm <= y_log, "shape out of range (x_logsize <= 30, d_logsize <= 16, y_logsize <= 16, clm <= y_logsize)");
    GM_REQUIRE(y_size >= 1 && y_size <= (1u << y_log) && (uint64_t)y_size * d_log <= 256, "bad y_size / y_logsize (y_size * d_logsize <= 256, pushforward.rs:358)");
    // GM_OK is NOT acceptance by itself: Pippenger::verify ends with vkey.kzg_vk.verify_pair(ps_pair) (pippenger.rs:403-405); here the
    // pair is handed back for gm_kzg_verify_pair, so the caller must take it
    GM_REQUIRE(h_pair, "h_pair is required: the proof is accepted only if gm_kzg_verify_pair(h_pair, h0, h1) also returns GM_OK");
    const uint32_t cm = 1u << clm, n_mat = (y_size + cm - 1) / cm;
    std::vector<G1Aff> c(n_mat), d(n_mat), c_pull(n_mat), d_pull(n_mat);
    G1Aff p_0, p_1, ac_c, ac_d;
    TRY(tr->read_points(n_mat, c.data()));
    TRY(tr->read_points(n_mat, d.data()));
    TRY(tr->read_points(1, &p_0));
    TRY(tr->read_points(1, &p_1));
    TRY(tr->read_points(1, &ac_c));
    TRY(tr->read_points(1, &ac_d));
    // self.ending.verify (PippengerBucketed::verify, pippenger_ending.rs:151-163)
    VClaims cl;
    cl.point.assign(claim_point, claim_point + y_log);
    cl.evs.assign(claim_evs, claim_evs + 3 * (d_log + 1));
    TRY(gkr_verify(tr, triangle_layers(y_log + d_log - 2, y_log), &cl, "bucket reduction (triangle_add)"));
    TRY(split_verify(tr, &cl, true, y_log, 3));
    TRY(split_verify(tr, &cl, true, y_log, 3));
    TRY(gkr_verify(tr, bintree_layers(y_log + d_log + x_log, x_log, x_log, true), &cl, "bucket sums (bintree_add)"));
    {   // GlueSplit::verify = prove (splits.rs:185-201)
        Fr r;
        TRY(tr->challenge(&r));
        VERIFY(cl.evs.size() == 6, "GlueSplit expects 6 evaluations, got %zu", cl.evs.size());
        const std::vector<Fr> e = cl.evs;
        cl.evs = {fr_add(e[0], fr_mul(r, fr_sub(e[2], e[0]))), fr_add(e[1], fr_mul(r, fr_sub(e[3], e[1]))),
                  fr_add(e[4], fr_mul(r, fr_sub(e[5], e[4])))};
        cl.point.push_back(r);
    }
    TRY(tr->read_points(n_mat, c_pull.data()));
    TRY(tr->read_points(n_mat, d_pull.data()));
    PfFinal pf;
    TRY(pushforward_verify(tr, x_log, y_log, y_size, d_log, cl, &pf));
    const Fr gamma = pf.gamma;
    const std::vector<Fr>& mpt = pf.matrix.point;
    const Fr p_folded_ev = pf.matrix.evs[0], c_pull_ev = pf.matrix.evs[1], d_pull_ev = pf.matrix.evs[2], c_ev = pf.matrix.evs[3],
             d_ev = pf.matrix.evs[4];
    const uint32_t nv = x_log + clm;
    std::vector<std::vector<Fr>> pts(4, std::vector<Fr>(nv, fr_zero()));
    VERIFY(pf.ac_c.point.size() == x_log && pf.ac_d.point.size() == d_log && mpt.size() == (size_t)x_log + y_log,
           "pushforward claims have unexpected point sizes");
    for (uint32_t i = 0; i < x_log; i++) pts[0][clm + i] = mpt[y_log + i];
    for (uint32_t i = 0; i < x_log; i++) pts[1][clm + i] = pf.ac_c.point[i];
    for (uint32_t i = 0; i < d_log; i++) pts[2][nv - d_log + i] = pf.ac_d.point[i];
    for (uint32_t i = 0; i < nv; i++) pts[3][i] = mpt[y_log - clm + i];
    // multirow_evs = EqPoly(y_log - clm, matrix_pt[..y_log - clm]).evals()
    std::vector<Fr> multirow((size_t)1 << (y_log - clm), fr_zero());
    multirow[0] = fr_one();
    for (uint32_t i = 0; i < y_log - clm; i++)
        for (uint64_t j = (1ull << i); j-- > 0;) {
            const Fr w = multirow[j], m = fr_mul(mpt[i], w);
            multirow[2 * j] = fr_sub(w, m);
            multirow[2 * j + 1] = m;
        }
    // sum_m multirow_evs[m] * commitment[m] for the four commitment vectors (pippenger.rs:344-347): 4 n_mat scalar
    // multiplications, the bulk of the verifier's time, spread over a few host threads
    const std::vector<G1Aff>* vecs[4] = {&c, &d, &c_pull, &d_pull};
    const uint32_t n_comb = (uint32_t)(n_mat < multirow.size() ? n_mat : multirow.size());
    std::vector<G1Jac> prod(4 * (size_t)n_comb);
    host_parallel_for(4 * n_comb, [&](uint32_t t) { prod[t] = mul_fr(jac((*vecs[t / n_comb])[t % n_comb]), multirow[t % n_comb]); }, 2);
    auto comb = [&](int which) {
        G1Jac acc = g1_inf();
        for (uint32_t i = 0; i < n_comb; i++) acc = g1_add(acc, prod[(size_t)which * n_comb + i]);
        return acc;
    };
    const G1Jac c_comb = comb(0), d_comb = comb(1), cp_comb = comb(2), dp_comb = comb(3);
    Fr u;
    TRY(tr->challenge(&u, 1, 512));
    const Fr u2 = fr_mul(u, u), u3 = fr_mul(u2, u);
    const G1Jac combined_comm = g1_add(g1_add(c_comb, mul_fr(d_comb, u)), g1_add(mul_fr(cp_comb, u2), mul_fr(dp_comb, u3)));
    const Fr combined_ev = fr_add(fr_add(c_ev, fr_mul(d_ev, u)), fr_add(fr_mul(c_pull_ev, u2), fr_mul(d_pull_ev, u3)));
    VERIFY(pf.ac_c.evs.size() >= 1 && pf.ac_d.evs.size() >= 1, "access-count claims are empty");
    const std::vector<Fr> mo_in = {fr_sub(p_folded_ev, fr_mul(gamma, gamma)), pf.ac_c.evs[0], pf.ac_d.evs[0], combined_ev};
    std::vector<Fr> mo_pt, mo_evs;
    TRY(multiopen_verify(tr, nv, pts, mo_in, &mo_pt, &mo_evs));
    Fr q;
    TRY(tr->challenge(&q));
    const Fr q2 = fr_mul(q, q), q3 = fr_mul(q2, q);
    const G1Jac parts[4] = {g1_add(jac(p_0), mul_fr(jac(p_1), gamma)), jac(ac_c), jac(ac_d), combined_comm};
    const G1Jac folded_comm = g1_add(g1_add(parts[0], mul_fr(parts[1], q)), g1_add(mul_fr(parts[2], q2), mul_fr(parts[3], q3)));
    G1Jac a, b;
    TRY(knuckles_verify(tr, jac(g0_aff), k, nv, folded_comm, mo_pt, gamma_rlc(q, mo_evs), &a, &b));
    if (h_pair) {
        const G1Aff aa = g1_to_aff(a), ba = g1_to_aff(b);
        memcpy(h_pair, &aa, sizeof(G1Aff));
        memcpy(h_pair + 12, &ba, sizeof(G1Aff));
    }
    return GM_OK;
}

}  // namespace

extern "C" int32_t gm_pippenger_verify_tr(uint32_t x_logsize, uint32_t d_logsize, uint32_t y_size, uint32_t y_logsize,
                                          uint32_t commitment_log_multiplicity, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                          const uint64_t* h_g0_aff, const uint64_t* h_k, const gm_transcript_reader* tr, uint64_t* h_pair) {
    try {
        GM_REQUIRE(h_claim_point && h_claim_evs && h_g0_aff && h_k && tr && tr->read_scalars && tr->read_points && tr->challenge,
                   "null argument");
        Reader rd;
        rd.cb = tr;
        G1Aff g0;
        Fr k;
        memcpy(&g0, h_g0_aff, sizeof(G1Aff));
        memcpy(&k, h_k, sizeof(Fr));
        return pippenger_verify(&rd, x_logsize, d_logsize, y_size, y_logsize, commitment_log_multiplicity,
                                reinterpret_cast<const Fr*>(h_claim_point), reinterpret_cast<const Fr*>(h_claim_evs), g0, k, h_pair);
    } catch (const std::exception& e) {
        return set_err(GM_ERR_INVALID, "gm_pippenger_verify_tr: %s", e.what());
    }
}

extern "C" int32_t gm_pippenger_verify(uint32_t x_logsize, uint32_t d_logsize, uint32_t y_size, uint32_t y_logsize,
                                       uint32_t commitment_log_multiplicity, const uint64_t* h_claim_point, const uint64_t* h_claim_evs,
                                       const uint64_t* h_g0_aff, const uint64_t* h_k, const uint64_t* h_scalars, uint64_t n_scalars,
                                       const uint64_t* h_points_aff, uint64_t n_points, const uint64_t* h_tape, uint64_t n_tape,
                                       uint64_t* h_pair, uint64_t* tape_used) {
    try {
        GM_REQUIRE(h_claim_point && h_claim_evs && h_g0_aff && h_k && (h_scalars || !n_scalars) && (h_points_aff || !n_points) &&
                       (h_tape || !n_tape),
                   "null argument");
        Reader rd;
        rd.scalars = reinterpret_cast<const Fr*>(h_scalars);
        rd.points = reinterpret_cast<const G1Aff*>(h_points_aff);
        rd.tape = h_tape;
        rd.n_scalars = n_scalars; rd.n_points = n_points; rd.n_tape = n_tape;
        G1Aff g0;
        Fr k;
        memcpy(&g0, h_g0_aff, sizeof(G1Aff));
        memcpy(&k, h_k, sizeof(Fr));
        TRY(pippenger_verify(&rd, x_logsize, d_logsize, y_size, y_logsize, commitment_log_multiplicity,
                             reinterpret_cast<const Fr*>(h_claim_point), reinterpret_cast<const Fr*>(h_claim_evs), g0, k, h_pair));
        if (rd.si != n_scalars || rd.pi != n_points)
            return set_err(GM_ERR_VERIFY, "proof has unread messages (%llu of %llu scalars, %llu of %llu points read)",
                           (unsigned long long)rd.si, (unsigned long long)n_scalars, (unsigned long long)rd.pi, (unsigned long long)n_points);
        if (tape_used) *tape_used = rd.pos;
        return GM_OK;
    } catch (const std::exception& e) {
        return set_err(GM_ERR_INVALID, "gm_pippenger_verify: %s", e.what());
    }
}

// KzgVerifyingKey::verify_pair (kzg.rs:61-67): e(A, h0) == e(B, h1), as e(A, h0) e(-B, h1) == 1 with one final exponentiation.
// h_pair = A, B affine (2 x 12 u64); h_h0, h_h1 = G2 affine: x.c0, x.c1, y.c0, y.c1, Montgomery 6 x u64 each (24 u64)
extern "C" int32_t gm_kzg_verify_pair(const uint64_t* h_pair, const uint64_t* h_h0, const uint64_t* h_h1) {
    GM_REQUIRE(h_pair && h_h0 && h_h1, "null argument");
    G1Aff ps[2];
    G2Aff qs[2];
    memcpy(&ps[0], h_pair, sizeof(G1Aff));
    memcpy(&ps[1], h_pair + 12, sizeof(G1Aff));
    memcpy(&qs[0], h_h0, sizeof(G2Aff));
    memcpy(&qs[1], h_h1, sizeof(G2Aff));
    GM_REQUIRE(g1_aff_on_curve(ps[0]) && g1_aff_on_curve(ps[1]), "pairing pair is not on the curve");
    GM_REQUIRE(g2_aff_on_curve(qs[0]) && g2_aff_on_curve(qs[1]), "verifying key is not on the twist");
    ps[1] = g1_aff_is_inf(ps[1]) ? ps[1] : g1_aff_neg(ps[1]);
    if (!pairing_product_is_one(ps, qs, 2)) return set_err(GM_ERR_VERIFY, "pairing check failed: e(A, h0) != e(B, h1) (kzg.rs:66)");
    return GM_OK;
}

// The G2 half of KzgProvingKey::mock_setup's verifying key (kzg.rs: h0 = [1]_2, h1 = [tau]_2) for a known tau: tests and benches
// that generate their own SRS (gm_g1_mock_srs) need it for gm_kzg_verify_pair.  h_tau: Montgomery Fr.
extern "C" int32_t gm_kzg_mock_vk(const uint64_t* h_tau, uint64_t* h_h0, uint64_t* h_h1) {
    GM_REQUIRE(h_tau && h_h0 && h_h1, "null argument");
    Fr t;
    memcpy(&t, h_tau, sizeof(Fr));
    t = fr_from_mont(t);
    const G2Aff g = g2_generator();
    const G2Aff h1 = g2_mul(g, t.l, 8);
    memcpy(h_h0, &g, sizeof(G2Aff));
    memcpy(h_h1, &h1, sizeof(G2Aff));
    return GM_OK;
}

// e(P, Q) as 12 Fq coordinates (Montgomery, tower order a.a.a, a.a.b, a.b.a, ... b.c.b): test hook for the pairing itself
extern "C" int32_t gm_pairing(const uint64_t* h_p_aff, const uint64_t* h_q_aff, uint64_t* h_gt) {
    GM_REQUIRE(h_p_aff && h_q_aff && h_gt, "null argument");
    G1Aff p;
    G2Aff q;
    memcpy(&p, h_p_aff, sizeof(G1Aff));
    memcpy(&q, h_q_aff, sizeof(G2Aff));
    GM_REQUIRE(g1_aff_on_curve(p) && g2_aff_on_curve(q), "point not on its curve");
    const Fq12 e = pairing(p, q);
    memcpy(h_gt, &e, sizeof(Fq12));
    return GM_OK;
}

// ---------------------------------------------------------------------------------------------- gen-1
// The verifier side of gkr_msm_prove's GKR (gkr_msm_simple.rs:248-338): BintreeVerifier::round (protocol/bintree.rs:313-395)
// over SumcheckPolyMapVerifier::round (protocol/sumcheck.rs:595-657) and SplitVerifier::round (protocol/split.rs:99-115), on the
// transcript stream gm_gkr_msm_prove emits: the output polynomials, then per mapping layer the round polynomials (all
// coefficients, as `append_scalars(b"poly", ..)` sees them) and the final evaluations.  The reference's proof object stores the
// round polynomials compressed and rebuilds the linear coefficient from the running sum (sumcheck.rs:649); on the full
// coefficients that is the check p(0) + p(1) == sum.
namespace {

int32_t gkr_msm_verify(Reader* tr, uint32_t lp, uint32_t lb, Fr* out_point, uint32_t* n_point, Fr* out_evs, uint64_t* rounds) {
    GM_REQUIRE(lp >= 1 && lb >= 1 && lp <= 30 && lb <= 16, "bad log_num_points / log_num_scalar_bits (<= 30 / <= 16)");
    struct L1 { bool is_map; int prim; uint32_t nv; };
    std::vector<L1> layers;   // gkr_msm_simple.rs:248-269 unrolled (bintree.rs:81-123)
    {
        uint32_t nv = lp + lb;
        auto map = [&](int id) { layers.push_back(L1{true, id, nv}); };
        auto split = [&]() { layers.push_back(L1{false, 0, nv}); nv--; };
        map(GM_FN_PT_BIT_CHOICE);
        split();
        map(GM_FN_AFF_L1); map(GM_FN_AFF_L2); map(GM_FN_AFF_L3);
        for (uint32_t i = 0; i + 1 < lp; i++) { split(); map(GM_FN_PROJ_L1); map(GM_FN_PROJ_L2); map(GM_FN_PROJ_L3); }
    }
    const uint64_t nout = 1ull << lb;
    std::vector<std::vector<Fr>> out(3, std::vector<Fr>(nout));
    for (int c = 0; c < 3; c++) TRY(tr->read_scalars(nout, out[c].data()));
    std::vector<Fr> point(lb), evs(3);
    for (uint32_t i = 0; i < lb; i++) TRY(tr->challenge(&point[i], 1, 512));
    for (int c = 0; c < 3; c++) {  // FragmentedPoly::evaluate (fragmented.rs:748-761)
        std::vector<Fr> v = out[c];
        for (int k = (int)lb - 1; k >= 0; k--) {
            for (size_t i = 0; i < v.size() / 2; i++) v[i] = fr_add(v[2 * i], fr_mul(point[k], fr_sub(v[2 * i + 1], v[2 * i])));
            v.resize(v.size() / 2);
        }
        evs[c] = v[0];
    }
    uint64_t nrounds = 0;
    for (size_t li = layers.size(); li-- > 0;) {
        const L1& L = layers[li];
        Fr c0;
        TRY(tr->challenge(&c0, 1, 512));
        if (!L.is_map) {  // SplitVerifier::round
            const size_t h = evs.size() / 2;
            std::vector<Fr> nw(h);
            for (size_t i = 0; i < h; i++) nw[i] = fr_add(evs[i], fr_mul(c0, fr_sub(evs[h + i], evs[i])));
            evs = nw;
            point.push_back(c0);
            continue;
        }
        const SegPlan sp = plan_of(mkfn(L.prim, 1));
        VERIFY((int)evs.size() == sp.n_outs && point.size() == L.nv, "Verifier failure. Claim ill-formed at layer %zu (sumcheck.rs:546-558)", li);
        Fr sum = fr_zero(), gp = fr_one();   // make_folded_claim (sumcheck.rs:659-673)
        for (size_t i = 0; i < evs.size(); i++) { sum = fr_add(sum, fr_mul(evs[i], gp)); gp = fr_mul(gp, c0); }
        std::vector<Fr> rs;
        const uint32_t ncoef = (uint32_t)sp.deg + 2;   // degree f.degree + 1
        for (uint32_t rd = 0; rd < L.nv; rd++) {
            std::vector<Fr> poly(ncoef);
            TRY(tr->read_scalars(ncoef, poly.data()));
            Fr at1 = fr_zero();
            for (const Fr& c : poly) at1 = fr_add(at1, c);
            VERIFY(fr_eq(fr_add(poly[0], at1), sum), "Verifier failure: round polynomial does not sum to the claim (layer %zu, round %u)", li, rd);
            Fr r;
            TRY(tr->challenge(&r, 1, 512));
            rs.insert(rs.begin(), r);   // fix_var_bot
            sum = evaluate_univar(poly, r);
            nrounds++;
        }
        std::vector<Fr> fe(sp.n_ins), fo(sp.n_outs);
        TRY(tr->read_scalars(sp.n_ins, fe.data()));
        seg_plan_exec_host(sp, fe.data(), fo.data());
        Fr folded = fr_zero();
        gp = fr_one();
        for (int o = 0; o < sp.n_outs; o++) { folded = fr_add(folded, fr_mul(fo[o], gp)); gp = fr_mul(gp, c0); }
        VERIFY(fr_eq(fr_mul(folded, eq_eval(point, rs)), sum), "Verifier failure: final check incorrect (layer %zu, sumcheck.rs:638)", li);
        evs = fe;
        point = rs;
    }
    if (n_point) *n_point = (uint32_t)point.size();
    if (out_point) memcpy(out_point, point.data(), point.size() * sizeof(Fr));
    if (out_evs) memcpy(out_evs, evs.data(), evs.size() * sizeof(Fr));
    if (rounds) *rounds = nrounds;
    return GM_OK;
}

}  // namespace

extern "C" int32_t gm_gkr_msm_verify(uint32_t log_num_points, uint32_t log_num_scalar_bits, const uint64_t* h_msgs, uint64_t n_msgs,
                                     const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_final_point, uint32_t* n_final_point,
                                     uint64_t* h_final_evs, uint64_t* tape_used, uint64_t* rounds) {
    try {
        GM_REQUIRE((h_msgs || !n_msgs) && (h_tape || !n_tape), "null argument");
        Reader rd;
        rd.scalars = reinterpret_cast<const Fr*>(h_msgs);
        rd.tape = h_tape;
        rd.n_scalars = n_msgs; rd.n_tape = n_tape;
        TRY(gkr_msm_verify(&rd, log_num_points, log_num_scalar_bits, reinterpret_cast<Fr*>(h_final_point), n_final_point,
                           reinterpret_cast<Fr*>(h_final_evs), rounds));
        if (rd.si != n_msgs) return set_err(GM_ERR_VERIFY, "transcript has unread messages (%llu of %llu read)", (unsigned long long)rd.si,
                                            (unsigned long long)n_msgs);
        if (tape_used) *tape_used = rd.pos;
        return GM_OK;
    } catch (const std::exception& e) {
        return set_err(GM_ERR_INVALID, "gm_gkr_msm_verify: %s", e.what());
    }
}

extern "C" int32_t gm_gkr_msm_verify_tr(uint32_t log_num_points, uint32_t log_num_scalar_bits, const gm_transcript_reader* tr,
                                        uint64_t* h_final_point, uint32_t* n_final_point, uint64_t* h_final_evs, uint64_t* rounds) {
    try {
        GM_REQUIRE(tr && tr->read_scalars && tr->challenge, "null argument");
        Reader rd;
        rd.cb = tr;
        return gkr_msm_verify(&rd, log_num_points, log_num_scalar_bits, reinterpret_cast<Fr*>(h_final_point), n_final_point,
                              reinterpret_cast<Fr*>(h_final_evs), rounds);
    } catch (const std::exception& e) {
        return set_err(GM_ERR_INVALID, "gm_gkr_msm_verify_tr: %s", e.what());
    }
}
