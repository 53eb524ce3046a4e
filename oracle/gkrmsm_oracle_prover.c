/* TEST INFRASTRUCTURE ONLY -- see gkrmsm_oracle.h.
 *
 * CPU restatement of the gen-2 "image part" prover: VecVec polynomials, the deg-2 sumcheck objects and the
 * bintree / triangle GKR drivers.  Row-at-a-time loops over explicit row arrays, as the reference stores them
 * (Vec<Vec<F>>); OpenMP over rows / chunks where the reference uses rayon (and also where it is serial: the
 * "fair" CPU baseline of BASELINE.md section 3).
 */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>

/* CPU-baseline variants.  0 = "fair": every data-parallel loop is threaded.  1 = "reference-faithful":
 * the loops the reference runs serially stay serial -- the round sums of DenseDeg2SumcheckObjectSO (dense_eq.rs:121-139) and of
 * VecVecDeg2SumcheckObjectSO (vecvec_eq.rs:320-361), and vecvec_map_split (vecvec.rs:579-594; dense algfn_map_split,
 * dense.rs:130-135, is serial in both) -- everything the reference hands to rayon stays threaded.  Results are identical. */
static int g_faithful = 0;
void or_set_reference_faithful(int on) { g_faithful = on ? 1 : 0; }

#endif

#include "gkrmsm_oracle.h"

static const or_fr ONE = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
static const or_fr ZERO = {{0, 0, 0, 0}};

static or_fr f_add(or_fr a, or_fr b) { or_fr r; or_fr_add(&r, &a, &b); return r; }
static or_fr f_sub(or_fr a, or_fr b) { or_fr r; or_fr_sub(&r, &a, &b); return r; }
static or_fr f_mul(or_fr a, or_fr b) { or_fr r; or_fr_mul(&r, &a, &b); return r; }
static or_fr f_inv(or_fr a) { or_fr r; or_fr_inv(&r, &a); return r; }
static or_fr f_from_u64(uint64_t v) {
    or_fr a = {{v, 0, 0, 0}}, r;
    or_fr_batch(5, &a, NULL, &r, 1);
    return r;
}

/* ------------------------------------------------------------------------------------------------ containers */
typedef struct {
    int k;
    uint32_t nrows, row_log, col_log;
    uint32_t* len;  /* stored row lengths (even), shared by the k polynomials */
    or_fr*** rows;  /* rows[c][r] -> len[r] elements */
    or_fr* row_pad;
    or_fr* col_pad;
} vvset;

typedef struct {
    int k;
    uint64_t len;
    or_fr** col;
} dset;

typedef struct {
    int kind; /* 0 EMPTY, 1 VECVEC, 2 DENSE */
    vvset* vv;
    dset* d;
} advice;

static vvset* vv_new(int k, uint32_t nrows, uint32_t row_log, uint32_t col_log) {
    vvset* v = (vvset*)calloc(1, sizeof(vvset));
    v->k = k; v->nrows = nrows; v->row_log = row_log; v->col_log = col_log;
    v->len = (uint32_t*)calloc(nrows ? nrows : 1, sizeof(uint32_t));
    v->rows = (or_fr***)calloc(k ? k : 1, sizeof(or_fr**));
    for (int c = 0; c < k; c++) v->rows[c] = (or_fr**)calloc(nrows ? nrows : 1, sizeof(or_fr*));
    v->row_pad = (or_fr*)calloc(k ? k : 1, sizeof(or_fr));
    v->col_pad = (or_fr*)calloc(k ? k : 1, sizeof(or_fr));
    return v;
}
static void vv_free(vvset* v) {
    if (!v) return;
    for (int c = 0; c < v->k; c++) {
        for (uint32_t r = 0; r < v->nrows; r++) free(v->rows[c][r]);
        free(v->rows[c]);
    }
    free(v->rows); free(v->len); free(v->row_pad); free(v->col_pad); free(v);
}
static dset* d_new(int k, uint64_t len) {
    dset* d = (dset*)calloc(1, sizeof(dset));
    d->k = k; d->len = len;
    d->col = (or_fr**)calloc(k ? k : 1, sizeof(or_fr*));
    for (int c = 0; c < k; c++) d->col[c] = (or_fr*)malloc((len ? len : 1) * sizeof(or_fr));
    return d;
}
static void d_free(dset* d) {
    if (!d) return;
    for (int c = 0; c < d->k; c++) free(d->col[c]);
    free(d->col); free(d);
}

/* ------------------------------------------------------------------------------------------------ maps */
/* vecvec_map, vecvec.rs:480-540 (exec reads args[0..n_ins), extra polys are ignored) */
static vvset* vv_map(const or_fn* f, const vvset* in) {
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
    vvset* o = vv_new(no, in->nrows, in->row_log, in->col_log);
    or_fr a[64], b[64];
    for (int i = 0; i < ni; i++) a[i] = in->row_pad[i];
    or_fn_exec(f, a, b);
    for (int i = 0; i < no; i++) o->row_pad[i] = b[i];
    for (int i = 0; i < ni; i++) a[i] = in->col_pad[i];
    or_fn_exec(f, a, b);
    for (int i = 0; i < no; i++) o->col_pad[i] = b[i];
#pragma omp parallel for schedule(dynamic, 8)
    for (uint32_t r = 0; r < in->nrows; r++) {
        or_fr x[64], y[64];
        o->len[r] = in->len[r];
        for (int c = 0; c < no; c++) o->rows[c][r] = (or_fr*)malloc((in->len[r] ? in->len[r] : 1) * sizeof(or_fr));
        for (uint32_t i = 0; i < in->len[r]; i++) {
            for (int c = 0; c < ni; c++) x[c] = in->rows[c][r][i];
            or_fn_exec(f, x, y);
            for (int c = 0; c < no; c++) o->rows[c][r][i] = y[c];
        }
    }
    return o;
}

static int bundle_col(int oc, int half, int bundle) { return 2 * (oc / bundle) * bundle + half * bundle + oc % bundle; }

/* vecvec_map_split with LO(0), vecvec.rs:542-606 */
static vvset* vv_map_split(const or_fn* f, const vvset* in, int bundle) {
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
    vvset* o = vv_new(2 * no, in->nrows, in->row_log - 1, in->col_log);
    or_fr a[64], rp[64], cp[64];
    for (int i = 0; i < ni; i++) a[i] = in->row_pad[i];
    or_fn_exec(f, a, rp);
    for (int i = 0; i < ni; i++) a[i] = in->col_pad[i];
    or_fn_exec(f, a, cp);
    for (int oc = 0; oc < no; oc++)
        for (int h = 0; h < 2; h++) {
            o->row_pad[bundle_col(oc, h, bundle)] = rp[oc];
            o->col_pad[bundle_col(oc, h, bundle)] = cp[oc];
        }
#pragma omp parallel for schedule(dynamic, 8) if (!g_faithful)
    for (uint32_t r = 0; r < in->nrows; r++) {
        or_fr x[64], y[64];
        uint32_t half = in->len[r] / 2, plen = half + (half & 1);
        o->len[r] = plen;
        for (int c = 0; c < 2 * no; c++) o->rows[c][r] = (or_fr*)malloc((plen ? plen : 1) * sizeof(or_fr));
        for (uint32_t i = 0; i < in->len[r]; i++) {
            for (int c = 0; c < ni; c++) x[c] = in->rows[c][r][i];
            or_fn_exec(f, x, y);
            for (int oc = 0; oc < no; oc++) o->rows[bundle_col(oc, (int)(i & 1), bundle)][r][i >> 1] = y[oc];
        }
        if (half & 1)
            for (int oc = 0; oc < no; oc++)
                for (int h = 0; h < 2; h++) o->rows[bundle_col(oc, h, bundle)][r][half] = rp[oc];
    }
    return o;
}

/* vecvec_map_split_to_dense, vecvec.rs:608-654 */
static dset* vv_map_split_to_dense(const or_fn* f, const vvset* in, int bundle) {
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
    uint64_t n = 1ULL << in->col_log;
    dset* o = d_new(2 * no, n);
    or_fr a[64], rp[64], cp[64], y[64];
    for (int i = 0; i < ni; i++) a[i] = in->row_pad[i];
    or_fn_exec(f, a, rp);
    for (int i = 0; i < ni; i++) a[i] = in->col_pad[i];
    or_fn_exec(f, a, cp);
    for (uint64_t r = 0; r < n; r++) {
        if (r < in->nrows && in->len[r]) {
            for (uint32_t i = 0; i < 2; i++) {
                for (int c = 0; c < ni; c++) a[c] = in->rows[c][r][i];
                or_fn_exec(f, a, y);
                for (int oc = 0; oc < no; oc++) o->col[bundle_col(oc, (int)i, bundle)][r] = y[oc];
            }
        } else {
            for (int oc = 0; oc < no; oc++)
                for (int h = 0; h < 2; h++) o->col[bundle_col(oc, h, bundle)][r] = (r < in->nrows) ? rp[oc] : cp[oc];
        }
    }
    return o;
}

static dset* dn_map(const or_fn* f, const dset* in) {
    dset* o = d_new(or_fn_n_outs(f), in->len);
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < in->len; i++) {
        or_fr x[64], y[64];
        for (int c = 0; c < ni; c++) x[c] = in->col[c][i];
        or_fn_exec(f, x, y);
        for (int c = 0; c < no; c++) o->col[c][i] = y[c];
    }
    return o;
}
static dset* dn_map_split(const or_fn* f, const dset* in, uint32_t lo_bit, int bundle) {
    dset* o = d_new(2 * or_fn_n_outs(f), in->len / 2);
    or_dense_map_split(f, (const or_fr* const*)in->col, in->len, lo_bit, (uint32_t)bundle, o->col);
    return o;
}

/* ------------------------------------------------------------------------------------------------ univariates */
static void unipoly_from_evals(const or_fr* ev, int n, or_fr* coeffs) {
    for (int k = 0; k < n; k++) coeffs[k] = ZERO;
    for (int i = 0; i < n; i++) {
        or_fr num[8], den = ONE;
        int deg = 0;
        num[0] = ONE;
        for (int j = 0; j < n; j++) {
            if (j == i) continue;
            or_fr nx[8], fj = f_from_u64((uint64_t)j);
            for (int k = 0; k <= deg + 1; k++) nx[k] = ZERO;
            for (int k = 0; k <= deg; k++) {
                nx[k + 1] = f_add(nx[k + 1], num[k]);
                nx[k] = f_sub(nx[k], f_mul(fj, num[k]));
            }
            deg++;
            memcpy(num, nx, sizeof(or_fr) * (deg + 1));
            or_fr d = (i > j) ? f_from_u64((uint64_t)(i - j)) : f_sub(ZERO, f_from_u64((uint64_t)(j - i)));
            den = f_mul(den, d);
        }
        or_fr sc = f_mul(ev[i], f_inv(den));
        for (int k = 0; k < n; k++) coeffs[k] = f_add(coeffs[k], f_mul(num[k], sc));
    }
}
static or_fr evaluate_univar(const or_fr* c, int n, or_fr x) { /* sumcheck.rs:33-44 */
    or_fr r = ZERO;
    for (int i = n - 1; i >= 0; i--) r = f_add(f_mul(r, x), c[i]);
    return r;
}
static void from12(or_fr p1, or_fr p2, or_fr eq1, or_fr claim, or_fr* coeffs) { /* vecvec_eq.rs:197-216 */
    or_fr eq0 = f_sub(ONE, eq1), eq2 = f_sub(f_add(eq1, eq1), eq0), eq3 = f_sub(f_add(eq2, eq2), eq1);
    or_fr prod1 = f_mul(p1, eq1), prod0 = f_sub(claim, prod1), p0 = f_mul(prod0, f_inv(eq0));
    or_fr p3 = f_add(f_sub(f_sub(f_add(f_add(p2, p2), p2), f_add(p1, p1)), p1), p0);
    or_fr ev[4] = {prod0, prod1, f_mul(p2, eq2), f_mul(p3, eq3)};
    unipoly_from_evals(ev, 4, coeffs);
}
static or_fr eq_bind_factor(or_fr q, or_fr t) {
    or_fr qt = f_mul(q, t);
    return f_add(f_sub(f_sub(ONE, q), t), f_add(qt, qt));
}

/* gamma-combined f at the "1" and "2" points of one pair (value-at-2 = 2 p1 - p0) */
static void eval12(const or_fn* f, int ni, int no, const or_fr* p0, const or_fr* p1, const or_fr* gp, or_fr* A1, or_fr* A2) {
    or_fr v2[64], o1[64], o2[64];
    for (int c = 0; c < ni; c++) v2[c] = f_sub(f_add(p1[c], p1[c]), p0[c]);
    or_fn_exec(f, p1, o1);
    or_fn_exec(f, v2, o2);
    or_fr a1 = o1[0], a2 = o2[0];
    for (int o = 1; o < no; o++) { a1 = f_add(a1, f_mul(gp[o], o1[o])); a2 = f_add(a2, f_mul(gp[o], o2[o])); }
    *A1 = a1; *A2 = a2;
}

/* ------------------------------------------------------------------------------------------------ transcript */
typedef struct {
    const uint64_t* tape;
    uint64_t n, pos;
    or_fr* msgs;
    uint64_t cap, nmsgs, rounds;
    int err;
} tape_t;
static or_fr tp_challenge(tape_t* t) {
    if (t->pos >= t->n) { t->err = 1; return ZERO; }
    or_fr c, r;
    memcpy(&c, t->tape + 4 * t->pos, 32);
    t->pos++;
    or_fr_batch(5, &c, NULL, &r, 1);
    return r;
}
static void tp_write(tape_t* t, const or_fr* v, int n) {
    for (int i = 0; i < n; i++) {
        if (t->nmsgs < t->cap) t->msgs[t->nmsgs] = v[i]; else t->err = 2;
        t->nmsgs++;
    }
}
typedef struct { or_fr point[80]; int npoint; or_fr evs[80]; int nevs; } claims_t;

static void make_gamma_pows(or_fr gamma, int n, or_fr* gp) { /* utils.rs:126-135 */
    gp[0] = ONE;
    if (n > 1) gp[1] = gamma;
    for (int i = 2; i < n; i++) gp[i] = f_mul(gp[i - 1], gamma);
}

/* DenseSumcheckObjectSO rounds for F = EqWrapper(GammaWrapper(f)) over cols (last col = eq), sumcheck.rs:237-347;
 * runs `nv` rounds of GenericSumcheckProtocol::prove (sumcheck.rs:101-123); cols are consumed (folded in place). */
static void dense_eqgamma_rounds(tape_t* tr, const or_fn* f, const or_fr* gp, or_fr** cols, int ncols, uint32_t nv,
                                 or_fr claim, or_fr* r_out) {
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
    for (uint32_t rd = 0; rd < nv; rd++) {
        uint64_t half = 1ULL << (nv - rd - 1);
        or_fr acc[3] = {ZERO, ZERO, ZERO};
        for (uint64_t i = 0; i < half; i++) {
            or_fr a[64], d[64], o[64];
            for (int c = 0; c < ncols; c++) { a[c] = cols[c][2 * i + 1]; d[c] = f_sub(cols[c][2 * i + 1], cols[c][2 * i]); }
            for (int s = 0; s < 3; s++) {
                if (s) for (int c = 0; c < ncols; c++) a[c] = f_add(a[c], d[c]);
                or_fn_exec(f, a, o);
                or_fr g = o[0];
                for (int k = 1; k < no; k++) g = f_add(g, f_mul(gp[k], o[k]));
                acc[s] = f_add(acc[s], f_mul(g, a[ni]));
            }
        }
        or_fr ev[4] = {f_sub(claim, acc[0]), acc[0], acc[1], acc[2]}, co[4];
        unipoly_from_evals(ev, 4, co);
        or_fr msg[3] = {co[0], co[2], co[3]};
        tp_write(tr, msg, 3);
        or_fr x = tp_challenge(tr);
        r_out[rd] = x;
        for (int c = 0; c < ncols; c++) or_dense_bind(cols[c], 2 * half, &x, cols[c]);
        claim = evaluate_univar(co, 4, x);
        tr->rounds++;
    }
}

/* DenseDeg2Sumcheck::prove, dense_eq.rs:198-229 (object :61-173).  Input columns are copied (the advice is consumed). */
static void dense_deg2_prove(tape_t* tr, const or_fn* f, uint32_t nv, claims_t* cl, const dset* adv) {
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
    or_fr gamma = tp_challenge(tr), gp[64];
    make_gamma_pows(gamma, no, gp);
    or_fr claim = cl->evs[0];
    for (int i = 1; i < no; i++) claim = f_add(claim, f_mul(gp[i], cl->evs[i]));
    uint64_t n = 1ULL << nv;
    or_fr** cols = (or_fr**)malloc(sizeof(or_fr*) * ni);
    for (int c = 0; c < ni; c++) { cols[c] = (or_fr*)malloc(n * sizeof(or_fr)); memcpy(cols[c], adv->col[c], n * sizeof(or_fr)); }
    /* eq_poly_sequence(point[0..nv-1]): levels packed, level i at offset 2^i - 1 */
    or_fr* eqs = (or_fr*)malloc(n * sizeof(or_fr));
    eqs[0] = ONE;
    for (uint32_t i = 1; i < nv; i++) {
        const or_fr* prev = eqs + ((1ULL << (i - 1)) - 1);
        or_fr* nxt = eqs + ((1ULL << i) - 1);
        for (uint64_t j = 0; j < (1ULL << (i - 1)); j++) {
            or_fr m = f_mul(cl->point[i - 1], prev[j]);
            nxt[2 * j] = f_sub(prev[j], m);
            nxt[2 * j + 1] = m;
        }
    }
    or_fr mult = ONE, r[64];
    int np = (int)nv;
    for (uint32_t rd = 0; rd < nv; rd++) {
        uint64_t half = 1ULL << (nv - rd - 1);
        const or_fr* eq = eqs + (half - 1);
        or_fr s1 = ZERO, s2 = ZERO;
#pragma omp parallel if (!g_faithful)
        {
            or_fr l1 = ZERO, l2 = ZERO;
#pragma omp for schedule(static) nowait
            for (uint64_t i = 0; i < half; i++) {
                or_fr p0[64], p1[64], A1, A2;
                for (int c = 0; c < ni; c++) { p0[c] = cols[c][2 * i]; p1[c] = cols[c][2 * i + 1]; }
                eval12(f, ni, no, p0, p1, gp, &A1, &A2);
                l1 = f_add(l1, f_mul(A1, eq[i]));
                l2 = f_add(l2, f_mul(A2, eq[i]));
            }
#pragma omp critical
            { s1 = f_add(s1, l1); s2 = f_add(s2, l2); }
        }
        or_fr co[4];
        from12(f_mul(s1, mult), f_mul(s2, mult), cl->point[np - 1], claim, co);
        or_fr msg[3] = {co[0], co[2], co[3]};
        tp_write(tr, msg, 3);
        or_fr x = tp_challenge(tr);
        r[rd] = x;
        mult = f_mul(mult, eq_bind_factor(cl->point[np - 1], x));
        np--;
#pragma omp parallel for schedule(static)
        for (int c = 0; c < ni; c++) or_dense_bind(cols[c], 2 * half, &x, cols[c]);
        claim = evaluate_univar(co, 4, x);
        tr->rounds++;
    }
    for (uint32_t i = 0; i < nv; i++) cl->point[i] = r[nv - 1 - i];
    cl->npoint = (int)nv;
    for (int c = 0; c < ni; c++) cl->evs[c] = cols[c][0];
    cl->nevs = ni;
    tp_write(tr, cl->evs, ni);
    for (int c = 0; c < ni; c++) free(cols[c]);
    free(cols); free(eqs);
}

static uint32_t log2_lasso(uint32_t n) { /* liblasso Math::log_2 */
    uint32_t bl = 0;
    while ((1u << bl) < n) bl++;
    return bl;
}

/* VecVecDeg2Sumcheck::prove, vecvec_eq.rs:424-456 (objects :72-398, EQPolyData vecvec.rs:68-147) */
static void vecvec_deg2_prove(tape_t* tr, const or_fn* f, uint32_t nv, claims_t* cl, const vvset* adv) {
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
    uint32_t nrows = adv->nrows, col_log = adv->col_log, row_log = adv->row_log;
    or_fr gamma = tp_challenge(tr), gp[64];
    make_gamma_pows(gamma, no, gp);
    or_fr claim = cl->evs[0];
    for (int i = 1; i < no; i++) claim = f_add(claim, f_mul(gp[i], cl->evs[i]));
    /* working copy of the rows */
    uint32_t* len = (uint32_t*)malloc((nrows ? nrows : 1) * sizeof(uint32_t));
    or_fr*** rows = (or_fr***)malloc(sizeof(or_fr**) * ni);
    uint32_t maxlen = 0;
    for (uint32_t r = 0; r < nrows; r++) { len[r] = adv->len[r]; if (len[r] > maxlen) maxlen = len[r]; }
    for (int c = 0; c < ni; c++) {
        rows[c] = (or_fr**)malloc(sizeof(or_fr*) * (nrows ? nrows : 1));
        for (uint32_t r = 0; r < nrows; r++) {
            rows[c][r] = (or_fr*)malloc((len[r] ? len[r] : 1) * sizeof(or_fr));
            memcpy(rows[c][r], adv->rows[c][r], len[r] * sizeof(or_fr));
        }
    }
    /* EQPolyData::new */
    uint32_t max_seg_log = log2_lasso(maxlen);
    uint32_t padded = (nv - max_seg_log) - col_log, nseq = (nv - 1) - col_log;
    uint64_t ncoef = 1ULL << col_log;
    or_fr* coef = (or_fr*)malloc(ncoef * sizeof(or_fr));
    or_eq_table(&ONE, cl->point, col_log, coef);
    or_fr* tail = (or_fr*)malloc((ncoef + 1) * sizeof(or_fr));
    tail[ncoef] = ZERO;
    for (int64_t i = (int64_t)ncoef - 1; i >= 0; i--) tail[i] = f_add(tail[i + 1], coef[i]);
    /* padded_eq_poly_sequence (utils.rs:189-220) and prefix sums */
    or_fr** seq = (or_fr**)malloc(sizeof(or_fr*) * (nseq + 1));
    or_fr** pre = (or_fr**)malloc(sizeof(or_fr*) * (nseq + 1));
    uint32_t* slen = (uint32_t*)malloc(sizeof(uint32_t) * (nseq + 1));
    const or_fr* rpt = cl->point + col_log;
    for (uint32_t i = 0; i <= nseq; i++) {
        slen[i] = (i <= padded) ? 1u : (1u << (i - padded));
        seq[i] = (or_fr*)malloc(slen[i] * sizeof(or_fr));
        if (i == 0) seq[0][0] = ONE;
        else if (i <= padded) seq[i][0] = f_mul(seq[i - 1][0], f_sub(ONE, rpt[i - 1]));
        else
            for (uint32_t j = 0; j < slen[i - 1]; j++) {
                or_fr m = f_mul(rpt[i - 1], seq[i - 1][j]);
                seq[i][2 * j] = f_sub(seq[i - 1][j], m);
                seq[i][2 * j + 1] = m;
            }
        pre[i] = (or_fr*)malloc((slen[i] + 1) * sizeof(or_fr));
        pre[i][0] = ZERO;
        for (uint32_t j = 0; j < slen[i]; j++) pre[i][j + 1] = f_add(pre[i][j], seq[i][j]);
    }
    or_fr padin[64], pr[64], pc[64];
    for (int c = 0; c < ni; c++) padin[c] = adv->row_pad[c];
    or_fn_exec(f, padin, pr);
    for (int c = 0; c < ni; c++) padin[c] = adv->col_pad[c];
    or_fn_exec(f, padin, pc);
    or_fr padsum = pr[0], colsum = pc[0];
    for (int o = 1; o < no; o++) { padsum = f_add(padsum, f_mul(gp[o], pr[o])); colsum = f_add(colsum, f_mul(gp[o], pc[o])); }

    or_fr mult = ONE, rch[80];
    int bind_idx = (int)nv - 1;
    uint32_t bound = 0, rd = 0;
    or_fr co[4];
    for (; rd < nv && (uint32_t)bind_idx >= col_log; rd++) {
        const or_fr* eq = seq[nseq - bound];
        const or_fr* px = pre[nseq - bound];
        or_fr s1 = ZERO, s2 = ZERO;
#pragma omp parallel if (!g_faithful)
        {
            or_fr l1 = ZERO, l2 = ZERO;
#pragma omp for schedule(dynamic, 8) nowait
            for (uint32_t r = 0; r < nrows; r++) {
                or_fr a1 = ZERO, a2 = ZERO;
                uint32_t seg = len[r] / 2;
                for (uint32_t i = 0; i < seg; i++) {
                    or_fr p0[64], p1[64], A1, A2;
                    for (int c = 0; c < ni; c++) { p0[c] = rows[c][r][2 * i]; p1[c] = rows[c][r][2 * i + 1]; }
                    eval12(f, ni, no, p0, p1, gp, &A1, &A2);
                    a1 = f_add(a1, f_mul(A1, eq[i]));
                    a2 = f_add(a2, f_mul(A2, eq[i]));
                }
                or_fr tr_ = f_mul(padsum, f_sub(ONE, px[seg])); /* get_trailing_sum */
                l1 = f_add(l1, f_mul(f_add(a1, tr_), coef[r]));
                l2 = f_add(l2, f_mul(f_add(a2, tr_), coef[r]));
            }
#pragma omp critical
            { s1 = f_add(s1, l1); s2 = f_add(s2, l2); }
        }
        if (nrows < ncoef) { or_fr e = f_mul(colsum, tail[nrows]); s1 = f_add(s1, e); s2 = f_add(s2, e); }
        from12(f_mul(s1, mult), f_mul(s2, mult), cl->point[bind_idx], claim, co);
        or_fr msg[3] = {co[0], co[2], co[3]};
        tp_write(tr, msg, 3);
        or_fr x = tp_challenge(tr);
        rch[rd] = x;
        tr->rounds++;
        if ((uint32_t)bind_idx > col_log) {
            /* sparse bind (bind_21, vecvec.rs:420-441) */
#pragma omp parallel for schedule(dynamic, 8)
            for (uint32_t r = 0; r < nrows; r++) {
                uint32_t half = len[r] / 2, plen = half + (half & 1);
                for (int c = 0; c < ni; c++) {
                    or_fr* row = rows[c][r];
                    for (uint32_t i = 0; i < half; i++) row[i] = f_add(row[2 * i], f_mul(x, f_sub(row[2 * i + 1], row[2 * i])));
                    if (half & 1) row[half] = adv->row_pad[c];
                }
                len[r] = plen;
            }
            mult = f_mul(mult, eq_bind_factor(cl->point[bind_idx], x));
            bind_idx--;
            bound++;
            claim = evaluate_univar(co, 4, x);
        } else {
            /* bind_into_dense (vecvec_eq.rs:157-190), then the dense rounds */
            uint64_t nd = 1ULL << col_log;
            or_fr** cols = (or_fr**)malloc(sizeof(or_fr*) * (ni + 1));
            for (int c = 0; c < ni; c++) {
                cols[c] = (or_fr*)malloc(nd * sizeof(or_fr));
                for (uint64_t r = 0; r < nd; r++) {
                    if (r >= nrows) cols[c][r] = adv->col_pad[c];
                    else if (len[r] == 0) cols[c][r] = adv->row_pad[c];
                    else cols[c][r] = f_add(rows[c][r][0], f_mul(x, f_sub(rows[c][r][1], rows[c][r][0])));
                }
            }
            or_fr m2 = f_mul(mult, eq_bind_factor(cl->point[bind_idx], x));
            cols[ni] = (or_fr*)malloc(nd * sizeof(or_fr));
            or_eq_table(&m2, cl->point, col_log, cols[ni]);
            claim = evaluate_univar(co, 4, x);
            dense_eqgamma_rounds(tr, f, gp, cols, ni + 1, col_log, claim, rch + rd + 1);
            for (uint32_t i = 0; i < nv; i++) cl->point[i] = rch[nv - 1 - i];
            cl->npoint = (int)nv;
            for (int c = 0; c < ni; c++) cl->evs[c] = cols[c][0];
            cl->nevs = ni;  /* poly_evs.pop(): the eq column is dropped */
            tp_write(tr, cl->evs, ni);
            for (int c = 0; c <= ni; c++) free(cols[c]);
            free(cols);
            rd = nv;
            break;
        }
    }
    for (int c = 0; c < ni; c++) { for (uint32_t r = 0; r < nrows; r++) free(rows[c][r]); free(rows[c]); }
    free(rows); free(len); free(coef); free(tail);
    for (uint32_t i = 0; i <= nseq; i++) { free(seq[i]); free(pre[i]); }
    free(seq); free(pre); free(slen);
    (void)row_log;
}

/* SplitAt::prove, splits.rs:121-143 */
static void split_at_prove(tape_t* tr, claims_t* c, int hi, uint32_t idx, int bundle) {
    or_fr r = tp_challenge(tr), l[80], rr[80], nw[80];
    int nl = 0, nr = 0;
    for (int base = 0; base < c->nevs; base += bundle)
        for (int i = base; i < base + bundle && i < c->nevs; i++) {
            if ((base / bundle) % 2 == 0) l[nl++] = c->evs[i]; else rr[nr++] = c->evs[i];
        }
    int n = nl < nr ? nl : nr;
    for (int i = 0; i < n; i++) nw[i] = f_add(l[i], f_mul(r, f_sub(rr[i], l[i])));
    int pos = hi ? (int)idx : c->npoint - (int)idx;
    memmove(&c->point[pos + 1], &c->point[pos], sizeof(or_fr) * (size_t)(c->npoint - pos));
    c->point[pos] = r;
    c->npoint++;
    memcpy(c->evs, nw, sizeof(or_fr) * (size_t)n);
    c->nevs = n;
}

/* ------------------------------------------------------------------------------------------------ witness */
struct or_pip_witness {
    uint32_t x_log, y_log, d_log;
    int n_bt, n_tri;
    advice* bt;   /* bintree advices  (bintree_add.rs:137-184) */
    advice* tri;  /* triangle advices (triangle_add.rs:101-158) */
    dset* bucket_sums;
    dset* output;
};

static or_fn mkfn(int p0, int c0, int p1, int c1) {
    or_fn f;
    memset(&f, 0, sizeof(f));
    f.nseg = p1 ? 2 : 1;
    f.prim[0] = p0; f.count[0] = c0; f.prim[1] = p1; f.count[1] = c1;
    return f;
}

static advice adv_map(const or_fn* f, const advice* a) {
    advice o = {0, NULL, NULL};
    if (a->kind == 1) { o.kind = 1; o.vv = vv_map(f, a->vv); }
    else { o.kind = 2; o.d = dn_map(f, a->d); }
    return o;
}
static advice adv_map_split(const or_fn* f, const advice* a, uint32_t layer_idx, uint32_t row_logsize) {
    advice o = {0, NULL, NULL};
    if (a->kind == 1) {
        if (layer_idx + 2 == row_logsize) { o.kind = 2; o.d = vv_map_split_to_dense(f, a->vv, 3); }
        else { o.kind = 1; o.vv = vv_map_split(f, a->vv, 3); }
    } else { o.kind = 2; o.d = dn_map_split(f, a->d, 0, 3); }
    return o;
}

or_pip_witness* or_pip_witness_create(const or_fr* pts, const uint64_t* scalars, uint32_t x_log, uint32_t d_log,
                                      uint32_t y_size, uint32_t y_log, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    if (x_log < 2 || x_log < d_log || d_log < 2 || (uint64_t)y_size * d_log > 256 || (1u << y_log) < y_size) return NULL;
    const uint64_t N = 1ULL << x_log;
    const uint32_t nd = 1u << d_log, nrows = y_size << d_log, mask = nd - 1;
    /* bucket image (pushforward.rs:342-349, 380-381, 411-426, 477-487): polys x, y, z; pads (0,1,0); odd rows padded */
    vvset* img = vv_new(3, nrows, x_log, y_log + d_log);
    img->row_pad[0] = ZERO; img->row_pad[1] = ONE; img->row_pad[2] = ZERO;
    memcpy(img->col_pad, img->row_pad, 3 * sizeof(or_fr));
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t y = 0; y < y_size; y++) {
        uint32_t* cnt = (uint32_t*)calloc(nd, sizeof(uint32_t));
        uint16_t* dg = (uint16_t*)malloc(N * sizeof(uint16_t));
        uint32_t bit = y * d_log;
        for (uint64_t x = 0; x < N; x++) {
            const uint64_t* s = scalars + 4 * x;
            uint32_t li = bit >> 6, sh = bit & 63;
            uint64_t v = s[li] >> sh;
            if (sh && li < 3) v |= s[li + 1] << (64 - sh);
            dg[x] = (uint16_t)(v & mask);
            cnt[dg[x]]++;
        }
        for (uint32_t k = 0; k < nd; k++) {
            uint32_t r = (y << d_log) + k, pl = cnt[k] + (cnt[k] & 1);
            img->len[r] = pl;
            for (int c = 0; c < 3; c++) img->rows[c][r] = (or_fr*)malloc((pl ? pl : 1) * sizeof(or_fr));
            if (cnt[k] & 1) for (int c = 0; c < 3; c++) img->rows[c][r][cnt[k]] = img->row_pad[c];
            cnt[k] = 0;
        }
        for (uint64_t x = 0; x < N; x++) {
            uint32_t r = (y << d_log) + dg[x], i = cnt[dg[x]]++;
            img->rows[0][r][i] = pts[2 * x];
            img->rows[1][r][i] = pts[2 * x + 1];
            img->rows[2][r][i] = ONE;
        }
        free(cnt); free(dg);
    }
    /* GlueSplit::witness (splits.rs:172-176) */
    vvset xy = *img, z = *img;
    xy.k = 2;
    z.k = 1; z.rows = img->rows + 2; z.row_pad = img->row_pad + 2; z.col_pad = img->col_pad + 2;
    or_fn id2 = mkfn(8, 2, 0, 0), id1 = mkfn(8, 1, 0, 0);
    vvset* a = vv_map_split(&id2, &xy, 2);
    vvset* b = vv_map_split(&id1, &z, 1);
    vvset* g = vv_new(0, nrows, a->row_log, a->col_log);
    free(g->rows); free(g->row_pad); free(g->col_pad); free(g->len);
    g->k = 6;
    g->len = a->len;
    g->rows = (or_fr***)malloc(6 * sizeof(or_fr**));
    g->row_pad = (or_fr*)malloc(6 * sizeof(or_fr));
    g->col_pad = (or_fr*)malloc(6 * sizeof(or_fr));
    for (int c = 0; c < 4; c++) { g->rows[c] = a->rows[c]; g->row_pad[c] = a->row_pad[c]; g->col_pad[c] = a->col_pad[c]; }
    for (int c = 0; c < 2; c++) { g->rows[4 + c] = b->rows[c]; g->row_pad[4 + c] = b->row_pad[c]; g->col_pad[4 + c] = b->col_pad[c]; }
    free(a->rows); free(a->row_pad); free(a->col_pad); free(a);
    free(b->rows); free(b->row_pad); free(b->col_pad); free(b->len); free(b);
    vv_free(img);

    or_pip_witness* w = (or_pip_witness*)calloc(1, sizeof(or_pip_witness));
    w->x_log = x_log; w->y_log = y_log; w->d_log = d_log;
    /* bintree_add::builder::witness::build(advice, horizontal, horizontal, true) */
    const uint32_t na = x_log;
    w->bt = (advice*)calloc(4 * na + 2, sizeof(advice));
    advice cur = {1, g, NULL};
    for (uint32_t add = 0; add < na; add++) {
        int last = add + 1 == na;
        for (int step = 0; step < 3; step++) {
            advice nxt = {0, NULL, NULL};
            int have = 1;
            or_fn f;
            if (step == 0) { f = mkfn(add == 0 ? 1 : 4, 1, 0, 0); nxt = adv_map(&f, &cur); }
            else if (step == 1) { f = mkfn(add == 0 ? 2 : 5, 1, 0, 0); nxt = adv_map(&f, &cur); }
            else if (last) have = 0;
            else { f = mkfn(add == 0 ? 3 : 6, 1, 0, 0); nxt = adv_map_split(&f, &cur, add, na); }
            w->bt[w->n_bt++] = cur;
            if (add == 0 && step == 0) w->n_bt++; /* EMPTY for the ZeroCheck layer */
            if (have) cur = nxt;
        }
        if (!last) w->n_bt++; /* EMPTY for the split */
    }
    /* last_step, splits, triangle witness, dense output (pippenger_ending.rs:46-61, pippenger.rs:531-534) */
    or_fn l3 = mkfn(6, 1, 0, 0), id3 = mkfn(8, 3, 0, 0), id6 = mkfn(8, 6, 0, 0);
    w->bucket_sums = dn_map(&l3, w->bt[w->n_bt - 1].d);
    const uint32_t nv = y_log + d_log;
    dset* s1 = dn_map_split(&id3, w->bucket_sums, nv - 1 - y_log, 3);
    dset* s2 = dn_map_split(&id6, s1, (nv - 1) - 1 - y_log, 3);
    d_free(s1);
    const uint32_t tnv = nv - 2, layers = tnv - y_log;
    w->tri = (advice*)calloc(4 * (layers + 1) + 2, sizeof(advice));
    advice tc = {2, NULL, s2};
    for (uint32_t l = 0; l <= layers; l++) {
        for (int step = 0; step < 3; step++) {
            advice nxt = {2, NULL, NULL};
            int have = 1;
            or_fn f;
            if (step == 0) { f = mkfn(7, 1, 4, (int)l); nxt.d = dn_map(&f, tc.d); }
            else if (step == 1) { f = mkfn(5, (int)l + 3, 0, 0); nxt.d = dn_map(&f, tc.d); }
            else if (l == layers) have = 0;
            else { f = mkfn(6, (int)l + 3, 0, 0); nxt.d = dn_map_split(&f, tc.d, (tnv - l) - 1 - y_log, 3); }
            w->tri[w->n_tri++] = tc;
            if (have) tc = nxt;
        }
        if (l < layers) w->n_tri++;
    }
    or_fn lf = mkfn(6, (int)(d_log - 2) + 3, 0, 0);
    w->output = dn_map(&lf, w->tri[w->n_tri - 1].d);
    return w;
}

void or_pip_witness_destroy(or_pip_witness* w) {
    if (!w) return;
    for (int i = 0; i < w->n_bt; i++) { if (w->bt[i].kind == 1) vv_free(w->bt[i].vv); else if (w->bt[i].kind == 2) d_free(w->bt[i].d); }
    for (int i = 0; i < w->n_tri; i++) if (w->tri[i].kind == 2) d_free(w->tri[i].d);
    d_free(w->bucket_sums); d_free(w->output);
    free(w->bt); free(w->tri); free(w);
}

void or_pip_witness_output(const or_pip_witness* w, or_fr* out) {
    for (int c = 0; c < w->output->k; c++) memcpy(out + (size_t)c * w->output->len, w->output->col[c], w->output->len * sizeof(or_fr));
}

int or_pip_prove_image_part(or_pip_witness* w, const or_fr* claim_point, const or_fr* claim_evs, const uint64_t* tape,
                            uint64_t n_tape, or_fr* msgs, uint64_t msgs_cap, uint64_t* n_msgs, or_fr* final_point,
                            uint32_t* n_final_point, or_fr* final_evs, uint64_t* tape_used, uint64_t* rounds, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    tape_t tr = {tape, n_tape, 0, msgs, msgs_cap, 0, 0, 0};
    claims_t c;
    const uint32_t ml = w->y_log, bk = w->d_log, hz = w->x_log;
    memcpy(c.point, claim_point, ml * sizeof(or_fr));
    c.npoint = (int)ml;
    c.nevs = 3 * ((int)bk + 1);
    memcpy(c.evs, claim_evs, (size_t)c.nevs * sizeof(or_fr));
    /* TriangleAdd::prove: layers reversed (gkr.rs:45-50, triangle_add.rs:173-232) */
    const uint32_t tnv = ml + bk - 2, layers = tnv - ml;
    int ai = w->n_tri - 1;
    for (int64_t l = layers; l >= 0; l--) {
        if ((uint32_t)l < layers) { split_at_prove(&tr, &c, 1, ml, 3); ai--; }
        or_fn f3 = mkfn(6, (int)l + 3, 0, 0), f2 = mkfn(5, (int)l + 3, 0, 0), f1 = mkfn(7, 1, 4, (int)l);
        dense_deg2_prove(&tr, &f3, tnv - (uint32_t)l, &c, w->tri[ai--].d);
        dense_deg2_prove(&tr, &f2, tnv - (uint32_t)l, &c, w->tri[ai--].d);
        dense_deg2_prove(&tr, &f1, tnv - (uint32_t)l, &c, w->tri[ai--].d);
    }
    split_at_prove(&tr, &c, 1, ml, 3);
    split_at_prove(&tr, &c, 1, ml, 3);
    /* VecVecBintreeAdd::prove (bintree_add.rs:247-375) */
    const uint32_t bnv = ml + bk + hz;
    ai = w->n_bt - 1;
    for (int64_t i = (int64_t)hz - 1; i >= 0; i--) {
        if ((uint32_t)i != hz - 1) { split_at_prove(&tr, &c, 0, 0, 3); ai--; }
        const int vv = (i == 0) || ((uint32_t)i + 1 < hz);
        for (int step = 2; step >= 0; step--) {
            or_fn f = (i == 0) ? mkfn(step == 0 ? 1 : step == 1 ? 2 : 3, 1, 0, 0) : mkfn(step == 0 ? 4 : step == 1 ? 5 : 6, 1, 0, 0);
            if (i == 0 && step == 0) {
                /* ZeroCheck (zero_check.rs:24-28) then the stacked bit-check layer */
                c.evs[c.nevs++] = ZERO;
                c.evs[c.nevs++] = ZERO;
                ai--;
                f = mkfn(1, 1, 9, 2);
            }
            if (vv) vecvec_deg2_prove(&tr, &f, bnv - (uint32_t)i - 1, &c, w->bt[ai--].vv);
            else dense_deg2_prove(&tr, &f, bnv - (uint32_t)i - 1, &c, w->bt[ai--].d);
        }
    }
    /* GlueSplit::prove (splits.rs:185-197) */
    or_fr r = tp_challenge(&tr);
    or_fr nw[3] = {f_add(c.evs[0], f_mul(r, f_sub(c.evs[2], c.evs[0]))), f_add(c.evs[1], f_mul(r, f_sub(c.evs[3], c.evs[1]))),
                   f_add(c.evs[4], f_mul(r, f_sub(c.evs[5], c.evs[4])))};
    c.point[c.npoint++] = r;
    if (n_msgs) *n_msgs = tr.nmsgs;
    if (final_point) memcpy(final_point, c.point, (size_t)c.npoint * sizeof(or_fr));
    if (n_final_point) *n_final_point = (uint32_t)c.npoint;
    if (final_evs) memcpy(final_evs, nw, 3 * sizeof(or_fr));
    if (tape_used) *tape_used = tr.pos;
    if (rounds) *rounds = tr.rounds;
    return tr.err;
}

/* ================================================================================================
 * gen-1 prover gkr_msm_prove (Fr part), full shapes.  Restates
 *   src/gkr_msm_simple.rs:82-338, src/protocol/bintree.rs:81-123, 168-288,
 *   src/protocol/sumcheck.rs:67-257, 659-701, src/protocol/split.rs:37-82, src/polynomial/fragmented.rs:676-761,
 *   src/copoly.rs:457-633 (EqPoly on a full shape = the eq table), src/utils.rs:104-113, 167-173.
 */
typedef struct { int is_map; or_fn f; int n_split; uint32_t nv; } g1_layer;

int or_gkr_msm_prove(const or_fr* points_xy, const uint8_t* bits, uint32_t lp, uint32_t lb, const uint64_t* tape,
                     uint64_t n_tape, or_fr* msgs, uint64_t msgs_cap, uint64_t* n_msgs, or_fr* output /* 3 * 2^lb */,
                     or_fr* final_point, uint32_t* n_final_point, or_fr* final_evs, uint64_t* tape_used, uint64_t* rounds,
                     int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    if (lp < 1 || lb < 1 || lp + lb > 30) return 1;
    const uint32_t nv0 = lp + lb;
    const uint64_t n0 = 1ULL << nv0;
    g1_layer layers[4 * 32 + 8];
    int nl = 0;
    {
        uint32_t nv = nv0;
        g1_layer L;
        memset(&L, 0, sizeof(L));
#define G1_MAP(id) do { L.is_map = 1; L.f = mkfn(id, 1, 0, 0); L.n_split = 0; L.nv = nv; layers[nl++] = L; } while (0)
#define G1_SPLIT(n) do { L.is_map = 0; L.f = mkfn(8, n, 0, 0); L.n_split = n; L.nv = nv; layers[nl++] = L; nv--; } while (0)
        G1_MAP(10); G1_SPLIT(2); G1_MAP(1); G1_MAP(2); G1_MAP(3);
        for (uint32_t i = 0; i + 1 < lp; i++) { G1_SPLIT(3); G1_MAP(4); G1_MAP(5); G1_MAP(6); }
    }
    /* base layer: index = point * 2^lb + bit (gkr_msm_simple.rs:120, 150-186) */
    dset** trace = (dset**)calloc((size_t)nl + 1, sizeof(dset*));
    dset* cur = d_new(3, n0);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n0; i++) {
        cur->col[0][i] = bits[i] ? ONE : ZERO;
        cur->col[1][i] = points_xy[2 * (i >> lb)];
        cur->col[2][i] = points_xy[2 * (i >> lb) + 1];
    }
    for (int li = 0; li < nl; li++) {
        trace[li] = cur;
        cur = layers[li].is_map ? dn_map(&layers[li].f, cur) : dn_map_split(&layers[li].f, cur, 0, layers[li].n_split);
    }
    tape_t tr = {tape, n_tape, 0, msgs, msgs_cap, 0, 0, 0};
    const uint64_t nout = 1ULL << lb;
    for (int c = 0; c < 3; c++) {
        tp_write(&tr, cur->col[c], (int)nout);
        if (output) memcpy(output + (size_t)c * nout, cur->col[c], nout * sizeof(or_fr));
    }
    /* gen-1 challenges are full field elements: the tape holds canonical values < p */
    claims_t cl;
    for (uint32_t i = 0; i < lb; i++) cl.point[i] = tp_challenge(&tr);
    cl.npoint = (int)lb;
    for (int c = 0; c < 3; c++) { /* FragmentedPoly::evaluate */
        or_fr* v = (or_fr*)malloc(nout * sizeof(or_fr));
        memcpy(v, cur->col[c], nout * sizeof(or_fr));
        uint64_t len = nout;
        for (int k = (int)lb - 1; k >= 0; k--) { or_dense_bind(v, len, &cl.point[k], v); len /= 2; }
        cl.evs[c] = v[0];
        free(v);
    }
    cl.nevs = 3;
    d_free(cur);
    for (int li = nl - 1; li >= 0; li--) {
        const g1_layer* L = &layers[li];
        or_fr c0 = tp_challenge(&tr);
        if (!L->is_map) { /* SplitProver::round */
            int h = cl.nevs / 2;
            for (int i = 0; i < h; i++) cl.evs[i] = f_add(cl.evs[i], f_mul(c0, f_sub(cl.evs[h + i], cl.evs[i])));
            cl.nevs = h;
            cl.point[cl.npoint++] = c0; /* fix_var_top */
            continue;
        }
        const or_fn* f = &L->f;
        int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
        uint32_t nv = L->nv;
        or_fr gp[64];
        gp[0] = ONE; gp[1] = c0;
        for (int i = 2; i < no; i++) gp[i] = f_mul(gp[i - 1], c0);
        uint64_t n = 1ULL << nv;
        or_fr** cols = (or_fr**)malloc(sizeof(or_fr*) * (size_t)(ni + 1));
        for (int c = 0; c < ni; c++) { cols[c] = (or_fr*)malloc(n * sizeof(or_fr)); memcpy(cols[c], trace[li]->col[c], n * sizeof(or_fr)); }
        cols[ni] = (or_fr*)malloc(n * sizeof(or_fr));
        or_eq_table(&ONE, cl.point, nv, cols[ni]);
        or_fr rs[64];
        for (uint32_t rd = 0; rd < nv; rd++) {
            uint64_t half = 1ULL << (nv - rd - 1);
            or_fr S[4] = {ZERO, ZERO, ZERO, ZERO};
#pragma omp parallel
            {
                or_fr loc[4] = {ZERO, ZERO, ZERO, ZERO};
#pragma omp for schedule(static) nowait
                for (uint64_t i = 0; i < half; i++) {
                    or_fr a[64], d[64], o[64];
                    for (int c = 0; c <= ni; c++) { a[c] = cols[c][2 * i]; d[c] = f_sub(cols[c][2 * i + 1], cols[c][2 * i]); }
                    for (int k = 0; k < 4; k++) { /* evaluations at 0, 1, 2, 3 (sumcheck.rs:99-151) */
                        if (k) for (int c = 0; c <= ni; c++) a[c] = f_add(a[c], d[c]);
                        or_fn_exec(f, a, o);
                        or_fr g = o[0];
                        for (int q = 1; q < no; q++) g = f_add(g, f_mul(gp[q], o[q]));
                        loc[k] = f_add(loc[k], f_mul(g, a[ni]));
                    }
                }
#pragma omp critical
                for (int k = 0; k < 4; k++) S[k] = f_add(S[k], loc[k]);
            }
            or_fr co[4];
            unipoly_from_evals(S, 4, co);
            tp_write(&tr, co, 4); /* full coefficient vector (sumcheck.rs:250) */
            or_fr r = tp_challenge(&tr);
            rs[rd] = r;
#pragma omp parallel for schedule(static)
            for (int c = 0; c <= ni; c++) or_dense_bind(cols[c], 2 * half, &r, cols[c]);
            tr.rounds++;
        }
        for (uint32_t i = 0; i < nv; i++) cl.point[i] = rs[nv - 1 - i]; /* fix_var_bot */
        cl.npoint = (int)nv;
        for (int c = 0; c < ni; c++) cl.evs[c] = cols[c][0];
        cl.nevs = ni;
        tp_write(&tr, cl.evs, ni);
        for (int c = 0; c <= ni; c++) free(cols[c]);
        free(cols);
    }
    for (int li = 0; li < nl; li++) d_free(trace[li]);
    free(trace);
    if (n_msgs) *n_msgs = tr.nmsgs;
    if (final_point) memcpy(final_point, cl.point, (size_t)cl.npoint * sizeof(or_fr));
    if (n_final_point) *n_final_point = (uint32_t)cl.npoint;
    if (final_evs) memcpy(final_evs, cl.evs, 3 * sizeof(or_fr));
    if (tape_used) *tape_used = tr.pos;
    if (rounds) *rounds = tr.rounds;
    return tr.err;
}
