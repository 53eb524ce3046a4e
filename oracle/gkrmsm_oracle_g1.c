/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the BLS12-381 G1 side of the hot path (never linked into the product).
 *
 * Restates, in plain C99 (6 x 64-bit limbs, unsigned __int128 -- a different technique from the product's 12 x 32-bit
 * device code), what the reference does with G1 on this path:
 *   msm_bigint_wnaf_nonaff + make_digits + ln_without_floats      src/msm_nonaffine.rs:89-161, 275-322
 *     (the branch `Projective<g1::Config>` takes: NEGATION_IS_CHEAP, msm_nonaffine.rs:45-46); windows in parallel like the
 *     reference's cfg_into_iter!(0..digits_count) (:123)
 *   the G1 part of PushForwardState::new                           src/cleanup/protocols/pushforward/pushforward.rs:395-456, 504-524
 *     (parallel over the y_size digit rows like the reference's par_chunks_mut, :401)
 *   binary_msm                                                     src/binary_msm.rs:19-29
 * Point arithmetic: ark-ec 0.4.2 is not vendored; the standard a = 0 Jacobian formulas are used and results are compared as
 * group elements (affine), which is `Projective`'s own equality.  Pinned by tests/test_g1_cpu.py against the Python
 * big-int oracle (itself pinned by public BLS12-381 vectors).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fq;
typedef struct { fq x, y; } g1a;       /* affine, (0,0) = infinity */
typedef struct { fq x, y, z; } g1j;    /* Jacobian, z = 0 = infinity */

static const uint64_t Q[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                              0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t Q_INV = 0x89f3fffcfffcfffdULL; /* -q^-1 mod 2^64 */
static const fq FQ_ONE = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                           0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};

static int fq_is_zero(const fq* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0; }


static int geq_q(const uint64_t* a) {
    for (int i = 5; i >= 0; i--) {
        if (a[i] > Q[i]) return 1;
        if (a[i] < Q[i]) return 0;
    }
    return 1;
}
static void sub_q(uint64_t* a) {
    u128 b = 0;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)a[i] - Q[i] - b;
        a[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
static void fq_add(fq* r, const fq* a, const fq* b) {
    u128 c = 0;
    fq t;
    for (int i = 0; i < 6; i++) {
        c += (u128)a->l[i] + b->l[i];
        t.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_q(t.l)) sub_q(t.l);
    *r = t;
}
static void fq_sub(fq* r, const fq* a, const fq* b) {
    u128 br = 0;
    fq t;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - br;
        t.l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 6; i++) {
            c += (u128)t.l[i] + Q[i];
            t.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    *r = t;
}
static void fq_neg(fq* r, const fq* a) {
    fq z;
    memset(&z, 0, sizeof z);
    fq_sub(r, &z, a);
}
/* Montgomery product, operand scanning: full 12-limb product first, then 6 reduction sweeps (SOS) */
static void fq_mul(fq* r, const fq* a, const fq* b) {
    uint64_t t[13];
    memset(t, 0, sizeof t);
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) {
            c += (u128)a->l[j] * b->l[i] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 6] = (uint64_t)c;
    }
    for (int i = 0; i < 6; i++) {
        const uint64_t m = t[i] * Q_INV;
        u128 c = 0;
        for (int j = 0; j < 6; j++) {
            c += (u128)m * Q[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        for (int k = i + 6; k < 13 && c; k++) {
            c += t[k];
            t[k] = (uint64_t)c;
            c >>= 64;
        }
    }
    fq o;
    memcpy(o.l, t + 6, 48);
    if (t[12] || geq_q(o.l)) sub_q(o.l);
    *r = o;
}
static void fq_sqr(fq* r, const fq* a) { fq_mul(r, a, a); }
static void fq_dbl(fq* r, const fq* a) { fq_add(r, a, a); }
static void fq_inv(fq* r, const fq* a) {
    /* a^(q-2) */
    uint64_t e[6];
    memcpy(e, Q, 48);
    e[0] -= 2;
    fq acc = FQ_ONE;
    for (int i = 5; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            fq_sqr(&acc, &acc);
            if ((e[i] >> b) & 1) fq_mul(&acc, &acc, a);
        }
    *r = acc;
}

static int g1a_is_inf(const g1a* p) { return fq_is_zero(&p->x) && fq_is_zero(&p->y); }
static int g1j_is_inf(const g1j* p) { return fq_is_zero(&p->z); }
static void g1j_set_inf(g1j* p) { memset(p, 0, sizeof *p); p->y = FQ_ONE; }
static void g1j_from_aff(g1j* r, const g1a* p) {
    if (g1a_is_inf(p)) { g1j_set_inf(r); return; }
    r->x = p->x; r->y = p->y; r->z = FQ_ONE;
}

static void g1j_dbl(g1j* r, const g1j* p) {
    if (g1j_is_inf(p)) { *r = *p; return; }
    fq A, B, C, D, E, F, t;
    fq_sqr(&A, &p->x); fq_sqr(&B, &p->y); fq_sqr(&C, &B);
    fq_add(&t, &p->x, &B); fq_sqr(&t, &t); fq_sub(&t, &t, &A); fq_sub(&t, &t, &C); fq_dbl(&D, &t);
    fq_dbl(&E, &A); fq_add(&E, &E, &A);
    fq_sqr(&F, &E);
    g1j o;
    fq_dbl(&t, &D); fq_sub(&o.x, &F, &t);
    fq_mul(&o.z, &p->y, &p->z); fq_dbl(&o.z, &o.z);
    fq_sub(&t, &D, &o.x); fq_mul(&t, &E, &t);
    fq_dbl(&C, &C); fq_dbl(&C, &C); fq_dbl(&C, &C);
    fq_sub(&o.y, &t, &C);
    *r = o;
}

static void g1j_add(g1j* r, const g1j* p, const g1j* q) {
    if (g1j_is_inf(p)) { *r = *q; return; }
    if (g1j_is_inf(q)) { *r = *p; return; }
    fq z1z1, z2z2, u1, u2, s1, s2, h, rr, i, j, v, t;
    fq_sqr(&z1z1, &p->z); fq_sqr(&z2z2, &q->z);
    fq_mul(&u1, &p->x, &z2z2); fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s1, &p->y, &q->z); fq_mul(&s1, &s1, &z2z2);
    fq_mul(&s2, &q->y, &p->z); fq_mul(&s2, &s2, &z1z1);
    fq_sub(&h, &u2, &u1); fq_sub(&rr, &s2, &s1);
    if (fq_is_zero(&h)) {
        if (fq_is_zero(&rr)) g1j_dbl(r, p); else g1j_set_inf(r);
        return;
    }
    fq_dbl(&rr, &rr);
    fq_dbl(&i, &h); fq_sqr(&i, &i);
    fq_mul(&j, &h, &i); fq_mul(&v, &u1, &i);
    g1j o;
    fq_sqr(&o.x, &rr); fq_sub(&o.x, &o.x, &j); fq_dbl(&t, &v); fq_sub(&o.x, &o.x, &t);
    fq_sub(&t, &v, &o.x); fq_mul(&o.y, &rr, &t); fq_mul(&t, &s1, &j); fq_dbl(&t, &t); fq_sub(&o.y, &o.y, &t);
    fq_add(&t, &p->z, &q->z); fq_sqr(&t, &t); fq_sub(&t, &t, &z1z1); fq_sub(&t, &t, &z2z2); fq_mul(&o.z, &t, &h);
    *r = o;
}

static void g1j_add_mixed(g1j* r, const g1j* p, const g1a* q) {
    if (g1a_is_inf(q)) { *r = *p; return; }
    if (g1j_is_inf(p)) { g1j_from_aff(r, q); return; }
    fq z1z1, u2, s2, h, rr, i, j, v, t;
    fq_sqr(&z1z1, &p->z);
    fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s2, &q->y, &p->z); fq_mul(&s2, &s2, &z1z1);
    fq_sub(&h, &u2, &p->x); fq_sub(&rr, &s2, &p->y);
    if (fq_is_zero(&h)) {
        if (fq_is_zero(&rr)) g1j_dbl(r, p); else g1j_set_inf(r);
        return;
    }
    fq_dbl(&rr, &rr);
    fq_dbl(&i, &h); fq_sqr(&i, &i);
    fq_mul(&j, &h, &i); fq_mul(&v, &p->x, &i);
    g1j o;
    fq_sqr(&o.x, &rr); fq_sub(&o.x, &o.x, &j); fq_dbl(&t, &v); fq_sub(&o.x, &o.x, &t);
    fq_sub(&t, &v, &o.x); fq_mul(&o.y, &rr, &t); fq_mul(&t, &p->y, &j); fq_dbl(&t, &t); fq_sub(&o.y, &o.y, &t);
    fq_mul(&o.z, &p->z, &h); fq_dbl(&o.z, &o.z);
    *r = o;
}

static void g1j_neg(g1j* r, const g1j* p) { *r = *p; fq_neg(&r->y, &p->y); }

static void g1j_to_aff(g1a* r, const g1j* p) {
    if (g1j_is_inf(p)) { memset(r, 0, sizeof *r); return; }
    fq zi, zi2, zi3;
    fq_inv(&zi, &p->z); fq_sqr(&zi2, &zi); fq_mul(&zi3, &zi2, &zi);
    fq_mul(&r->x, &p->x, &zi2); fq_mul(&r->y, &p->y, &zi3);
}

/* ---------------------------------------------------------------- msm_nonaffine.rs */
static unsigned log2_ceil(uint64_t a) { unsigned l = 0; while ((1ull << l) < a) l++; return l; }
static unsigned ln_without_floats(uint64_t a) { return log2_ceil(a) * 69 / 100; }            /* :319-322 */
static unsigned num_bits_256(const uint64_t* s) {
    for (int i = 3; i >= 0; i--) if (s[i]) return 64 * i + 64 - __builtin_clzll(s[i]);
    return 0;
}
/* make_digits, msm_nonaffine.rs:275-314 */
static void make_digits(const uint64_t* scalar, unsigned w, unsigned num_bits, int64_t* digits, unsigned count) {
    const uint64_t radix = 1ull << w, mask = radix - 1;
    uint64_t carry = 0;
    (void)num_bits;
    for (unsigned i = 0; i < count; i++) {
        const unsigned off = i * w, ui = off / 64, bi = off % 64;
        uint64_t buf;
        if (bi < 64 - w || ui == 3) buf = scalar[ui] >> bi;
        else buf = (scalar[ui] >> bi) | (scalar[ui + 1] << (64 - bi));
        const uint64_t coef = carry + (buf & mask);
        carry = (coef + radix / 2) >> w;
        digits[i] = (int64_t)coef - (int64_t)(carry << w);
    }
    digits[count - 1] += (int64_t)(carry << w);
}

/* msm_bigint_wnaf_nonaff (msm_nonaffine.rs:89-161).  bases: n Jacobian points, scalars: n canonical bigints (4 x u64).
 * out: affine.  threads: OpenMP threads over the windows (the reference's rayon granularity). */
int or_g1_msm_wnaf_nonaff(const g1j* bases, const uint64_t* scalars, uint64_t n, int threads, g1a* out) {
    unsigned max_bits = 1;
    for (uint64_t i = 0; i < n; i++) {
        const unsigned b = num_bits_256(scalars + 4 * i);
        if (b > max_bits) max_bits = b;
        if (max_bits > 60) { max_bits = 255; break; }      /* "hack for early exit", :100-103 */
    }
    const unsigned c = n < 32 ? 3 : ln_without_floats(n) + 2;
    const unsigned count = (max_bits + c - 1) / c;
    int64_t* digs = (int64_t*)malloc((size_t)n * count * sizeof(int64_t));
    g1j* sums = (g1j*)malloc(count * sizeof(g1j));
    if (!digs || !sums) { free(digs); free(sums); return 1; }
    for (uint64_t i = 0; i < n; i++) make_digits(scalars + 4 * i, c, max_bits, digs + i * count, count);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (unsigned w = 0; w < count; w++) {
        const size_t nb = (size_t)1 << c;
        g1j* buckets = (g1j*)malloc(nb * sizeof(g1j));
        for (size_t b = 0; b < nb; b++) g1j_set_inf(&buckets[b]);
        for (uint64_t i = 0; i < n; i++) {
            const int64_t s = digs[i * count + w];
            if (s > 0) g1j_add(&buckets[s - 1], &buckets[s - 1], &bases[i]);
            else if (s < 0) { g1j nb_; g1j_neg(&nb_, &bases[i]); g1j_add(&buckets[-s - 1], &buckets[-s - 1], &nb_); }
        }
        g1j run, res;
        g1j_set_inf(&run); g1j_set_inf(&res);
        for (size_t b = nb; b-- > 0;) { g1j_add(&run, &run, &buckets[b]); g1j_add(&res, &res, &run); }
        sums[w] = res;
        free(buckets);
    }
    g1j total;
    g1j_set_inf(&total);
    for (unsigned w = count; w-- > 1;) {
        g1j_add(&total, &total, &sums[w]);
        for (unsigned k = 0; k < c; k++) g1j_dbl(&total, &total);
    }
    g1j_add(&total, &total, &sums[0]);
    g1j_to_aff(out, &total);
    free(digs); free(sums);
    return 0;
}

/* <G1 as VariableBaseMSM>::msm over affine bases (KzgProvingKey::commit, kzg.rs:123-126), via the same restatement */
int or_g1_msm_affine(const g1a* bases, const uint64_t* scalars, uint64_t n, int threads, g1a* out) {
    g1j* jb = (g1j*)malloc((size_t)(n ? n : 1) * sizeof(g1j));
    if (!jb) return 1;
    for (uint64_t i = 0; i < n; i++) g1j_from_aff(&jb[i], &bases[i]);
    const int rc = or_g1_msm_wnaf_nonaff(jb, scalars, n, threads, out);
    free(jb);
    return rc;
}

/* binary_msm (binary_msm.rs:19-29): tables = n_chunks x (2^gamma - 1) affine */
int or_g1_binary_msm(const uint8_t* coefs, const g1a* tables, uint64_t n_chunks, unsigned gamma, g1a* out) {
    const uint64_t tl = (1ull << gamma) - 1;
    g1j acc;
    g1j_set_inf(&acc);
    for (uint64_t t = 0; t < n_chunks; t++)
        if (coefs[t]) g1j_add_mixed(&acc, &acc, &tables[t * tl + coefs[t] - 1]);
    g1j_to_aff(out, &acc);
    return 0;
}

/* PushForwardState::new, G1 part (pushforward.rs:395-456, 504-524).  digits (u16) / counter (u32): [y][x];
 * basis: >= 2^(x_log + clm) affine points.  d_outer: n_mat * 2^d_log Jacobian, c_outer: n_mat * c_stride Jacobian
 * (c_stride = max counter + 1 over all rows, computed here and returned), d_comm / c_comm: n_mat affine. */
int or_g1_pushforward_outer(const uint16_t* digits, const uint32_t* counter, const g1a* basis, unsigned x_log, unsigned d_log,
                            unsigned y_size, unsigned clm, int threads, g1j* d_outer, g1j* c_outer, uint64_t c_cap,
                            uint32_t* c_stride, g1a* d_comm, g1a* c_comm) {
    const uint64_t N = 1ull << x_log;
    const unsigned nd = 1u << d_log, cm = 1u << clm, n_mat = (y_size + cm - 1) / cm;
    uint32_t cmax = 0;
    for (uint64_t t = 0; t < N * y_size; t++) if (counter[t] + 1 > cmax) cmax = counter[t] + 1;
    *c_stride = cmax;
    if ((uint64_t)n_mat * cmax > c_cap) return 2;
    g1j* d_rows = (g1j*)malloc((size_t)y_size * nd * sizeof(g1j));
    g1j* c_rows = (g1j*)malloc((size_t)y_size * cmax * sizeof(g1j));
    if (!d_rows || !c_rows) { free(d_rows); free(c_rows); return 1; }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (unsigned y = 0; y < y_size; y++) {
        g1j* db = d_rows + (size_t)y * nd;
        g1j* cb = c_rows + (size_t)y * cmax;
        for (unsigned i = 0; i < nd; i++) g1j_set_inf(&db[i]);
        for (uint32_t i = 0; i < cmax; i++) g1j_set_inf(&cb[i]);
        for (uint64_t x = 0; x < N; x++) {
            const g1a* pt = &basis[x + N * (y % cm)];
            g1j_add_mixed(&db[digits[y * N + x]], &db[digits[y * N + x]], pt);   /* d_outer_buckets[d] += point */
            g1j_add_mixed(&cb[counter[y * N + x]], &cb[counter[y * N + x]], pt); /* c_outer_buckets[c] += point */
        }
    }
    for (unsigned m = 0; m < n_mat; m++) {
        for (unsigned i = 0; i < nd; i++) {
            g1j acc;
            g1j_set_inf(&acc);
            for (unsigned y = m * cm; y < (m + 1) * cm && y < y_size; y++) g1j_add(&acc, &acc, &d_rows[(size_t)y * nd + i]);
            d_outer[(size_t)m * nd + i] = acc;
        }
        for (uint32_t i = 0; i < cmax; i++) {
            g1j acc;
            g1j_set_inf(&acc);
            for (unsigned y = m * cm; y < (m + 1) * cm && y < y_size; y++) g1j_add(&acc, &acc, &c_rows[(size_t)y * cmax + i]);
            c_outer[(size_t)m * cmax + i] = acc;
        }
    }
    free(d_rows); free(c_rows);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (unsigned k = 0; k < 2 * n_mat; k++) {
        const unsigned m = k >> 1;
        const g1j* b = (k & 1) ? c_outer + (size_t)m * cmax : d_outer + (size_t)m * nd;
        const uint32_t len = (k & 1) ? cmax : nd;
        g1j acc, run;
        g1j_set_inf(&acc); g1j_set_inf(&run);
        for (uint32_t i = 0; i + 1 < len; i++) { g1j_add(&run, &run, &b[len - i - 1]); g1j_add(&acc, &acc, &run); }
        g1j_to_aff((k & 1) ? &c_comm[m] : &d_comm[m], &acc);
    }
    return 0;
}

int or_g1_to_affine(const g1j* in, uint64_t n, g1a* out) {
    for (uint64_t i = 0; i < n; i++) g1j_to_aff(&out[i], &in[i]);
    return 0;
}
int or_g1_add_aff(const g1a* a, const g1a* b, uint64_t n, g1a* out) {
    for (uint64_t i = 0; i < n; i++) {
        g1j t;
        g1j_from_aff(&t, &a[i]);
        g1j_add_mixed(&t, &t, &b[i]);
        g1j_to_aff(&out[i], &t);
    }
    return 0;
}
int or_fq_mul(const fq* a, const fq* b, uint64_t n, fq* out) {
    for (uint64_t i = 0; i < n; i++) fq_mul(&out[i], &a[i], &b[i]);
    return 0;
}
