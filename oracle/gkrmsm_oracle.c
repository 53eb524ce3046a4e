/* TEST INFRASTRUCTURE ONLY -- see gkrmsm_oracle.h for scope and parity status.
 *
 * Plain C99 (gcc, unsigned __int128).  Deliberately a different implementation technique from the
 * product (4 x 64-bit limbs, generic Montgomery constants, row-at-a-time loops) so that agreement
 * between the two is evidence, not tautology.
 */
#include "gkrmsm_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* p = BLS12-381 scalar field modulus (ark-bls12-381 Fr) */
static const uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                              0x73eda753299d7d48ULL};
static const uint64_t P_INV = 0xfffffffeffffffffULL; /* -p^-1 mod 2^64 */
static const or_fr ONE = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL,
                           0x1824b159acc5056fULL}}; /* R mod p */
/* Bandersnatch d in Montgomery form: the reference's own constant, src/utils.rs:35 */
static const or_fr COEFF_D = {{12167860994669987632ULL, 4043113551995129031ULL, 6052647550941614584ULL,
                               3904213385886034240ULL}};

void or_coeff_d(or_fr* r) { *r = COEFF_D; }

static int geq_p(const or_fr* a) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > P[i]) return 1;
        if (a->l[i] < P[i]) return 0;
    }
    return 1;
}

static void sub_p(or_fr* a) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - P[i] - b;
        a->l[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}

void or_fr_add(or_fr* r, const or_fr* a, const or_fr* b) {
    u128 c = 0;
    or_fr t;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        t.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_p(&t)) sub_p(&t);
    *r = t;
}

void or_fr_sub(or_fr* r, const or_fr* a, const or_fr* b) {
    u128 br = 0;
    or_fr t;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - br;
        t.l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t.l[i] + P[i];
            t.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    *r = t;
}

void or_fr_neg(or_fr* r, const or_fr* a) {
    or_fr z = {{0, 0, 0, 0}};
    or_fr_sub(r, &z, a);
}

/* Montgomery CIOS, 4 x 64 */
void or_fr_mul(or_fr* r, const or_fr* a, const or_fr* b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * P_INV;
        c = ((u128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    or_fr o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(&o)) sub_p(&o);
    *r = o;
}

void or_fr_inv(or_fr* r, const or_fr* a) {
    /* a^(p-2) */
    uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
    or_fr acc = ONE;
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            or_fr_mul(&acc, &acc, &acc);
            if ((e[i] >> b) & 1) or_fr_mul(&acc, &acc, a);
        }
    *r = acc;
}

/* src/utils.rs:40-43 : t = x.double().double(); -(t + x) */
void or_fr_mul_by_a(or_fr* r, const or_fr* a) {
    or_fr t;
    or_fr_add(&t, a, a);
    or_fr_add(&t, &t, &t);
    or_fr_add(&t, &t, a);
    or_fr_neg(r, &t);
}

void or_fr_mul_by_d(or_fr* r, const or_fr* a) { or_fr_mul(r, a, &COEFF_D); }

static void fr_to_mont(or_fr* r, const or_fr* a) {
    static const or_fr R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL,
                              0x0748d9d99f59ff11ULL}};
    or_fr_mul(r, a, &R2);
}

static void fr_from_mont(or_fr* r, const or_fr* a) {
    or_fr one = {{1, 0, 0, 0}};
    or_fr_mul(r, a, &one);
}

void or_fr_batch(int op, const or_fr* a, const or_fr* b, or_fr* out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        switch (op) {
            case 0: or_fr_add(&out[i], &a[i], &b[i]); break;
            case 1: or_fr_sub(&out[i], &a[i], &b[i]); break;
            case 2: or_fr_mul(&out[i], &a[i], &b[i]); break;
            case 3: or_fr_neg(&out[i], &a[i]); break;
            case 4: or_fr_inv(&out[i], &a[i]); break;
            case 5: fr_to_mont(&out[i], &a[i]); break;
            case 6: fr_from_mont(&out[i], &a[i]); break;
            case 7: or_fr_mul_by_a(&out[i], &a[i]); break;
            default: or_fr_mul_by_d(&out[i], &a[i]); break;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Twisted-Edwards layers: cleanup/utils/twisted_edwards_ops.rs:10-80 */
static void aff_l1(const or_fr* p, or_fr* o) { /* :10-14 */
    or_fr t, u;
    or_fr_mul(&o[0], &p[0], &p[3]);
    or_fr_mul(&o[1], &p[2], &p[1]);
    or_fr_mul(&t, &p[1], &p[3]);
    or_fr_mul(&u, &p[0], &p[2]);
    or_fr_mul_by_a(&u, &u);
    or_fr_sub(&o[2], &t, &u);
}
static void aff_l2(const or_fr* p, or_fr* o) { /* :16-20 */
    or_fr s, m;
    or_fr_add(&s, &p[0], &p[1]);
    or_fr_mul(&m, &p[0], &p[1]);
    or_fr t = p[2];
    o[0] = s; o[1] = t; o[2] = m;
}
static void aff_l3(const or_fr* p, or_fr* o) { /* :22-29 */
    or_fr dxy, m, q, x = p[0], y = p[1];
    or_fr_mul_by_d(&dxy, &p[2]);
    or_fr_sub(&m, &ONE, &dxy);
    or_fr_add(&q, &ONE, &dxy);
    or_fr_mul(&o[0], &m, &x);
    or_fr_mul(&o[1], &q, &y);
    or_fr_mul(&o[2], &m, &q);
}
static void proj_l1(const or_fr* p, or_fr* o) { /* :31-40 */
    or_fr r0, r1, r2, r3, u;
    or_fr_mul(&r0, &p[0], &p[4]);
    or_fr_mul(&r1, &p[3], &p[1]);
    or_fr_mul(&r2, &p[1], &p[4]);
    or_fr_mul(&u, &p[0], &p[3]);
    or_fr_mul_by_a(&u, &u);
    or_fr_sub(&r2, &r2, &u);
    or_fr_mul(&r3, &p[2], &p[5]);
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
static void proj_l2(const or_fr* p, or_fr* o) { /* :43-52 */
    or_fr r0, r1, r2, r3;
    or_fr_add(&r0, &p[0], &p[1]);
    or_fr_mul(&r0, &r0, &p[3]);
    or_fr_mul(&r1, &p[2], &p[3]);
    or_fr_mul(&r2, &p[3], &p[3]);
    or_fr_mul(&r3, &p[0], &p[1]);
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
static void proj_l3(const or_fr* p, or_fr* o) { /* :54-65 */
    or_fr dxy, m, q, x = p[0], y = p[1];
    or_fr_mul_by_d(&dxy, &p[3]);
    or_fr_sub(&m, &p[2], &dxy);
    or_fr_add(&q, &p[2], &dxy);
    or_fr_mul(&o[0], &m, &x);
    or_fr_mul(&o[1], &q, &y);
    or_fr_mul(&o[2], &m, &q);
}
static void tri_l1(const or_fr* p, or_fr* o) { /* :67-80 */
    or_fr in[6];
    memcpy(in, p, 3 * sizeof(or_fr)); memcpy(in + 3, p + 6, 3 * sizeof(or_fr));
    proj_l1(in, o);
    memcpy(in, p + 3, 3 * sizeof(or_fr)); memcpy(in + 3, p + 9, 3 * sizeof(or_fr));
    proj_l1(in, o + 4);
    memcpy(in, p + 6, 6 * sizeof(or_fr));
    proj_l1(in, o + 8);
}

static int prim_ins(int id) {
    static const int t[11] = {0, 4, 3, 3, 6, 4, 4, 12, 1, 1, 3};
    return (id >= 1 && id <= 10) ? t[id] : 0;
}
static int prim_outs(int id) {
    static const int t[11] = {0, 3, 3, 3, 4, 4, 3, 12, 1, 1, 2};
    return (id >= 1 && id <= 10) ? t[id] : 0;
}
static void prim_exec(int id, const or_fr* a, or_fr* o) {
    switch (id) {
        case 1: aff_l1(a, o); break;
        case 2: aff_l2(a, o); break;
        case 3: aff_l3(a, o); break;
        case 4: proj_l1(a, o); break;
        case 5: proj_l2(a, o); break;
        case 6: proj_l3(a, o); break;
        case 7: tri_l1(a, o); break;
        case 8: o[0] = a[0]; break;                       /* algfn.rs:148-150 */
        case 9: {                                          /* algfn.rs:273-275 */
            or_fr s;
            or_fr_mul(&s, &a[0], &a[0]);
            or_fr_sub(&o[0], &s, &a[0]);
        } break;
        case 10: {                                         /* gkr_msm_simple.rs:82-84 */
            or_fr bx, t;
            or_fr_mul(&bx, &a[0], &a[1]);
            or_fr_sub(&t, &a[2], &ONE);
            or_fr_mul(&t, &a[0], &t);
            or_fr_add(&t, &t, &ONE);
            o[0] = bx; o[1] = t;
        } break;
        default: break;
    }
}

int or_fn_n_ins(const or_fn* f) {
    int n = 0;
    for (int s = 0; s < f->nseg; s++) n += prim_ins(f->prim[s]) * f->count[s];
    return n;
}
int or_fn_n_outs(const or_fn* f) {
    int n = 0;
    for (int s = 0; s < f->nseg; s++) n += prim_outs(f->prim[s]) * f->count[s];
    return n;
}
void or_fn_exec(const or_fn* f, const or_fr* in, or_fr* out) {
    int io = 0, oo = 0;
    for (int s = 0; s < f->nseg; s++)
        for (int c = 0; c < f->count[s]; c++) {
            prim_exec(f->prim[s], in + io, out + oo);
            io += prim_ins(f->prim[s]);
            oo += prim_outs(f->prim[s]);
        }
}

/* ------------------------------------------------------------------------------------------------
 * dense polynomials */
void or_dense_bind(const or_fr* in, uint64_t n, const or_fr* t, or_fr* out) { /* sumcheck.rs:160-163 */
    for (uint64_t i = 0; i < n / 2; i++) {
        or_fr d;
        or_fr_sub(&d, &in[2 * i + 1], &in[2 * i]);
        or_fr_mul(&d, t, &d);
        or_fr_add(&out[i], &in[2 * i], &d);
    }
}
void or_dense_make21(or_fr* v, uint64_t n) { /* dense.rs:99-112 */
    for (uint64_t i = 0; i < n / 2; i++) {
        or_fr d;
        or_fr_add(&d, &v[2 * i + 1], &v[2 * i + 1]);
        or_fr_sub(&v[2 * i], &d, &v[2 * i]);
    }
}
void or_dense_bind21(const or_fr* in, uint64_t n, const or_fr* t, or_fr* out) { /* dense.rs:54-61 */
    or_fr tm1;
    or_fr_sub(&tm1, t, &ONE);
    for (uint64_t i = 0; i < n / 2; i++) {
        or_fr d;
        or_fr_sub(&d, &in[2 * i], &in[2 * i + 1]);
        or_fr_mul(&d, &tm1, &d);
        or_fr_add(&out[i], &in[2 * i + 1], &d);
    }
}
void or_dense_map(const or_fn* f, const or_fr* const* ci, uint64_t n, or_fr* const* co) { /* dense.rs:141-184 */
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
    or_fr in[64], out[64];
    for (uint64_t i = 0; i < n; i++) {
        for (int k = 0; k < ni; k++) in[k] = ci[k][i];
        or_fn_exec(f, in, out);
        for (int k = 0; k < no; k++) co[k][i] = out[k];
    }
}
void or_dense_map_split(const or_fn* f, const or_fr* const* ci, uint64_t n, uint32_t lo_bit, uint32_t bundle,
                        or_fr* const* co) { /* dense.rs:115-139 */
    int ni = or_fn_n_ins(f), no = or_fn_n_outs(f);
    or_fr in[64], out[64];
    uint64_t seg = 1ULL << lo_bit;
    uint64_t cnt[2] = {0, 0};
    for (uint64_t i = 0; i < n; i++) {
        for (int k = 0; k < ni; k++) in[k] = ci[k][i];
        or_fn_exec(f, in, out);
        int half = (int)((i / seg) % 2);
        for (int k = 0; k < no; k++) {
            /* bundle interleave: out col k of half h lands in final column 2*(k/bundle)*bundle + h*bundle + k%bundle */
            int col = 2 * (k / (int)bundle) * (int)bundle + half * (int)bundle + k % (int)bundle;
            co[col][cnt[half]] = out[k];
        }
        cnt[half]++;
    }
}
void or_eq_table(const or_fr* mult, const or_fr* pt, uint32_t nvars, or_fr* out) { /* utils.rs:222-250 */
    out[0] = *mult;
    for (uint32_t i = 1; i <= nvars; i++) {
        /* expand in place from the back: level i-1 occupies out[0 .. 2^(i-1)) */
        for (int64_t j = ((int64_t)1 << (i - 1)) - 1; j >= 0; j--) {
            or_fr w = out[j], m;
            or_fr_mul(&m, &pt[i - 1], &w);
            or_fr_sub(&out[2 * j], &w, &m);
            out[2 * j + 1] = m;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Pippenger MSM */
typedef struct { or_fr x, y, z; } pt3;

static pt3 aff_add(const or_fr* x1, const or_fr* y1, const or_fr* x2, const or_fr* y2) {
    or_fr in[4] = {*x1, *y1, *x2, *y2}, a[3], b[3], c[3];
    aff_l1(in, a); aff_l2(a, b); aff_l3(b, c);
    pt3 r = {c[0], c[1], c[2]};
    return r;
}
static pt3 proj_add(const pt3* p, const pt3* q) {
    or_fr in[6] = {p->x, p->y, p->z, q->x, q->y, q->z}, a[4], b[4], c[3];
    proj_l1(in, a); proj_l2(a, b); proj_l3(b, c);
    pt3 r = {c[0], c[1], c[2]};
    return r;
}

int or_msm(const or_fr* pts, const uint64_t* scalars, uint32_t x_log, uint32_t d_log, uint32_t y_size,
           uint32_t y0, uint32_t y1, int threads, uint16_t* digits_out, uint32_t* counter_out, uint32_t* row_len_out,
           or_fr* bx, or_fr* by, or_fr* bz, or_fr* window_cols) {
    if (x_log < 1 || d_log < 2 || d_log > 10 || y1 <= y0 || y1 > y_size || (uint64_t)y_size * d_log > 256) return 1;
    const uint64_t N = 1ULL << x_log;
    const uint32_t nd = 1u << d_log, nwin = y1 - y0, nrows = nwin << d_log;
    const uint32_t mask = nd - 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    uint16_t* digits = digits_out ? digits_out : (uint16_t*)malloc((size_t)nwin * N * sizeof(uint16_t));
    uint32_t* row_len = row_len_out ? row_len_out : (uint32_t*)malloc((size_t)nrows * sizeof(uint32_t));
    uint32_t** rows = (uint32_t**)calloc(nrows, sizeof(uint32_t*));
    pt3* bsum = (pt3*)malloc((size_t)nrows * sizeof(pt3));
    /* the identity column pads of the image: (x,y,z) = (0,1,0), pushforward.rs:380-381 */
    or_fr zero = {{0, 0, 0, 0}};

    /* digits + stable scatter, parallel over windows like the reference (pushforward.rs:401-429) */
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t* cnt = (uint32_t*)calloc(nd, sizeof(uint32_t));
        uint16_t* dg = digits + (size_t)w * N;
        uint32_t bit = (y0 + w) * d_log;
        for (uint64_t x = 0; x < N; x++) { /* pushforward.rs:351-361 */
            const uint64_t* s = scalars + 4 * x;
            uint32_t li = bit >> 6, sh = bit & 63;
            uint64_t v = s[li] >> sh;
            if (sh && li < 3) v |= s[li + 1] << (64 - sh);
            dg[x] = (uint16_t)(v & mask);
            cnt[dg[x]]++;
        }
        for (uint32_t k = 0; k < nd; k++) {
            row_len[w * nd + k] = cnt[k];
            rows[w * nd + k] = (uint32_t*)malloc(((size_t)cnt[k] + 1) * sizeof(uint32_t));
            cnt[k] = 0;
        }
        for (uint64_t x = 0; x < N; x++) { /* pushforward.rs:411-426 */
            uint32_t k = dg[x];
            if (counter_out) counter_out[(size_t)w * N + x] = cnt[k];
            rows[w * nd + k][cnt[k]++] = (uint32_t)x;
        }
        free(cnt);
    }

    /* bucket sums: x_log levels of pairwise adds inside every row, rows re-padded to even length with the
     * image of the pad after every level (vecvec.rs:181-186, 579-594; bintree_add.rs:137-239) */
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t r = 0; r < nrows; r++) {
        uint32_t len = row_len[r];
        uint32_t plen = len + (len & 1);
        pt3 pad;
        pt3* cur = (pt3*)malloc(((size_t)plen / 2 + 2) * sizeof(pt3));
        uint32_t n = 0;
        pt3 res;
        /* level 0: affine */
        pad = aff_add(&zero, &ONE, &zero, &ONE);
        for (uint32_t i = 0; i < plen / 2; i++) {
            const or_fr* a = &pts[2 * (size_t)rows[r][2 * i]];
            uint32_t j1 = 2 * i + 1;
            if (j1 < len) {
                const or_fr* b = &pts[2 * (size_t)rows[r][j1]];
                cur[n++] = aff_add(&a[0], &a[1], &b[0], &b[1]);
            } else {
                cur[n++] = aff_add(&a[0], &a[1], &zero, &ONE);
            }
        }
        if (x_log == 1) {
            res = n ? cur[0] : pad;
        } else {
            for (uint32_t level = 1; level < x_log; level++) {
                if (n & 1) cur[n++] = pad;           /* re-pad to even with the current pad image */
                pt3 npad = proj_add(&pad, &pad);
                if (level + 1 == x_log) {
                    /* last add works on the dense layout: empty rows hold the pad (vecvec.rs:640-646) */
                    res = n ? proj_add(&cur[0], &cur[1]) : npad;
                } else {
                    for (uint32_t i = 0; i < n / 2; i++) cur[i] = proj_add(&cur[2 * i], &cur[2 * i + 1]);
                    n /= 2;
                }
                pad = npad;
            }
        }
        bsum[r] = res;
        free(cur);
        free(rows[r]);
    }
    if (bx) for (uint32_t r = 0; r < nrows; r++) { bx[r] = bsum[r].x; by[r] = bsum[r].y; bz[r] = bsum[r].z; }

    /* bucket reduction per window (windows never mix: all splits are on digit bits).
     * pippenger_ending.rs:46-58 : two HI(y_logsize) splits -> a,b,c,d = digit top bits 00,01,10,11
     * triangle_add.rs:101-158   : layers 0..d-2, L3 fused with a split on the next digit bit (bundle 3)
     * triangle_add.rs:88-99     : last_step */
    if (window_cols) {
        for (uint32_t w = 0; w < nwin; w++) {
            pt3* cur = (pt3*)malloc((size_t)nd * sizeof(pt3));
            pt3* nxt = (pt3*)malloc((size_t)nd * sizeof(pt3));
            memcpy(cur, bsum + (size_t)w * nd, (size_t)nd * sizeof(pt3)); /* array q at cur[q*n + i] */
            uint32_t n = nd >> 2, layers = d_log - 2;
            for (uint32_t l = 0; l <= layers; l++) {
                uint32_t nadds = l + 3;
                for (uint32_t k = 0; k < nadds; k++) {
                    uint32_t ia, ib;
                    if (k == 0) { ia = 0; ib = 2; } else if (k == 1) { ia = 1; ib = 3; }
                    else if (k == 2) { ia = 2; ib = 3; } else { ia = 4 + 2 * (k - 3); ib = ia + 1; }
                    for (uint32_t i = 0; i < n; i++) {
                        pt3 s = proj_add(&cur[ia * n + i], &cur[ib * n + i]);
                        if (l < layers) {
                            uint32_t h = n >> 1;
                            nxt[(2 * k + i / h) * h + i % h] = s;
                        } else {
                            nxt[k] = s;
                        }
                    }
                }
                pt3* t = cur; cur = nxt; nxt = t;
                n >>= 1;
            }
            for (uint32_t k = 0; k <= d_log; k++) {
                window_cols[(size_t)(3 * k + 0) * nwin + w] = cur[k].x;
                window_cols[(size_t)(3 * k + 1) * nwin + w] = cur[k].y;
                window_cols[(size_t)(3 * k + 2) * nwin + w] = cur[k].z;
            }
            free(cur); free(nxt);
        }
    }
    free(bsum); free(rows);
    if (!digits_out) free(digits);
    if (!row_len_out) free(row_len);
    return 0;
}

void or_msm_combine(const or_fr* cols, uint32_t d_log, uint32_t nwin, or_fr* out_xy) { /* pippenger.rs:586-602 */
    pt3 acc = {{{0, 0, 0, 0}}, ONE, ONE};
    for (int64_t w = (int64_t)nwin - 1; w >= 0; w--)
        for (int64_t i = d_log; i >= 1; i--) {
            pt3 p = {cols[(size_t)(3 * i + 0) * nwin + w], cols[(size_t)(3 * i + 1) * nwin + w],
                     cols[(size_t)(3 * i + 2) * nwin + w]};
            acc = proj_add(&acc, &acc);
            acc = proj_add(&acc, &p);
        }
    or_fr zi;
    or_fr_inv(&zi, &acc.z);
    or_fr_mul(&out_xy[0], &acc.x, &zi);
    or_fr_mul(&out_xy[1], &acc.y, &zi);
}
