// BLS12-381 optimal-ate pairing on the host, for the verifier's final check KzgVerifyingKey::verify_pair
// (/root/reference/src/commitments/kzg.rs:61-67: e(A, h0) == e(B, h1)).  The reference gets it from ark-ec 0.4.2 /
// ark-bls12-381 0.4.0 (Cargo.lock:112-113,135-136; not vendored); this is a restatement of the published construction, written
// for clarity, not speed (a verifier computes two Miller loops per proof):
//   Fq2 = Fq[u]/(u^2 + 1),  Fq6 = Fq2[v]/(v^3 - (u + 1)),  Fq12 = Fq6[w]/(w^2 - v)
//   G2 on the M-twist  y^2 = x^3 + 4 (u + 1);  untwist (x', y') -> (x' / w^2, y' / w^3) in E(Fq12): y^2 = x^3 + 4
//   Miller loop over |x| = 0xd201000000010000: affine steps on the twist (slopes in Fq2), lines embedded as sparse Fq12
//   elements (vertical lines dropped: they lie in Fq6), conjugation for the negative x; final exponentiation
//   (q^12 - 1) / r = (q^6 - 1)(q^2 + 1) * (q^4 - q^2 + 1) / r: conjugate / inverse, q^2-Frobenius, windowed power.
// GT equality is only ever tested between outputs of this file, so the result is specified up to the choice of Fq12 basis.
#pragma once
#include <cstdint>

#include "g1.hip.h"

namespace gm {

struct Fq2 { Fq a, b; };            // a + b u
struct Fq6 { Fq2 a, b, c; };        // a + b v + c v^2
struct Fq12 { Fq6 a, b; };          // a + b w
struct G2Aff { Fq2 x, y; };         // (0, 0) = infinity (wire form, like G1)

// ---- Fq2
inline Fq2 fq2_zero() { return {fq_zero(), fq_zero()}; }
inline Fq2 fq2_one() { return {fq_one(), fq_zero()}; }
inline bool fq2_is_zero(const Fq2& x) { return fq_is_zero(x.a) && fq_is_zero(x.b); }
inline bool fq2_eq(const Fq2& x, const Fq2& y) { return fq_eq(x.a, y.a) && fq_eq(x.b, y.b); }
inline Fq2 fq2_add(const Fq2& x, const Fq2& y) { return {fq_add(x.a, y.a), fq_add(x.b, y.b)}; }
inline Fq2 fq2_sub(const Fq2& x, const Fq2& y) { return {fq_sub(x.a, y.a), fq_sub(x.b, y.b)}; }
inline Fq2 fq2_neg(const Fq2& x) { return {fq_neg(x.a), fq_neg(x.b)}; }
inline Fq2 fq2_mul(const Fq2& x, const Fq2& y) {
    const Fq aa = fq_mul(x.a, y.a), bb = fq_mul(x.b, y.b);
    const Fq cross = fq_mul(fq_add(x.a, x.b), fq_add(y.a, y.b));
    return {fq_sub(aa, bb), fq_sub(fq_sub(cross, aa), bb)};
}
inline Fq2 fq2_sqr(const Fq2& x) { return fq2_mul(x, x); }
inline Fq2 fq2_mul_fq(const Fq2& x, const Fq& s) { return {fq_mul(x.a, s), fq_mul(x.b, s)}; }
inline Fq2 fq2_mul_xi(const Fq2& x) { return {fq_sub(x.a, x.b), fq_add(x.a, x.b)}; }  // * (1 + u)
inline Fq2 fq2_inv(const Fq2& x) {
    const Fq n = fq_inv(fq_add(fq_sqr(x.a), fq_sqr(x.b)));
    return {fq_mul(x.a, n), fq_neg(fq_mul(x.b, n))};
}

// ---- Fq6
inline Fq6 fq6_zero() { return {fq2_zero(), fq2_zero(), fq2_zero()}; }
inline Fq6 fq6_one() { return {fq2_one(), fq2_zero(), fq2_zero()}; }
inline bool fq6_is_zero(const Fq6& x) { return fq2_is_zero(x.a) && fq2_is_zero(x.b) && fq2_is_zero(x.c); }
inline bool fq6_eq(const Fq6& x, const Fq6& y) { return fq2_eq(x.a, y.a) && fq2_eq(x.b, y.b) && fq2_eq(x.c, y.c); }
inline Fq6 fq6_add(const Fq6& x, const Fq6& y) { return {fq2_add(x.a, y.a), fq2_add(x.b, y.b), fq2_add(x.c, y.c)}; }
inline Fq6 fq6_sub(const Fq6& x, const Fq6& y) { return {fq2_sub(x.a, y.a), fq2_sub(x.b, y.b), fq2_sub(x.c, y.c)}; }
inline Fq6 fq6_neg(const Fq6& x) { return {fq2_neg(x.a), fq2_neg(x.b), fq2_neg(x.c)}; }
inline Fq6 fq6_mul(const Fq6& x, const Fq6& y) {  // schoolbook with v^3 = xi
    const Fq2 aa = fq2_mul(x.a, y.a), ab = fq2_mul(x.a, y.b), ac = fq2_mul(x.a, y.c);
    const Fq2 ba = fq2_mul(x.b, y.a), bb = fq2_mul(x.b, y.b), bc = fq2_mul(x.b, y.c);
    const Fq2 ca = fq2_mul(x.c, y.a), cb = fq2_mul(x.c, y.b), cc = fq2_mul(x.c, y.c);
    Fq6 r;
    r.a = fq2_add(aa, fq2_mul_xi(fq2_add(bc, cb)));
    r.b = fq2_add(fq2_add(ab, ba), fq2_mul_xi(cc));
    r.c = fq2_add(fq2_add(ac, ca), bb);
    return r;
}
inline Fq6 fq6_mul_v(const Fq6& x) { return {fq2_mul_xi(x.c), x.a, x.b}; }
inline Fq6 fq6_inv(const Fq6& x) {
    // adjugate: t0 = a^2 - xi b c, t1 = xi c^2 - a b, t2 = b^2 - a c; norm = a t0 + xi (c t1 + b t2)
    const Fq2 t0 = fq2_sub(fq2_sqr(x.a), fq2_mul_xi(fq2_mul(x.b, x.c)));
    const Fq2 t1 = fq2_sub(fq2_mul_xi(fq2_sqr(x.c)), fq2_mul(x.a, x.b));
    const Fq2 t2 = fq2_sub(fq2_sqr(x.b), fq2_mul(x.a, x.c));
    const Fq2 n = fq2_inv(fq2_add(fq2_mul(x.a, t0), fq2_mul_xi(fq2_add(fq2_mul(x.c, t1), fq2_mul(x.b, t2)))));
    return {fq2_mul(t0, n), fq2_mul(t1, n), fq2_mul(t2, n)};
}

// ---- Fq12
inline Fq12 fq12_one() { return {fq6_one(), fq6_zero()}; }
inline bool fq12_eq(const Fq12& x, const Fq12& y) { return fq6_eq(x.a, y.a) && fq6_eq(x.b, y.b); }
inline Fq12 fq12_sub(const Fq12& x, const Fq12& y) { return {fq6_sub(x.a, y.a), fq6_sub(x.b, y.b)}; }
inline Fq12 fq12_mul(const Fq12& x, const Fq12& y) {
    const Fq6 aa = fq6_mul(x.a, y.a), bb = fq6_mul(x.b, y.b);
    const Fq6 cross = fq6_mul(fq6_add(x.a, x.b), fq6_add(y.a, y.b));
    return {fq6_add(aa, fq6_mul_v(bb)), fq6_sub(fq6_sub(cross, aa), bb)};
}
inline Fq12 fq12_sqr(const Fq12& x) { return fq12_mul(x, x); }
inline Fq12 fq12_conj(const Fq12& x) { return {x.a, fq6_neg(x.b)}; }  // x^(q^6)
inline Fq12 fq12_inv(const Fq12& x) {
    const Fq6 n = fq6_inv(fq6_sub(fq6_mul(x.a, x.a), fq6_mul_v(fq6_mul(x.b, x.b))));
    return {fq6_mul(x.a, n), fq6_neg(fq6_mul(x.b, n))};
}
inline Fq12 fq12_pow(const Fq12& x, const uint64_t* e, int nlimbs) {
    Fq12 acc = fq12_one();
    for (int i = nlimbs - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            acc = fq12_sqr(acc);
            if ((e[i] >> b) & 1) acc = fq12_mul(acc, x);
        }
    return acc;
}

// ---- G2 (the twist), affine
inline bool g2_aff_is_inf(const G2Aff& p) { return fq2_is_zero(p.x) && fq2_is_zero(p.y); }
inline bool g2_aff_on_curve(const G2Aff& p) {
    if (g2_aff_is_inf(p)) return true;
    const Fq four = fq_dbl(fq_dbl(fq_one()));
    const Fq2 b = fq2_mul_xi(Fq2{four, fq_zero()});
    return fq2_eq(fq2_sqr(p.y), fq2_add(fq2_mul(fq2_sqr(p.x), p.x), b));
}

inline G2Aff g2_generator() {
    static const uint32_t X0[12] = {0xc121bdb8u, 0xd48056c8u, 0xa805bbefu, 0x0bac0326u, 0x7ae3d177u, 0xb4510b64u, 0xfa403b02u, 0xc6e47ad4u, 0x2dc51051u, 0x26080527u, 0xf08f0a91u, 0x024aa2b2u};
    static const uint32_t X1[12] = {0x5d042b7eu, 0xe5ac7d05u, 0x13945d57u, 0x334cf112u, 0xdc7f5049u, 0xb5da61bbu, 0x9920b61au, 0x596bd0d0u, 0x88274f65u, 0x7dacd3a0u, 0x52719f60u, 0x13e02b60u};
    static const uint32_t Y0[12] = {0x08b82801u, 0xe1935486u, 0x3baca289u, 0x923ac9ccu, 0x5160d12cu, 0x6d429a69u, 0x8cbdd3a7u, 0xadfd9baau, 0xda2e351au, 0x8cc9cdc6u, 0x727d6e11u, 0x0ce5d527u};
    static const uint32_t Y1[12] = {0xf05f79beu, 0xaaa9075fu, 0x5cec1da1u, 0x3f370d27u, 0x572e99abu, 0x267492abu, 0x85a763afu, 0xcb3e287eu, 0x2bc28b99u, 0x32acd2b0u, 0x2ea734ccu, 0x0606c4a0u};
    auto mk = [](const uint32_t* l) {
        Fq c;
        for (int i = 0; i < 12; i++) c.l[i] = l[i];
        return fq_to_mont(c);
    };
    return G2Aff{{mk(X0), mk(X1)}, {mk(Y0), mk(Y1)}};
}

// affine chord-and-tangent addition on the twist (one Fq2 inversion each; used for the few scalar multiplications a setup needs)
inline G2Aff g2_add(const G2Aff& p, const G2Aff& q) {
    if (g2_aff_is_inf(p)) return q;
    if (g2_aff_is_inf(q)) return p;
    Fq2 lam;
    if (fq2_eq(p.x, q.x)) {
        if (!fq2_eq(p.y, q.y) || fq2_is_zero(p.y)) return G2Aff{fq2_zero(), fq2_zero()};
        const Fq2 xx = fq2_sqr(p.x);
        lam = fq2_mul(fq2_add(fq2_add(xx, xx), xx), fq2_inv(fq2_add(p.y, p.y)));
    } else {
        lam = fq2_mul(fq2_sub(q.y, p.y), fq2_inv(fq2_sub(q.x, p.x)));
    }
    G2Aff r;
    r.x = fq2_sub(fq2_sub(fq2_sqr(lam), p.x), q.x);
    r.y = fq2_sub(fq2_mul(lam, fq2_sub(p.x, r.x)), p.y);
    return r;
}
// k canonical (not Montgomery), 8 x u32 little-endian
inline G2Aff g2_mul(const G2Aff& p, const uint32_t* k, int nlimbs) {
    G2Aff acc = {fq2_zero(), fq2_zero()};
    for (int i = nlimbs - 1; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            acc = g2_add(acc, acc);
            if ((k[i] >> b) & 1) acc = g2_add(acc, p);
        }
    return acc;
}

// f_{|x|, Q}(P); P affine in G1, Q affine on the twist, neither at infinity.  The slopes are computed on the twist in Fq2 (one
// Fq2 inversion per step); with the untwist X = x' w^-2, Y = y' w^-3 the line through T with twist-slope lam, evaluated at P, is
//   (yP - Y_T) - lam w^-1 (xP - X_T) = yP  +  [(lam x'_T - y'_T) / xi] v w  +  [-lam xP / xi] v^2 w
// (w^-1 = v^2 w / xi, w^-3 = v w / xi): three non-zero coefficients of the twelve.  Vertical lines are dropped (they lie in Fq6).
inline Fq12 miller_loop(const G1Aff& p, const G2Aff& q) {
    static const uint64_t X_ABS = 0xd201000000010000ull;
    const Fq2 xi_inv = fq2_inv(fq2_mul_xi(fq2_one()));
    const Fq2 three = {fq_add(fq_dbl(fq_one()), fq_one()), fq_zero()};
    G2Aff T = q;
    Fq12 f = fq12_one();
    auto line = [&](const G2Aff& A, const Fq2& lam) {
        Fq12 l = {fq6_zero(), fq6_zero()};
        l.a.a.a = p.y;
        l.b.b = fq2_mul(fq2_sub(fq2_mul(lam, A.x), A.y), xi_inv);
        l.b.c = fq2_neg(fq2_mul(fq2_mul_fq(lam, p.x), xi_inv));
        return l;
    };
    auto step = [&](const G2Aff& A, const G2Aff& B, const Fq2& lam) {
        G2Aff r;
        r.x = fq2_sub(fq2_sub(fq2_sqr(lam), A.x), B.x);
        r.y = fq2_sub(fq2_mul(lam, fq2_sub(A.x, r.x)), A.y);
        return r;
    };
    for (int b = 62; b >= 0; b--) {  // bit 63 is the leading one
        const Fq2 lam = fq2_mul(fq2_mul(three, fq2_sqr(T.x)), fq2_inv(fq2_add(T.y, T.y)));
        f = fq12_mul(fq12_sqr(f), line(T, lam));
        T = step(T, T, lam);
        if ((X_ABS >> b) & 1) {
            const Fq2 lam2 = fq2_mul(fq2_sub(q.y, T.y), fq2_inv(fq2_sub(q.x, T.x)));
            f = fq12_mul(f, line(T, lam2));
            T = step(T, q, lam2);
        }
    }
    return fq12_conj(f);  // x < 0
}

// f^(q^2): w^(q^2) = zeta w with zeta = xi^((q^2 - 1) / 6), a primitive sixth root of unity in Fq; the coefficient of w^i
// (tower order: a = (w^0, w^2, w^4), b = (w^1, w^3, w^5)) is multiplied by zeta^i, the Fq2 coefficients themselves are fixed
inline Fq12 fq12_frob2(const Fq12& x) {
    static const uint32_t Z[12] = {0xfffeffffu, 0x2e01ffffu, 0x620a0002u, 0xde17d813u, 0xe6f89688u, 0xddb3a93bu,
                                   0x6a0f77eau, 0xba69c607u, 0xdf76ce51u, 0x5f19672fu, 0x00000000u, 0x00000000u};
    Fq z1;
    for (int i = 0; i < 12; i++) z1.l[i] = Z[i];
    z1 = fq_to_mont(z1);
    const Fq z2 = fq_sqr(z1), z3 = fq_mul(z2, z1), z4 = fq_sqr(z2), z5 = fq_mul(z4, z1);
    Fq12 r;
    r.a.a = x.a.a;
    r.a.b = fq2_mul_fq(x.a.b, z2);
    r.a.c = fq2_mul_fq(x.a.c, z4);
    r.b.a = fq2_mul_fq(x.b.a, z1);
    r.b.b = fq2_mul_fq(x.b.b, z3);
    r.b.c = fq2_mul_fq(x.b.c, z5);
    return r;
}

// x^e with a 4-bit fixed window
inline Fq12 fq12_pow_w4(const Fq12& x, const uint64_t* e, int nlimbs) {
    Fq12 tab[16];
    tab[0] = fq12_one();
    for (int i = 1; i < 16; i++) tab[i] = fq12_mul(tab[i - 1], x);
    Fq12 acc = fq12_one();
    for (int i = nlimbs - 1; i >= 0; i--)
        for (int b = 60; b >= 0; b -= 4) {
            acc = fq12_sqr(fq12_sqr(fq12_sqr(fq12_sqr(acc))));
            const unsigned d = (unsigned)(e[i] >> b) & 15u;
            if (d) acc = fq12_mul(acc, tab[d]);
        }
    return acc;
}

// (q^12 - 1) / r = (q^6 - 1) (q^2 + 1) * (q^4 - q^2 + 1) / r
inline Fq12 final_exponentiation(const Fq12& f) {
    static const uint64_t EXP_HARD[20] = {0xe516c3f438e3ba79ull, 0xfa9912aae208ccf1ull, 0x905ce937335d5b68ull, 0xc71a2629b0dea236ull,
                                          0x83774940996754c8ull, 0x21d160aeb6a1e799ull, 0x2ed0b283ed237db4ull, 0x915c97f36c6f1821ull,
                                          0x67f17fcbde783765ull, 0x2378b9039096d1b7ull, 0x7988f8761bdc51dcull, 0x2076995003fc77a1ull,
                                          0x827eca0ba621315bull, 0xe5a72bce8d63cb9full, 0xf68f7764c28b6f8aull, 0x2f230063cf081517ull,
                                          0x94506632528d6a9aull, 0xd3cde88eeb996ca3ull, 0xc0bd38c3195c899eull, 0x000f686b3d807d01ull};
    const Fq12 t = fq12_mul(fq12_conj(f), fq12_inv(f));   // f^(q^6 - 1)
    const Fq12 u = fq12_mul(fq12_frob2(t), t);            // ^(q^2 + 1)
    return fq12_pow_w4(u, EXP_HARD, 20);
}

// e(P, Q); infinity on either side gives 1
inline Fq12 pairing(const G1Aff& p, const G2Aff& q) {
    if (g1_aff_is_inf(p) || g2_aff_is_inf(q)) return fq12_one();
    return final_exponentiation(miller_loop(p, q));
}

// prod_i e(P_i, Q_i) == 1, with one final exponentiation
inline bool pairing_product_is_one(const G1Aff* ps, const G2Aff* qs, int n) {
    Fq12 f = fq12_one();
    for (int i = 0; i < n; i++) {
        if (g1_aff_is_inf(ps[i]) || g2_aff_is_inf(qs[i])) continue;
        f = fq12_mul(f, miller_loop(ps[i], qs[i]));
    }
    return fq12_eq(final_exponentiation(f), fq12_one());
}

}  // namespace gm
