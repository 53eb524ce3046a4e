// BLS12-381 G1 (y^2 = x^3 + 4 over Fq) point arithmetic, host + device.
//
// The reference reaches G1 only through ark-ec 0.4.2 (`Projective<g1::Config>` = Jacobian (X, Y, Z), x = X/Z^2,
// y = Y/Z^3; `Affine{x, y, infinity}`); that crate is not vendored (Cargo.lock:135-136).  `Projective` equality there is
// equality of the represented group element, so the G1 outputs of this library are specified -- and tested -- as group
// elements (affine coordinates), not as a particular (X, Y, Z) triple.  Formulas: the standard a = 0 Jacobian ones
// (EFD dbl-2009-l, add-2007-bl, madd-2007-bl, mmadd-2007-bl), complete via explicit special cases.
//
// Wire forms at the C ABI (the Rust shim marshals; ark's struct layouts are not repr(C)):
//   affine:   x, y  Montgomery 6 x u64 each = 96 bytes; the point at infinity is (0, 0) (not on the curve)
//   jacobian: X, Y, Z = 144 bytes; infinity is Z = 0
#pragma once
#include "fq.hip.h"
#if defined(__HIPCC__)
#include "fq14.hip.h"
#endif

namespace gm {

struct G1Aff {
    Fq x, y;
};
struct G1Jac {
    Fq x, y, z;
};

GM_HD bool g1_aff_is_inf(const G1Aff& p) { return fq_is_zero(p.x) && fq_is_zero(p.y); }
GM_HD bool g1_is_inf(const G1Jac& p) { return fq_is_zero(p.z); }

GM_HD G1Jac g1_inf() {
    G1Jac r;
    r.x = fq_zero(); r.y = fq_one(); r.z = fq_zero();
    return r;
}

GM_HD G1Jac g1_from_aff(const G1Aff& p) {
    if (g1_aff_is_inf(p)) return g1_inf();
    G1Jac r;
    r.x = p.x; r.y = p.y; r.z = fq_one();
    return r;
}

GM_HD G1Aff g1_aff_neg(const G1Aff& p) {
    G1Aff r;
    r.x = p.x; r.y = fq_neg(p.y);
    return r;
}

GM_HD G1Jac g1_neg(const G1Jac& p) {
    G1Jac r = p;
    r.y = fq_neg(p.y);
    return r;
}

// dbl-2009-l (a = 0): 2M + 5S
GM_HD G1Jac g1_dbl(const G1Jac& p) {
    if (g1_is_inf(p)) return p;
    const Fq A = fq_sqr(p.x), B = fq_sqr(p.y), C = fq_sqr(B);
    const Fq D = fq_dbl(fq_sub(fq_sub(fq_sqr(fq_add(p.x, B)), A), C));
    const Fq E = fq_add(fq_dbl(A), A), F = fq_sqr(E);
    G1Jac r;
    r.x = fq_sub(F, fq_dbl(D));
    const Fq C8 = fq_dbl(fq_dbl(fq_dbl(C)));
    r.z = fq_dbl(fq_mul(p.y, p.z));
    r.y = fq_sub(fq_mul(E, fq_sub(D, r.x)), C8);
    return r;
}

// shared tail of the three addition formulas: H = U2 - U1, rr = 2 (S2 - S1) given; X1' = U1, Y1' = S1, Zmul = the factor
// Z3 gets besides H (2 Z1 Z2, computed by the caller)
GM_HD G1Jac g1_add_tail(const Fq& U1, const Fq& S1, const Fq& H, const Fq& rr, const Fq& zfac) {
    const Fq I = fq_sqr(fq_dbl(H)), J = fq_mul(H, I), V = fq_mul(U1, I);
    G1Jac r;
    r.x = fq_sub(fq_sub(fq_sqr(rr), J), fq_dbl(V));
    r.y = fq_sub(fq_mul(rr, fq_sub(V, r.x)), fq_dbl(fq_mul(S1, J)));
    r.z = fq_mul(zfac, H);
    return r;
}

// add-2007-bl: 11M + 5S
GM_HD G1Jac g1_add_c(const G1Jac& p, const G1Jac& q) {
    if (g1_is_inf(p)) return q;
    if (g1_is_inf(q)) return p;
    const Fq Z1Z1 = fq_sqr(p.z), Z2Z2 = fq_sqr(q.z);
    const Fq U1 = fq_mul(p.x, Z2Z2), U2 = fq_mul(q.x, Z1Z1);
    const Fq S1 = fq_mul(fq_mul(p.y, q.z), Z2Z2), S2 = fq_mul(fq_mul(q.y, p.z), Z1Z1);
    const Fq H = fq_sub(U2, U1), d = fq_sub(S2, S1);
    if (fq_is_zero(H)) return fq_is_zero(d) ? g1_dbl(p) : g1_inf();
    const Fq zfac = fq_sub(fq_sub(fq_sqr(fq_add(p.z, q.z)), Z1Z1), Z2Z2);  // 2 Z1 Z2
    return g1_add_tail(U1, S1, H, fq_dbl(d), zfac);
}

// madd-2007-bl (Z2 = 1): 7M + 4S
GM_HD G1Jac g1_add_mixed_c(const G1Jac& p, const G1Aff& q) {
    if (g1_aff_is_inf(q)) return p;
    if (g1_is_inf(p)) return g1_from_aff(q);
    const Fq Z1Z1 = fq_sqr(p.z);
    const Fq U2 = fq_mul(q.x, Z1Z1), S2 = fq_mul(fq_mul(q.y, p.z), Z1Z1);
    const Fq H = fq_sub(U2, p.x), d = fq_sub(S2, p.y);
    if (fq_is_zero(H)) return fq_is_zero(d) ? g1_dbl(p) : g1_inf();
    return g1_add_tail(p.x, p.y, H, fq_dbl(d), fq_dbl(p.z));
}

// mmadd-2007-bl (Z1 = Z2 = 1): 4M + 2S
GM_HD G1Jac g1_add_aff_c(const G1Aff& p, const G1Aff& q) {
    if (g1_aff_is_inf(p)) return g1_from_aff(q);
    if (g1_aff_is_inf(q)) return g1_from_aff(p);
    const Fq H = fq_sub(q.x, p.x), d = fq_sub(q.y, p.y);
    if (fq_is_zero(H)) return fq_is_zero(d) ? g1_dbl(g1_from_aff(p)) : g1_inf();
    Fq two = fq_dbl(fq_one());
    return g1_add_tail(p.x, p.y, H, fq_dbl(d), two);
}

#if defined(__HIPCC__) && !defined(GM_G1_FQ12)
// ---- the same three additions with the field arithmetic in the 14 x 28-bit form (fq14.hip.h): identical stored coordinates
// (the formulas are the ones above, Z3 = 2 Z1 Z2 H written as a product), 1.2x the product rate, cheaper squares.  Bounds per line:
// S = value / q, L = limb bound below the top limb; loads and products come out with L < 2^28, S <= 1.1.
// The P = +-Q cases are detected on H = U2 - U1 (fq14_maybe_zero: exact for "no", rare false "maybe") and handed, with the
// original operands, to the 12 x 32 formulas above.
// A point in this form: limbs normalised (< 2^28), S <= 10.2 (what an addition leaves) or 1.1 (a load); infinity is z = 0 exactly
// (a point at infinity only ever enters as a converted input or through the exact path, both of which give all-zero limbs; the Z3
// of an addition with H != 0 is a product of non-zero residues).  The sum-by-key tree keeps its intermediate cells like this
// (g1.hip: 42 words per cell) and converts to the canonical wire form once, at the end.
struct G1P14 {
    Fq14 x, y, z;
};
__device__ __forceinline__ bool fq14_limbs_zero(const Fq14& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) o |= a.l[i];
    return o == 0;
}
__device__ __forceinline__ G1P14 g1p14_from(const G1Jac& p) {
    G1P14 r;
    r.x = fq14_from(p.x); r.y = fq14_from(p.y); r.z = fq14_from(p.z);
    return r;
}
__device__ __forceinline__ G1Jac g1p14_to(const G1P14& p) {
    G1Jac r;
    r.x = fq14_to(p.x); r.y = fq14_to(p.y); r.z = fq14_to(p.z);
    return r;
}
// shared tail; independent products go two at a time (fq14_mul2 / fq14_sqr2).
// ZKIND 0: Z3 = 2 H (affine + affine);  1: Z3 = 2 za H (mixed, za = Z1);  2: Z3 = 2 za zb H (general, za zb = Z1 Z2)
// U1, S1, za, zb: limbs < 2^28, S <= 10.2.
template <int ZKIND>
__device__ __forceinline__ G1P14 g1_tail14(const Fq14& U1, const Fq14& S1, const Fq14& H, const Fq14& d, const Fq14& za, const Fq14& zb) {
    // H, d: a - b + 4 q, L < 2^29.6, S < 5.2
    Fq14 HH, dd, J, V, dW, YJ, ZZ, ZH;
    fq14_sqr2(H, d, HH, dd);                                          // S 1.02; rr^2 = 4 dd
    const Fq14 I = fq14_shl<2>(HH);                                   // (2H)^2: L < 2^30, S 4.1
    fq14_mul2(H, I, U1, I, J, V);                                     // 2^59.6: S 1.02
    const Fq14 T = fq14_norm(fq14_add(J, fq14_shl<1>(V)));            // J + 2V: S 3.1
    G1P14 r;
    r.x = fq14_norm(fq14_sub4(fq14_shl<2>(dd), T));                   // S 8.2
    const Fq14 W = fq14_sub16(V, r.x);                                // L < 2^29.6, S 17.1
    if (ZKIND == 2) {
        fq14_mul2(S1, J, za, zb, YJ, ZZ);
        fq14_mul2(d, W, ZZ, H, dW, ZH);                               // 2^59.2: S 1.04
    } else if (ZKIND == 1) {
        fq14_mul2(S1, J, za, H, YJ, ZH);
        dW = fq14_mul(d, W);
    } else {
        fq14_mul2(d, W, S1, J, dW, YJ);
    }
    r.y = fq14_norm(fq14_shl<1>(fq14_sub4(dW, YJ)));                  // rr (V - X3) - 2 S1 J: S 10.1
    r.z = fq14_norm(fq14_shl<1>(ZKIND == 0 ? H : ZH));                // S 10.2
    return r;
}
// general addition on this form (inputs: limbs < 2^28, S <= 10.2)
__device__ __forceinline__ G1P14 g1_add14p(const G1P14& p, const G1P14& q) {
    if (fq14_limbs_zero(p.z)) return q;
    if (fq14_limbs_zero(q.z)) return p;
    Fq14 Z1Z1, Z2Z2, U1, U2, A, B, S1, S2;
    fq14_sqr2(p.z, q.z, Z1Z1, Z2Z2);
    fq14_mul2(p.x, Z2Z2, q.x, Z1Z1, U1, U2);
    const Fq14 H = fq14_sub4(U2, U1);
    if (fq14_maybe_zero(H)) return g1p14_from(g1_add_c(g1p14_to(p), g1p14_to(q)));
    fq14_mul2(p.y, q.z, q.y, p.z, A, B);
    fq14_mul2(A, Z2Z2, B, Z1Z1, S1, S2);
    return g1_tail14<2>(U1, S1, H, fq14_sub4(S2, S1), p.z, q.z);
}
__device__ __forceinline__ G1P14 g1_add_aff14p(const G1Aff& p, const G1Aff& q) {
    if (g1_aff_is_inf(p)) return g1p14_from(g1_from_aff(q));
    if (g1_aff_is_inf(q)) return g1p14_from(g1_from_aff(p));
    const Fq14 X1 = fq14_from(p.x), Y1 = fq14_from(p.y);
    const Fq14 H = fq14_sub4(fq14_from(q.x), X1);
    if (fq14_maybe_zero(H)) return g1p14_from(g1_add_aff_c(p, q));
    return g1_tail14<0>(X1, Y1, H, fq14_sub4(fq14_from(q.y), Y1), X1, X1);
}
// the wire-form entry points
__device__ __forceinline__ G1Jac g1_add14(const G1Jac& p, const G1Jac& q) {
    if (g1_is_inf(p)) return q;
    if (g1_is_inf(q)) return p;
    return g1p14_to(g1_add14p(g1p14_from(p), g1p14_from(q)));
}
__device__ __forceinline__ G1Jac g1_add_mixed14(const G1Jac& p, const G1Aff& q) {
    if (g1_aff_is_inf(q)) return p;
    if (g1_is_inf(p)) return g1_from_aff(q);
    const Fq14 Z1 = fq14_from(p.z), X1 = fq14_from(p.x);
    const Fq14 Z1Z1 = fq14_sqr(Z1);
    Fq14 U2, t;
    fq14_mul2(fq14_from(q.x), Z1Z1, fq14_from(q.y), Z1, U2, t);
    const Fq14 H = fq14_sub4(U2, X1);
    if (fq14_maybe_zero(H)) return g1_add_mixed_c(p, q);
    const Fq14 Y1 = fq14_from(p.y);
    const Fq14 S2 = fq14_mul(t, Z1Z1);
    return g1p14_to(g1_tail14<1>(X1, Y1, H, fq14_sub4(S2, Y1), Z1, Z1));
}
__device__ __forceinline__ G1Jac g1_add_aff14(const G1Aff& p, const G1Aff& q) { return g1p14_to(g1_add_aff14p(p, q)); }
#define GM_G1_DEVICE_FQ14 1

// ---- XYZZ cells: (X, Y, ZZ, ZZZ) with x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2 -- what the sum-by-key tree keeps between its levels
// (g1.hip).  A general addition is 12 products + 2 squares (EFD add-2008-s) against the Jacobian 12 + 4, affine + affine 4 + 2
// (mmadd-2008-s) like the Jacobian one; the tree's results are group elements, so the representative does not matter, and the wire
// form is only produced once, at the end: (X ZZ, Y ZZZ, ZZ) is a Jacobian representative of the same point (Z' = ZZ).
// Invariant of a cell: limbs < 2^28; S <= 5.1 for X and Y, <= 1.1 for ZZ and ZZZ; infinity is ZZ = 0 limb for limb (Z = 0 converts
// to it exactly, and ZZ3 of an addition with P != 0 is a product of non-zero residues).  tests/test_fq14_model_cpu.py asserts the
// bounds below on the integer model.
struct G1X14 {
    Fq14 x, y, zz, zzz;
};
__device__ __forceinline__ G1X14 g1x_from_jac(const G1Jac& p) {
    G1X14 r;
    const Fq14 z = fq14_from(p.z);
    r.x = fq14_from(p.x); r.y = fq14_from(p.y);
    r.zz = fq14_sqr(z);
    r.zzz = fq14_mul(z, r.zz);
    return r;
}
__device__ __forceinline__ G1Jac g1x_to_jac(const G1X14& p) {
    if (fq14_limbs_zero(p.zz)) return g1_inf();
    Fq14 a, b;
    fq14_mul2(p.x, p.zz, p.y, p.zzz, a, b);
    G1Jac r;
    r.x = fq14_to(a); r.y = fq14_to(b); r.z = fq14_to(p.zz);
    return r;
}
// P = U2 - U1 + 4 q, R = S2 - S1 + 4 q (unnormalised: L < 2^29.6, S < 5.2).  AFF: ZZ3 = PP, ZZZ3 = PPP (both inputs affine)
template <bool AFF>
__device__ __forceinline__ G1X14 g1x_tail(const Fq14& U1, const Fq14& S1, const Fq14& P, const Fq14& R, const Fq14& zz12, const Fq14& zzz12) {
    Fq14 PP, RR, PPP, Q, RW, SP;
    fq14_sqr2(P, R, PP, RR);                                          // S 1.01
    fq14_mul2(P, PP, U1, PP, PPP, Q);                                 // 2^57.6: S 1.002
    const Fq14 T = fq14_norm(fq14_add(PPP, fq14_shl<1>(Q)));          // PPP + 2 Q: S 3.01
    G1X14 r;
    r.x = fq14_norm(fq14_sub4(RR, T));                                // S 5.01
    const Fq14 W = fq14_sub16(Q, r.x);                                // L < 2^29.6, S 17.0
    fq14_mul2(R, W, S1, PPP, RW, SP);                                 // 2^59.2: S 1.04
    r.y = fq14_norm(fq14_sub4(RW, SP));                               // S 5.04
    if (AFF) { r.zz = PP; r.zzz = PPP; }
    else fq14_mul2(zz12, PP, zzz12, PPP, r.zz, r.zzz);                // S 1.001
    return r;
}
__device__ __forceinline__ G1X14 g1x_add(const G1X14& p, const G1X14& q) {
    if (fq14_limbs_zero(p.zz)) return q;
    if (fq14_limbs_zero(q.zz)) return p;
    Fq14 U1, U2, S1, S2, zz12, zzz12;
    fq14_mul2(p.x, q.zz, q.x, p.zz, U1, U2);
    const Fq14 P = fq14_sub4(U2, U1);
    if (fq14_maybe_zero(P)) return g1x_from_jac(g1_add_c(g1x_to_jac(p), g1x_to_jac(q)));   // P = +-Q (or a rare false "maybe"): the exact path
    fq14_mul2(p.y, q.zzz, q.y, p.zzz, S1, S2);
    fq14_mul2(p.zz, q.zz, p.zzz, q.zzz, zz12, zzz12);
    return g1x_tail<false>(U1, S1, P, fq14_sub4(S2, S1), zz12, zzz12);
}
__device__ __forceinline__ G1X14 g1x_add_aff(const G1Aff& p, const G1Aff& q) {
    if (g1_aff_is_inf(p)) return g1x_from_jac(g1_from_aff(q));
    if (g1_aff_is_inf(q)) return g1x_from_jac(g1_from_aff(p));
    const Fq14 X1 = fq14_from(p.x), Y1 = fq14_from(p.y);
    const Fq14 P = fq14_sub4(fq14_from(q.x), X1);
    if (fq14_maybe_zero(P)) return g1x_from_jac(g1_add_aff_c(p, q));
    return g1x_tail<true>(X1, Y1, P, fq14_sub4(fq14_from(q.y), Y1), X1, X1);
}
#endif

GM_HD G1Jac g1_add(const G1Jac& p, const G1Jac& q) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GM_G1_DEVICE_FQ14)
    return g1_add14(p, q);
#else
    return g1_add_c(p, q);
#endif
}
GM_HD G1Jac g1_add_mixed(const G1Jac& p, const G1Aff& q) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GM_G1_DEVICE_FQ14)
    return g1_add_mixed14(p, q);
#else
    return g1_add_mixed_c(p, q);
#endif
}
GM_HD G1Jac g1_add_aff(const G1Aff& p, const G1Aff& q) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GM_G1_DEVICE_FQ14)
    return g1_add_aff14(p, q);
#else
    return g1_add_aff_c(p, q);
#endif
}

// into_affine: one field inversion
GM_HD G1Aff g1_to_aff(const G1Jac& p) {
    G1Aff r;
    if (g1_is_inf(p)) { r.x = fq_zero(); r.y = fq_zero(); return r; }
    const Fq zi = fq_inv(p.z), zi2 = fq_sqr(zi);
    r.x = fq_mul(p.x, zi2);
    r.y = fq_mul(p.y, fq_mul(zi2, zi));
    return r;
}

GM_HD bool g1_aff_on_curve(const G1Aff& p) {
    if (g1_aff_is_inf(p)) return true;
    Fq four = fq_dbl(fq_dbl(fq_one()));
    return fq_eq(fq_sqr(p.y), fq_add(fq_mul(fq_sqr(p.x), p.x), four));
}

// Prime-order subgroup membership of an on-curve point, host only: ark-bls12-381 0.4.0 `is_in_correct_subgroup_assuming_on_curve`
// (g1.rs; un-vendored dependency; reached through G1Affine::deserialize_compressed with Validate::Yes in the reference's
// read_points, cleanup/proof_transcript.rs:59-69) = Section 6 of eprint 2021/1130:  phi(P) == -[x^2] P  with phi(x, y) = (beta x, y),
// x = 0xd201000000010000 (|BLS parameter|), beta the cube root of unity below; early out: [x]P == P != O is outside the subgroup.
inline G1Jac g1_mul_u64_host(const G1Aff& p, uint64_t k) {
    G1Jac acc = g1_inf();
    for (int b = 63; b >= 0; b--) {
        acc = g1_dbl(acc);
        if ((k >> b) & 1) acc = g1_add_mixed(acc, p);
    }
    return acc;
}
inline bool g1_aff_in_subgroup_host(const G1Aff& p) {
    if (g1_aff_is_inf(p)) return true;
    static const Fq beta = [] {
        Fq b;
        const uint32_t c[12] = {0xfffefffeu, 0x2e01ffffu, 0x620a0002u, 0xde17d813u, 0xe6f89688u, 0xddb3a93bu,
                                0x6a0f77eau, 0xba69c607u, 0xdf76ce51u, 0x5f19672fu, 0x00000000u, 0x00000000u};
        for (int i = 0; i < 12; i++) b.l[i] = c[i];
        return fq_to_mont(b);
    }();
    const uint64_t X = 0xd201000000010000ull;
    const G1Aff xp = g1_to_aff(g1_mul_u64_host(p, X));
    if (fq_eq(xp.x, p.x) && fq_eq(xp.y, p.y)) return false;
    const G1Aff x2p = g1_to_aff(g1_mul_u64_host(xp, X));
    if (g1_aff_is_inf(x2p)) return false;  // phi(P) is a finite point
    return fq_eq(x2p.x, fq_mul(beta, p.x)) && fq_eq(fq_neg(x2p.y), p.y);
}

#if defined(__HIPCC__)
__device__ __forceinline__ G1Aff g1_aff_load(const G1Aff* p) {
    G1Aff r;
    r.x = fq_load(&p->x); r.y = fq_load(&p->y);
    return r;
}
__device__ __forceinline__ void g1_aff_store(G1Aff* p, const G1Aff& v) {
    fq_store(&p->x, v.x); fq_store(&p->y, v.y);
}
__device__ __forceinline__ G1Jac g1_load(const G1Jac* p) {
    G1Jac r;
    r.x = fq_load(&p->x); r.y = fq_load(&p->y); r.z = fq_load(&p->z);
    return r;
}
__device__ __forceinline__ void g1_store(G1Jac* p, const G1Jac& v) {
    fq_store(&p->x, v.x); fq_store(&p->y, v.y); fq_store(&p->z, v.z);
}
#endif

}  // namespace gm
