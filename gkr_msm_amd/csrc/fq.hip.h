// BLS12-381 base field Fq (381 bits) for gfx950, 12 x 32-bit limbs.
//
// In-memory form is the reference's: ark-ff 0.4.2 `Fp<MontBackend<FqConfig,6>>` = BigInt([u64;6]), Montgomery with
// R = 2^384, little-endian, 48 bytes per element; a 6 x u64 LE element is bit-identical to 12 x u32 LE.
// Used by the G1 side of the path: the outer-bucket accumulations of PushForwardState::new
// (/root/reference/src/cleanup/protocols/pushforward/pushforward.rs:395-456), msm_nonaffine.rs, binary_msm.rs,
// KzgProvingKey::commit (commitments/kzg.rs:123-126).
//
// Same multiplier shape as Fr (fr.hip.h): CIOS over 32-bit limbs, every 32x32 product one v_mad_u64_u32 that also absorbs
// the matching accumulator limb, high halves folded with one add-with-carry chain per row.  q is not 1 mod 2^32, so
// the quotient digit costs one v_mul_lo_u32 (m = t0 * (-q^-1 mod 2^32)).  q < 2^381: sums of two elements and the CIOS
// accumulator never need a 13th/14th limb.
#pragma once
#include "fr.hip.h"

namespace gm {

struct Fq {
    uint32_t l[12];
};

GM_HD uint32_t fq_p(int i) {
    switch (i) {
        case 0: return 0xffffaaabu; case 1: return 0xb9feffffu; case 2: return 0xb153ffffu; case 3: return 0x1eabfffeu;
        case 4: return 0xf6b0f624u; case 5: return 0x6730d2a0u; case 6: return 0xf38512bfu; case 7: return 0x64774b84u;
        case 8: return 0x434bacd7u; case 9: return 0x4b1ba7b6u; case 10: return 0x397fe69au; default: return 0x1a0111eau;
    }
}
#define GM_FQ_INV32 0xfffcfffdu  // -q^-1 mod 2^32

GM_HD Fq fq_zero() {
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = 0;
    return r;
}

// R mod q (Montgomery form of 1)
GM_HD Fq fq_one() {
    Fq r;
    r.l[0] = 0x0002fffdu; r.l[1] = 0x76090000u; r.l[2] = 0xc40c0002u; r.l[3] = 0xebf4000bu;
    r.l[4] = 0x53c758bau; r.l[5] = 0x5f489857u; r.l[6] = 0x70525745u; r.l[7] = 0x77ce5853u;
    r.l[8] = 0xa256ec6du; r.l[9] = 0x5c071a97u; r.l[10] = 0xfa80e493u; r.l[11] = 0x15f65ec3u;
    return r;
}

GM_HD Fq fq_r2() {
    Fq r;
    r.l[0] = 0x1c341746u; r.l[1] = 0xf4df1f34u; r.l[2] = 0x09d104f1u; r.l[3] = 0x0a76e6a6u;
    r.l[4] = 0x4c95b6d5u; r.l[5] = 0x8de5476cu; r.l[6] = 0x939d83c0u; r.l[7] = 0x67eb88a9u;
    r.l[8] = 0xb519952du; r.l[9] = 0x9a793e85u; r.l[10] = 0x92cae3aau; r.l[11] = 0x11988fe5u;
    return r;
}

GM_HD bool fq_is_zero(const Fq& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) o |= a.l[i];
    return o == 0;
}

GM_HD bool fq_eq(const Fq& a, const Fq& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

GM_HD Fq fq_reduce_once(const Fq& a) {
    Fq t;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint64_t d = (uint64_t)a.l[i] - fq_p(i) - borrow;
        t.l[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = borrow ? a.l[i] : t.l[i];
    return r;
}

GM_HD Fq fq_add(const Fq& a, const Fq& b) {
    Fq s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        s.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return fq_reduce_once(s);  // a, b < q < 2^381: no carry out of limb 11
}

GM_HD Fq fq_dbl(const Fq& a) { return fq_add(a, a); }

GM_HD Fq fq_sub(const Fq& a, const Fq& b) {
    Fq d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
        d.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    const uint32_t mask = borrow ? 0xffffffffu : 0u;
    uint64_t c = 0;
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        c += (uint64_t)d.l[i] + (fq_p(i) & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

GM_HD Fq fq_neg(const Fq& a) {
    if (fq_is_zero(a)) return a;
    Fq r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint64_t t = (uint64_t)fq_p(i) - a.l[i] - borrow;
        r.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    return r;
}

// Montgomery product a*b*R^-1 mod q, 32-bit-limb CIOS (device path; also compiled for the host as a cross-check).
GM_HD Fq fq_mul_c(const Fq& a, const Fq& b) {
    uint32_t t[13];
#pragma unroll
    for (int i = 0; i < 13; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint32_t bi = b.l[i];
        uint32_t lo[12], hi[12];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const uint64_t p = (uint64_t)a.l[j] * bi + t[j];
            lo[j] = (uint32_t)p;
            hi[j] = (uint32_t)(p >> 32);
        }
        uint32_t c = 0;
        t[0] = lo[0];
#pragma unroll
        for (int j = 1; j < 12; j++) t[j] = fr_addc(lo[j], hi[j - 1], c, &c);
        t[12] = fr_addc(t[12], hi[11], c, &c);
        const uint32_t m = t[0] * GM_FQ_INV32;
        uint32_t ql[12], qh[12];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const uint64_t p = (uint64_t)m * fq_p(j) + t[j];
            ql[j] = (uint32_t)p;  // ql[0] == 0 by the choice of m
            qh[j] = (uint32_t)(p >> 32);
        }
        c = 0;
#pragma unroll
        for (int j = 1; j < 12; j++) t[j - 1] = fr_addc(ql[j], qh[j - 1], c, &c);
        t[11] = fr_addc(t[12], qh[11], c, &c);
        t[12] = c;
    }
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = t[i];
    return fq_reduce_once(r);  // < 2q < 2^384
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host path: 6 x 64-bit CIOS with 128-bit products.
inline Fq fq_mul_host64(const Fq& a, const Fq& b) {
    typedef unsigned __int128 u128;
    static const uint64_t Qm[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                   0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
    uint64_t x[6], y[6], t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
    }
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) {
            c += (u128)x[j] * y[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * 0x89f3fffcfffcfffdULL;  // -q^-1 mod 2^64
        c = ((u128)m * Qm[0] + t[0]) >> 64;
        for (int j = 1; j < 6; j++) {
            c += (u128)m * Qm[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
    }
    uint64_t d[6];
    u128 br = 0;
    for (int i = 0; i < 6; i++) {
        const u128 v = (u128)t[i] - Qm[i] - br;
        d[i] = (uint64_t)v;
        br = (v >> 64) & 1;
    }
    const bool ge = t[6] != 0 || br == 0;
    Fq r;
    for (int i = 0; i < 6; i++) {
        const uint64_t v = ge ? d[i] : t[i];
        r.l[2 * i] = (uint32_t)v;
        r.l[2 * i + 1] = (uint32_t)(v >> 32);
    }
    return r;
}
#endif

GM_HD Fq fq_mul(const Fq& a, const Fq& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fq_mul_c(a, b);
#else
    return fq_mul_host64(a, b);
#endif
}

GM_HD Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }

GM_HD Fq fq_to_mont(const Fq& a) { return fq_mul(a, fq_r2()); }

GM_HD Fq fq_from_mont(const Fq& a) {
    Fq one = fq_zero();
    one.l[0] = 1;
    return fq_mul(a, one);
}

// a^(q-2)
GM_HD Fq fq_inv(const Fq& a) {
    Fq acc = fq_one();
    for (int i = 11; i >= 0; i--) {
        const uint32_t e = (i == 0) ? fq_p(0) - 2u : fq_p(i);  // q - 2: the low limb 0xffffaaab has no borrow
        for (int bit = 31; bit >= 0; bit--) {
            acc = fq_sqr(acc);
            if ((e >> bit) & 1) acc = fq_mul(acc, a);
        }
    }
    return acc;
}

#if defined(__HIPCC__)
__device__ __forceinline__ Fq fq_load(const Fq* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    Fq r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    return r;
}

__device__ __forceinline__ void fq_store(Fq* p, const Fq& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}
#endif

}  // namespace gm
