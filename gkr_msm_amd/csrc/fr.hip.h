// BLS12-381 scalar field Fr (= Bandersnatch base field) for gfx950, 8 x 32-bit limbs.
//
// In-memory form is the reference's: Montgomery (R = 2^256), little-endian, 32 bytes per element
// (ark-ff 0.4.2 `Fp<MontBackend<FrConfig,4>>` = BigInt([u64;4]); /root/reference/src/utils.rs:32-37
// spells out COEFF_D in exactly this form).  A 4xu64 LE element is bit-identical to 8xu32 LE, so
// device buffers are imported/exported without conversion.
//
// CDNA4 has no 64-bit integer multiplier; the natural machine word is v_mad_u64_u32
// (32x32+64 -> 64).  p = 1 (mod 2^32) gives -p^-1 = 0xffffffff, so the Montgomery quotient digit
// is just the negated low limb (no multiply), and p[1] = 0xffffffff turns one more product per
// reduction step into shifts/subtracts.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GM_HD __host__ __device__ __forceinline__
#else
#define GM_HD inline
#endif

namespace gm {

struct Fr {
    uint32_t l[8];
};

// modulus p, little-endian 32-bit limbs
#define GM_P0 0x00000001u
#define GM_P1 0xffffffffu
#define GM_P2 0xfffe5bfeu
#define GM_P3 0x53bda402u
#define GM_P4 0x09a1d805u
#define GM_P5 0x3339d808u
#define GM_P6 0x299d7d48u
#define GM_P7 0x73eda753u

GM_HD uint32_t fr_p(int i) {
    switch (i) {
        case 0: return GM_P0; case 1: return GM_P1; case 2: return GM_P2; case 3: return GM_P3;
        case 4: return GM_P4; case 5: return GM_P5; case 6: return GM_P6; default: return GM_P7;
    }
}

GM_HD Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
}

// R mod p  (Montgomery form of 1)
GM_HD Fr fr_one() {
    Fr r;
    r.l[0] = 0xfffffffeu; r.l[1] = 0x00000001u; r.l[2] = 0x00034802u; r.l[3] = 0x5884b7fau;
    r.l[4] = 0xecbc4ff5u; r.l[5] = 0x998c4fefu; r.l[6] = 0xacc5056fu; r.l[7] = 0x1824b159u;
    return r;
}

// Montgomery form of the Bandersnatch coefficient d  (KAT: /root/reference/src/utils.rs:35)
GM_HD Fr fr_coeff_d() {
    Fr r;
    r.l[0] = 0x47a2c730u; r.l[1] = 0xa8dced1bu; r.l[2] = 0xad3cccc7u; r.l[3] = 0x381c065au;
    r.l[4] = 0x188351f8u; r.l[5] = 0x53ff52e1u; r.l[6] = 0x990fe940u; r.l[7] = 0x362e8d63u;
    return r;
}

GM_HD bool fr_is_zero(const Fr& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i];
    return o == 0;
}

GM_HD bool fr_eq(const Fr& a, const Fr& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// r = a - p if a >= p else a      (a < 2p)
GM_HD Fr fr_reduce_once(const Fr& a) {
    Fr t;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a.l[i] - fr_p(i) - borrow;
        t.l[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : t.l[i];
    return r;
}

GM_HD Fr fr_add(const Fr& a, const Fr& b) {
    Fr s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        s.l[i] = (uint32_t)c;
        c >>= 32;
    }
    // a, b < p < 2^255  =>  no carry out of limb 7
    return fr_reduce_once(s);
}

GM_HD Fr fr_dbl(const Fr& a) { return fr_add(a, a); }

GM_HD Fr fr_sub(const Fr& a, const Fr& b) {
    Fr d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
        d.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    uint32_t mask = borrow ? 0xffffffffu : 0u;
    uint64_t c = 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)d.l[i] + (fr_p(i) & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

GM_HD Fr fr_neg(const Fr& a) {
    if (fr_is_zero(a)) return a;
    Fr r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)fr_p(i) - a.l[i] - borrow;
        r.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    return r;
}

GM_HD uint32_t fr_addc(uint32_t a, uint32_t b, uint32_t cin, uint32_t* cout) {
    return __builtin_addc(a, b, cin, cout);
}

// Montgomery product a*b*R^-1 mod p.  CIOS over 32-bit limbs; quotient digit m = -t0.
// Shape chosen for gfx950: every 32x32 product is one v_mad_u64_u32 that also absorbs the matching
// limb of t (a*b + t_j never overflows 64 bits); the high halves are then folded in with one 32-bit
// add-with-carry chain per row.  This keeps the operands of each mad in place (no 64-bit zero-extension
// shuffles), which is worth ~1.4x over the textbook running-carry loop with hipcc 7.2.
#if defined(__HIP_DEVICE_COMPILE__) && defined(GM_FR_MUL_ASM)
// Device path: hand-laid-out instruction stream with fixed VGPRs (scripts/gen/gen_fr_mul_asm.py explains the
// register plan).  Bit-identical to the C formulation below (scripts/ubench/fr_mul_asm_test.hip checks 2^20 random
// pairs + edge cases on the device against it, and the C formulation is what the host/oracle tests pin).
__device__ __forceinline__ Fr fr_mul_asm(const Fr& a, const Fr& b) {
    Fr r;
    asm volatile(
#include "fr_mul_asm.inc"
        : "={v16}"(r.l[0]), "={v18}"(r.l[1]), "={v20}"(r.l[2]), "={v22}"(r.l[3]), "={v24}"(r.l[4]), "={v26}"(r.l[5]),
          "={v28}"(r.l[6]), "={v30}"(r.l[7])
        : "{v0}"(a.l[0]), "{v1}"(a.l[1]), "{v2}"(a.l[2]), "{v3}"(a.l[3]), "{v4}"(a.l[4]), "{v5}"(a.l[5]), "{v6}"(a.l[6]),
          "{v7}"(a.l[7]), "{v8}"(b.l[0]), "{v9}"(b.l[1]), "{v10}"(b.l[2]), "{v11}"(b.l[3]), "{v12}"(b.l[4]),
          "{v13}"(b.l[5]), "{v14}"(b.l[6]), "{v15}"(b.l[7])
        : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "v17", "v19", "v21", "v23", "v25", "v27", "v29", "v31", "v32",
          "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
          "v49", "v50");
    return r;
}
#endif

GM_HD Fr fr_mul_c(const Fr& a, const Fr& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t bi = b.l[i];
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t p = (uint64_t)a.l[j] * bi + t[j];
            lo[j] = (uint32_t)p;
            hi[j] = (uint32_t)(p >> 32);
        }
        uint32_t c = 0;
        t[0] = lo[0];
#pragma unroll
        for (int j = 1; j < 8; j++) t[j] = fr_addc(lo[j], hi[j - 1], c, &c);
        // p < 2^255 and a,b < p keep the running value below 2p*2^32: no 10th limb is needed.
        t[8] = fr_addc(t[8], hi[7], c, &c);
        // reduction step: t = (t + m*p) / 2^32 with m = -t0  (p = 1 mod 2^32  =>  -p^-1 = 0xffffffff)
        const uint32_t m = 0u - t[0];
        uint32_t ql[8], qh[8];
        // limb 0: t0 + m*1 = 0 or 2^32; its carry joins limb 1: t1 + m*0xffffffff + carry fits 64 bits
        const uint64_t p1 = (uint64_t)m * GM_P1 + t[1] + ((t[0] != 0) ? 1u : 0u);
        ql[1] = (uint32_t)p1;
        qh[1] = (uint32_t)(p1 >> 32);
#pragma unroll
        for (int j = 2; j < 8; j++) {
            const uint64_t p = (uint64_t)m * fr_p(j) + t[j];
            ql[j] = (uint32_t)p;
            qh[j] = (uint32_t)(p >> 32);
        }
        c = 0;
        t[0] = ql[1];
#pragma unroll
        for (int j = 2; j < 8; j++) t[j - 1] = fr_addc(ql[j], qh[j - 1], c, &c);
        t[7] = fr_addc(t[8], qh[7], c, &c);
        t[8] = c;
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    // result < 2p and < 2^256 (t[8] == 0 since 2p < 2^256)
    return fr_reduce_once(r);
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host path (scalar glue of the drivers: recombination, interpolation, claims): 4 x 64-bit CIOS with 128-bit products.
inline Fr fr_mul_host64(const Fr& a, const Fr& b) {
    typedef unsigned __int128 u128;
    static const uint64_t Pm[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    uint64_t x[4], y[4], t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
    }
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)x[j] * y[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * 0xfffffffeffffffffULL;  // -p^-1 mod 2^64
        c = ((u128)m * Pm[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * Pm[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    // conditional subtraction
    uint64_t d[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 v = (u128)t[i] - Pm[i] - br;
        d[i] = (uint64_t)v;
        br = (v >> 64) & 1;
    }
    const bool ge = t[4] != 0 || br == 0;
    Fr r;
    for (int i = 0; i < 4; i++) {
        const uint64_t v = ge ? d[i] : t[i];
        r.l[2 * i] = (uint32_t)v;
        r.l[2 * i + 1] = (uint32_t)(v >> 32);
    }
    return r;
}
#endif

GM_HD Fr fr_mul(const Fr& a, const Fr& b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GM_FR_MUL_ASM)
    return fr_mul_asm(a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
    return fr_mul_c(a, b);
#else
    return fr_mul_host64(a, b);
#endif
}

GM_HD Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }

// x * R  (into Montgomery form): multiply by R^2
GM_HD Fr fr_r2() {
    Fr r;
    r.l[0] = 0xf3f29c6du; r.l[1] = 0xc999e990u; r.l[2] = 0x87925c23u; r.l[3] = 0x2b6cedcbu;
    r.l[4] = 0x7254398fu; r.l[5] = 0x05d31496u; r.l[6] = 0x9f59ff11u; r.l[7] = 0x0748d9d9u;
    return r;
}

GM_HD Fr fr_to_mont(const Fr& a) { return fr_mul(a, fr_r2()); }

GM_HD Fr fr_from_mont(const Fr& a) {
    Fr one = fr_zero();
    one.l[0] = 1;
    return fr_mul(a, one);
}

GM_HD Fr fr_from_u64(uint64_t v) {
    Fr a = fr_zero();
    a.l[0] = (uint32_t)v;
    a.l[1] = (uint32_t)(v >> 32);
    return fr_to_mont(a);
}

// a^(p-2)
GM_HD Fr fr_inv(const Fr& a) {
    // p - 2, little-endian 32-bit limbs
    const uint32_t e[8] = {0xffffffffu, 0xfffffffeu, GM_P2, GM_P3, GM_P4, GM_P5, GM_P6, GM_P7};
    Fr acc = fr_one();
    for (int i = 7; i >= 0; i--) {
        for (int bit = 31; bit >= 0; bit--) {
            acc = fr_sqr(acc);
            if ((e[i] >> bit) & 1) acc = fr_mul(acc, a);
        }
    }
    return acc;
}

// -5x  (/root/reference/src/utils.rs:40-43: two doublings, one add, one negation)
GM_HD Fr fr_mul_by_a(const Fr& x) {
    Fr t = fr_dbl(fr_dbl(x));
    return fr_neg(fr_add(t, x));
}

GM_HD Fr fr_mul_by_d(const Fr& x) { return fr_mul(x, fr_coeff_d()); }

// ---------------------------------------------------------------- memory (32-byte AoS elements)
#if defined(__HIPCC__)
__device__ __forceinline__ Fr fr_load(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    Fr r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
    r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}

__device__ __forceinline__ void fr_store(Fr* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
#endif

}  // namespace gm
