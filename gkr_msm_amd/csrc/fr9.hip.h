// BLS12-381 Fr in registers as 9 limbs of 29 bits ("Fr9"): the form the multiplication-bound kernels compute in.
//
// Why.  The 8 x 32-bit Montgomery product (fr.hip.h) is 120 v_mad_u64_u32 + 182 carry / select instructions: every 32 x 32
// product fills its 64-bit accumulator, so every one of them is followed by carry handling, and every product ends in a
// conditional subtraction.  With 29-bit limbs a 64-bit accumulator absorbs a whole column of the schoolbook product
// (18 terms < 2^60 each) with no carry at all: product scanning, one v_mad_u64_u32 per term, one shift + mask per column.
// The modulus keeps its gifts in this radix: p = 1 (mod 2^29), so the Montgomery quotient digit is m = -acc (mod 2^29), no
// multiplication, and m * p_0 is an addition.  R' = 2^261 leaves 6 bits of head room over p (255 bits): a product of
// operands a, b with (a / p)(b / p) <= 64 comes out below 2p, so chains of products, sums and differences need NO conditional
// subtraction until a value is stored.  Sums are limb-wise adds (9 instructions, no carry), differences add a limb-wise
// multiple of p first.
//
// Memory format is unchanged (canonical Montgomery form with R = 2^256, 8 x u32: what the reference's field elements are):
//   fr9_load :  X = x 2^256 mod p  ->  the limbs of 32 X = x 2^261 (mod p) as an integer below 2^260   (shifts only)
//   fr9_store:  Y  ->  (Y + m p) / 32 with m = -Y mod 32, one conditional subtraction, repacked          (Y < 32 p)
// Field values are identical to the 8 x 32 path's: same canonical bits in memory (scripts/ubench/fr9_mul_test.hip compares
// 2^20 random products and the edge cases against fr_mul; the kernels that use this form are pinned by the same oracle tests).
//
// Bounds are static, per formula, and written next to every use:  L = bound on every limb (the top limb included), S = bound on
// value / p.  2^261 / p = 70.66:
//   fr9_mul(a, b):  needs 9 L_a L_b + 9 2^58 + 2^36 < 2^64, i.e. L_a L_b <= 2^60.6;  gives limbs < 2^29, top limb < S 2^22.86,
//                   S = S_a S_b / 70.66 + 1
//   fr9_add: L, S add.   fr9_sub8 / fr9_sub2_32: L + 2^30.6, S + 8 / 32.   fr9_norm: limbs < 2^29 again, same value.
//   fr9_load: L = 2^29, S = 32 (32 X is not reduced).   fr9_store: needs S <= 33.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "fr.hip.h"

namespace gm {

struct Fr9 {
    uint32_t l[9];
};

static constexpr uint32_t M29 = (1u << 29) - 1;

__device__ __forceinline__ constexpr uint32_t fr9_p(int i) {
    return i == 0 ? 0x1u : i == 1 ? 0x1ffffff8u : i == 2 ? 0x1f96ffbfu : i == 3 ? 0x1b4805ffu : i == 4 ? 0x1d80553bu
         : i == 5 ? 0x0c0404d0u : i == 6 ? 0x1520cce7u : i == 7 ? 0x0a6533afu : 0x0073eda7u;
}

// canonical 8 x 32 Montgomery (R = 2^256) -> Fr9 (R' = 2^261): the limbs of X << 5.  Limbs < 2^29, value < 32 p.
__device__ __forceinline__ Fr9 fr9_from(const Fr& x) {
    Fr9 r;
    r.l[0] = (x.l[0] << 5) & M29;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        const int s = 29 * i - 5, w = s >> 5, off = s & 31;
        const uint32_t lo = x.l[w], hi = (w + 1 < 8) ? x.l[w + 1] : 0u;
        r.l[i] = (off ? __builtin_amdgcn_alignbit(hi, lo, off) : lo) & M29;
    }
    return r;
}

__device__ __forceinline__ Fr9 fr9_load(const Fr* p) { return fr9_from(fr_load(p)); }

// carry-propagate: limbs < 2^29 except the top one, which keeps what is left (value unchanged; limbs in < 2^32)
__device__ __forceinline__ Fr9 fr9_norm(const Fr9& a) {
    Fr9 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t t = a.l[i] + c;   // < 2^32 as long as limbs < 2^32 - 8
        r.l[i] = t & M29;
        c = t >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

// Fr9 (any limbs < 2^31, value < 32 p) -> canonical 8 x 32 Montgomery (R = 2^256)
__device__ __forceinline__ Fr fr9_to(const Fr9& y) {
    // Z = Y + m p, m = -Y mod 32 (p = 1 mod 32): divisible by 32; Z < 64 p
    const uint32_t m = (0u - y.l[0]) & 31u;
    uint32_t z[9];
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc += (uint64_t)m * fr9_p(i) + y.l[i];
        z[i] = (i < 8) ? ((uint32_t)acc & M29) : (uint32_t)acc;
        acc >>= 29;
    }
    // W = Z >> 5 < 2 p, repacked into 32-bit words: word j = bits [32 j + 5, 32 j + 37) of Z
    Fr w;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int s = 32 * j + 5, i0 = s / 29, o0 = s % 29;   // bit s of Z = bit o0 of limb i0
        uint32_t v = z[i0] >> o0;                            // 29 - o0 bits
        int have = 29 - o0;
        if (i0 + 1 < 9) { v |= z[i0 + 1] << have; have += 29; }
        if (have < 32 && i0 + 2 < 9) v |= z[i0 + 2] << have;
        w.l[j] = v;
    }
    return fr_reduce_once(w);
}

__device__ __forceinline__ void fr9_store(Fr* p, const Fr9& v) { fr_store(p, fr9_to(v)); }

// acc += a * b as ONE v_mad_u64_u32 (kept out of the compiler's hands: left alone it splits every column into two chains and
// merges them with 64-bit adds, 24 extra two-pass instructions per product).  The carry-out operand is dead: column sums stay
// below 2^64 by the limb bounds.
__device__ __forceinline__ void fr9_mad(uint64_t& acc, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void fr9_mad_k(uint64_t& acc, uint32_t a, uint32_t k_sgpr) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k_sgpr) : "vcc");
}

// Montgomery product a b 2^-261 (mod p), product scanning: 153 multiply-adds, 9 quotient digits at 3 instructions each,
// 17 shifts, 8 masks.  Limbs of a, b: 9 * max(a_i) * max(b_j) + 2^62 < 2^64 (e.g. both < 2^30, or < 2^31.5 and < 2^29).
// Result limbs < 2^29 (top limb: what is left), value < a b / 2^261 + p.
#ifndef GM_FR9_NO_ASM_BLOCK
// The same product as ONE asm block (fr9_mul_asm.inc, generated by scripts/gen/gen_fr9_mul_asm.py; the C formulation below is kept
// for reference and A/B: -DGM_FR9_NO_ASM_BLOCK).  The compiler pads every single-instruction asm statement with `s_nop 0`, and 170 of
// those per product cost 10-16 % of the multiply-add rate at 2-8 waves per SIMD, 40 % at one (scripts/ubench/mad_nop_test.hip).
__device__ __forceinline__ Fr9 fr9_mul(const Fr9& a, const Fr9& b) {
    Fr9 r;
    asm(
#include "fr9_mul_asm.inc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(r.l[8])
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]),
          "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(b.l[8])
        : "vcc", "v2", "v3", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59");
    return r;
}
__device__ __forceinline__ void fr9_mul2(const Fr9& a, const Fr9& b, const Fr9& c, const Fr9& d, Fr9& r, Fr9& q) {
    r = fr9_mul(a, b);   // (a dependent chain of multiply-adds issues as fast as two interleaved ones once the padding is gone)
    q = fr9_mul(c, d);
}
__device__ __forceinline__ Fr9 fr9_sqr(const Fr9& a) {
    Fr9 r;
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
    asm(
#include "fr9_sqr_asm.inc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(r.l[8]),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8])
        : "vcc", "v2", "v3", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59");
    (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)t5; (void)t6; (void)t7;
    return r;
}
#else
__device__ __forceinline__ Fr9 fr9_mul(const Fr9& a, const Fr9& b) {
    uint32_t m[9];
    Fr9 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) fr9_mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int j = 0; j < k; j++) fr9_mad_k(acc, m[j], fr9_p(k - j));
        m[k] = (0u - (uint32_t)acc) & M29;
        fr9_mad(acc, m[k], 1u);   // m[k] * p_0, p_0 = 1: the low 29 bits are now zero
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) fr9_mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int j = k - 8; j < 9; j++) fr9_mad_k(acc, m[j], fr9_p(k - j));
        r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// Two independent products with their instruction streams interleaved: a v_mad_u64_u32 that consumes the result of the one
// right before it costs a wait state (the compiler pads with s_nop); alternating the two chains removes every one of them.
__device__ __forceinline__ void fr9_mul2(const Fr9& a, const Fr9& b, const Fr9& c, const Fr9& d, Fr9& r, Fr9& q) {
    uint32_t m[9], n[9];
    uint64_t acc = 0, bcc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { fr9_mad(acc, a.l[i], b.l[k - i]); fr9_mad(bcc, c.l[i], d.l[k - i]); }
#pragma unroll
        for (int j = 0; j < k; j++) { fr9_mad_k(acc, m[j], fr9_p(k - j)); fr9_mad_k(bcc, n[j], fr9_p(k - j)); }
        m[k] = (0u - (uint32_t)acc) & M29;
        n[k] = (0u - (uint32_t)bcc) & M29;
        fr9_mad(acc, m[k], 1u);
        fr9_mad(bcc, n[k], 1u);
        acc >>= 29;
        bcc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { fr9_mad(acc, a.l[i], b.l[k - i]); fr9_mad(bcc, c.l[i], d.l[k - i]); }
#pragma unroll
        for (int j = k - 8; j < 9; j++) { fr9_mad_k(acc, m[j], fr9_p(k - j)); fr9_mad_k(bcc, n[j], fr9_p(k - j)); }
        r.l[k - 9] = (uint32_t)acc & M29;
        q.l[k - 9] = (uint32_t)bcc & M29;
        acc >>= 29;
        bcc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    q.l[8] = (uint32_t)bcc;
}

// a^2 2^-261 (mod p): the symmetric terms once, against the doubled limbs (126 multiply-adds).  Limbs of a < 2^29.5.
__device__ __forceinline__ Fr9 fr9_sqr(const Fr9& a) {
    uint32_t m[9], a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
    Fr9 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = (k < 9 ? 0 : k - 8); 2 * i < k; i++) fr9_mad(acc, a2[i], a.l[k - i]);
        if ((k & 1) == 0) fr9_mad(acc, a.l[k >> 1], a.l[k >> 1]);
        if (k < 9) {
#pragma unroll
            for (int j = 0; j < k; j++) fr9_mad_k(acc, m[j], fr9_p(k - j));
            m[k] = (0u - (uint32_t)acc) & M29;
            fr9_mad(acc, m[k], 1u);
        } else {
#pragma unroll
            for (int j = k - 8; j < 9; j++) fr9_mad_k(acc, m[j], fr9_p(k - j));
            r.l[k - 9] = (uint32_t)acc & M29;
        }
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

#endif  // GM_FR9_NO_ASM_BLOCK

// limb-wise sum (no carry): limb bounds add, values add
__device__ __forceinline__ Fr9 fr9_add(const Fr9& a, const Fr9& b) {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// 5 a (limb-wise): limb bounds and value times 5
__device__ __forceinline__ Fr9 fr9_mul5(const Fr9& a) {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (a.l[i] << 2) + a.l[i];
    return r;
}

// Differences: a - b + K p, limb-wise.  K p is written with every limb below the top one raised by 2^30 (and the next one lowered
// by 2 to pay for it), so a limb of the result cannot go negative as long as the limbs of b are <= 2^30 - 2 and its top limb is
// <= the top limb of the constant: K = 8: b < 2^25.8 2^232 (b < 7.9 p);  K = 32: b < 2^27.8 2^232 (b < 31.9 p).
// Result: limbs < limbs(a) + 2^30.6, value < a + K p.
__device__ __forceinline__ constexpr uint32_t fr9_bias8(int i) {
    return i == 0 ? 0x40000008u : i == 1 ? 0x5fffffbeu : i == 2 ? 0x5cb7fdfdu : i == 3 ? 0x5a402ffdu : i == 4 ? 0x4c02a9dcu
         : i == 5 ? 0x40202685u : i == 6 ? 0x49066739u : i == 7 ? 0x53299d7bu : 0x039f6d38u;
}
__device__ __forceinline__ constexpr uint32_t fr9_bias32(int i) {
    return i == 0 ? 0x40000020u : i == 1 ? 0x5ffffefeu : i == 2 ? 0x52dff7fdu : i == 3 ? 0x4900bffdu : i == 4 ? 0x500aa779u
         : i == 5 ? 0x40809a1bu : i == 6 ? 0x44199ceau : i == 7 ? 0x4ca675f3u : 0x0e7db4e8u;
}
__device__ __forceinline__ Fr9 fr9_sub8(const Fr9& a, const Fr9& b) {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + fr9_bias8(i) - b.l[i];
    return r;
}
// a - b + 64 p: b may be a shifted load (32 X, anything below 63.9 p)
__device__ __forceinline__ constexpr uint32_t fr9_bias64(int i) {
    return i == 0 ? 0x40000040u : i == 1 ? 0x5ffffdfeu : i == 2 ? 0x45bfeffdu : i == 3 ? 0x52017ffdu : i == 4 ? 0x40154ef4u
         : i == 5 ? 0x41013439u : i == 6 ? 0x483339d6u : i == 7 ? 0x594cebe8u : 0x1cfb69d2u;
}
__device__ __forceinline__ Fr9 fr9_sub64(const Fr9& a, const Fr9& b) {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + fr9_bias64(i) - b.l[i];
    return r;
}
// a - b - c + 32 p  (limbs of b + c <= 2^30 - 2: both normalised; top limbs: b + c < 31.9 p)
__device__ __forceinline__ Fr9 fr9_sub2_32(const Fr9& a, const Fr9& b, const Fr9& c) {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + fr9_bias32(i) - b.l[i] - c.l[i];
    return r;
}

// ---- the unshifted view: the limbs of X itself (value x 2^256 mod p, "domain 256").  A product of a domain-d value and a
// domain-e value is a domain-(d + e - 261) value, so formulas whose terms are homogeneous in their inputs can work on raw loads
// (S = 1, no conversion at all) and fix the domain once, with a constant, at the very end.  L = 2^29, S = 1.
__device__ __forceinline__ Fr9 fr9_from_raw(const Fr& x) {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int s = 29 * i, w = s >> 5, off = s & 31;
        const uint32_t lo = x.l[w], hi = (w + 1 < 8) ? x.l[w + 1] : 0u;
        r.l[i] = (off ? __builtin_amdgcn_alignbit(hi, lo, off) : lo) & M29;
    }
    return r;
}
__device__ __forceinline__ Fr9 fr9_load_raw(const Fr* p) { return fr9_from_raw(fr_load(p)); }
// normalised limbs (< 2^29, top limb what is left), value < 2 p, already in domain 256  ->  canonical 8 x 32
__device__ __forceinline__ Fr fr9_to_raw(const Fr9& z) {
    Fr w;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int s = 32 * j, i0 = s / 29, o0 = s % 29;
        uint32_t v = z.l[i0] >> o0;
        int have = 29 - o0;
        if (i0 + 1 < 9) { v |= z.l[i0 + 1] << have; have += 29; }
        if (have < 32 && i0 + 2 < 9) v |= z.l[i0 + 2] << have;
        w.l[j] = v;
    }
    return fr_reduce_once(w);
}
// constants as raw limbs
__device__ __forceinline__ Fr9 fr9_const(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6,
                                         uint32_t a7, uint32_t a8) {
    Fr9 r;
    r.l[0] = a0; r.l[1] = a1; r.l[2] = a2; r.l[3] = a3; r.l[4] = a4; r.l[5] = a5; r.l[6] = a6; r.l[7] = a7; r.l[8] = a8;
    return r;
}
// 2^256 mod p (the field's one in domain 256), 2^271 mod p, 2^276 mod p
__device__ __forceinline__ Fr9 fr9_one256() {
    return fr9_const(0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x001824b1u);
}
__device__ __forceinline__ Fr9 fr9_two271() {
    return fr9_const(0x1ffee558u, 0x0008d53fu, 0x0f2eaa00u, 0x0220139fu, 0x05e4224eu, 0x100eb2eau, 0x00c2a845u, 0x12a69488u, 0x0021b895u);
}
__device__ __forceinline__ Fr9 fr9_two266() {
    return fr9_const(0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u, 0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu);
}
// 2^251 (below p): the field's one in domain 251
__device__ __forceinline__ Fr9 fr9_one251() { return fr9_const(0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0x00080000u); }
__device__ __forceinline__ Fr9 fr9_two276() {
    return fr9_const(0x1fdcaaf7u, 0x011aa847u, 0x09864240u, 0x0e7a3defu, 0x13014aa7u, 0x15b231edu, 0x1a2dd48du, 0x1743bfd3u, 0x0023b7d0u);
}

__device__ __forceinline__ Fr9 fr9_zero() {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
// 2^261 mod p: the field's one
__device__ __forceinline__ Fr9 fr9_one() {
    Fr9 r;
    r.l[0] = 0x1fffffbau; r.l[1] = 0x0000022fu; r.l[2] = 0x1cb61180u; r.l[3] = 0x0a4e5c00u; r.l[4] = 0x0ee8b1a2u;
    r.l[5] = 0x16e6aedfu; r.l[6] = 0x1907f8bbu; r.l[7] = 0x0853ddf7u; r.l[8] = 0x004d043fu;
    return r;
}
// the Bandersnatch coefficient d (fr_coeff_d) in this form
__device__ __forceinline__ Fr9 fr9_coeff_d() {
    Fr9 r;
    r.l[0] = 0x1458e5f2u; r.l[1] = 0x1ced1bb7u; r.l[2] = 0x0c2440c6u; r.l[3] = 0x03a6574fu; r.l[4] = 0x06ebc6f2u;
    r.l[5] = 0x05d944c8u; r.l[6] = 0x185ecb02u; r.l[7] = 0x1cdb6c09u; r.l[8] = 0x006ed285u;
    return r;
}

}  // namespace gm
