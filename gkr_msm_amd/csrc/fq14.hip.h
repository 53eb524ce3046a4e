// BLS12-381 Fq in registers as 14 limbs of 28 bits ("Fq14"): the form the G1 additions multiply in on the device.
//
// Same idea as the 9 x 29 scalar field (fr9.hip.h).  The 12 x 32 Montgomery product (fq.hip.h) is 288 v_mad_u64_u32 plus as many
// add-with-carry instructions (every 32 x 32 product fills its accumulator) and a conditional subtraction; with 28-bit limbs a
// 64-bit accumulator takes a whole column of the schoolbook product (28 terms < 2^56.x) without a carry: product scanning,
// 392 multiply-adds, one multiplication, one shift and one mask per column (1.2x the 12 x 32 product's rate on MI355X).  R' = 2^392 leaves 11 bits over q (381 bits,
// 2^392 / q = 2520): products, sums and differences chain with NO conditional subtraction until a value is stored; a square pays
// the symmetric terms once (301 multiply-adds).
//
// Memory format is unchanged (canonical Montgomery form with R = 2^384, 12 x u32 = ark-ff's 6 x u64):
//   fq14_load :  X = x 2^384  ->  X 2^8 - k q (k from the top limb of X 2^8): the residue of x 2^392, below 1.1 q, limbs < 2^28
//   fq14_to   :  Y (< 256 q)  ->  (Y + m q) / 256 with m = -Y / q mod 256, one conditional subtraction, repacked
// so a formula evaluated in this form stores bit for bit what the 12 x 32 path stores (scripts/ubench/fq14_test.hip compares the
// three G1 additions on the device; tests/test_fq14_model_cpu.py is the integer model that asserts the bounds below).
//
// Bounds (L = every limb below the top one, S = value / q):
//   fq14_mul(a, b):  14 L_a L_b + 14 2^56 + 2^36 < 2^64, i.e. L_a L_b <= 2^60;  result limbs < 2^28, S = S_a S_b / 2520 + 1
//   fq14_sub4 / fq14_sub16(a, b): b normalised (limbs < 2^28) and below 3.99 q / 15.99 q;  result L_a + 2^29, S_a + 4 / 16
//   fq14_add / fq14_shl: limb-wise, bounds add / scale.   fq14_norm: limbs < 2^28 again, same value.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "fq.hip.h"
#include "fr9.hip.h"

namespace gm {

struct Fq14 {
    uint32_t l[14];
};

static constexpr uint32_t M28 = (1u << 28) - 1;
static constexpr uint32_t FQ14_QINV = 0x0ffcfffdu;   // -q^-1 mod 2^28

__device__ __forceinline__ constexpr uint32_t fq14_q(int i) {
    return i == 0 ? 0x0fffaaabu : i == 1 ? 0x0fefffffu : i == 2 ? 0x03ffffb9u : i == 3 ? 0x0fffeb15u : i == 4 ? 0x06241eabu
         : i == 5 ? 0x0a0f6b0fu : i == 6 ? 0x0f6730d2u : i == 7 ? 0x0f38512bu : i == 8 ? 0x04774b84u : i == 9 ? 0x04bacd76u
         : i == 10 ? 0x0ba7b643u : i == 11 ? 0x0e69a4b1u : i == 12 ? 0x01ea397fu : 0x0001a011u;
}
// 4 q and 16 q with every limb below the top one raised by 2^28 (the next one pays): dominates a normalised subtrahend limb-wise
__device__ __forceinline__ constexpr uint32_t fq14_bias4(int i) {
    return i == 0 ? 0x1ffeaaacu : i == 1 ? 0x1fbffffeu : i == 2 ? 0x1ffffee6u : i == 3 ? 0x1fffac53u : i == 4 ? 0x18907aaeu
         : i == 5 ? 0x183dac3cu : i == 6 ? 0x1d9cc349u : i == 7 ? 0x1ce144aeu : i == 8 ? 0x11dd2e12u : i == 9 ? 0x12eb35d8u
         : i == 10 ? 0x1e9ed90cu : i == 11 ? 0x19a692c5u : i == 12 ? 0x17a8e5feu : 0x00068043u;
}
__device__ __forceinline__ constexpr uint32_t fq14_bias16(int i) {
    return i == 0 ? 0x1ffaaab0u : i == 1 ? 0x1efffffeu : i == 2 ? 0x1ffffb9eu : i == 3 ? 0x1ffeb152u : i == 4 ? 0x1241eabeu
         : i == 5 ? 0x10f6b0f5u : i == 6 ? 0x16730d29u : i == 7 ? 0x138512beu : i == 8 ? 0x1774b84eu : i == 9 ? 0x1bacd763u
         : i == 10 ? 0x1a7b6433u : i == 11 ? 0x169a4b1au : i == 12 ? 0x1ea397fdu : 0x001a0110u;
}

// acc += (int32) a * (int32) b, 64-bit signed
__device__ __forceinline__ void fq14_mad_i(int64_t& acc, int32_t a, int32_t b_sgpr) {
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b_sgpr) : "vcc");
}

// canonical 12 x 32 Montgomery (R = 2^384) -> Fq14 (R' = 2^392): V = X << 8, minus k q with k = floor(top limb * 2^32 / (q >> 364 + 1)) / 2^32
// <= V / q: 0 <= V - k q < 1.1 q.  Limbs < 2^28.
__device__ __forceinline__ Fq14 fq14_from(const Fq& x) {
    uint32_t v[14];
    v[0] = (x.l[0] << 8) & M28;
#pragma unroll
    for (int i = 1; i < 14; i++) {
        const int s = 28 * i - 8, w = s >> 5, off = s & 31;
        const uint32_t lo = x.l[w], hi = (w + 1 < 12) ? x.l[w + 1] : 0u;
        v[i] = (off ? __builtin_amdgcn_alignbit(hi, lo, off) : lo) & M28;
    }
    const int32_t nk = -(int32_t)__umulhi(v[13], 40323u);
    Fq14 r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        fq14_mad_i(acc, nk, (int32_t)fq14_q(i));
        acc += v[i];
        r.l[i] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    return r;
}

// carry-propagate: limbs < 2^28 except the top one, which keeps what is left (value unchanged)
__device__ __forceinline__ Fq14 fq14_norm(const Fq14& a) {
    Fq14 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const uint32_t t = a.l[i] + c;
        r.l[i] = t & M28;
        c = t >> 28;
    }
    r.l[13] = a.l[13] + c;
    return r;
}

// Fq14 (limbs < 2^32, value < 256 q) -> canonical 12 x 32 Montgomery (R = 2^384)
__device__ __forceinline__ Fq fq14_to(const Fq14& y) {
    const uint32_t m = (y.l[0] * 253u) & 255u;   // Y + m q = 0 (mod 256): -q^-1 = 253 (mod 256)
    uint32_t z[14];
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        fr9_mad_k(acc, m, fq14_q(i));
        acc += y.l[i];
        z[i] = (i < 13) ? ((uint32_t)acc & M28) : (uint32_t)acc;
        acc >>= 28;
    }
    // W = Z >> 8 < 2 q, repacked into 32-bit words: word j = bits [32 j + 8, 32 j + 40) of Z
    Fq w;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int s = 32 * j + 8, i0 = s / 28, o0 = s % 28;
        uint32_t u = z[i0] >> o0;
        int have = 28 - o0;
        if (i0 + 1 < 14) { u |= z[i0 + 1] << have; have += 28; }
        if (have < 32 && i0 + 2 < 14) u |= z[i0 + 2] << have;
        w.l[j] = u;
    }
    return fq_reduce_once(w);
}

#ifndef GM_FQ14_NO_ASM_BLOCKS
// fq14_mul / fq14_sqr with the multiply-adds of a column in two asm blocks (generated: scripts/gen/gen_fq14_mul.py): the compiler pads
// every asm statement with `s_nop 0`, and at the two waves per SIMD of the G1 kernels 420 of those per product cost ~15 % of the
// multiply-add rate (scripts/ubench/mad_nop_test.hip).  The one-statement-per-instruction formulation below is the reference
// (-DGM_FQ14_NO_ASM_BLOCKS); without the padding a dependent chain issues as fast as two interleaved ones, so the "two at a time"
// helpers are plain pairs.
#include "fq14_mul_gen.inc"
__device__ __forceinline__ void fq14_mul2(const Fq14& a, const Fq14& b, const Fq14& c, const Fq14& d, Fq14& r, Fq14& q) {
    r = fq14_mul(a, b);
    q = fq14_mul(c, d);
}
__device__ __forceinline__ void fq14_sqr2(const Fq14& a, const Fq14& c, Fq14& r, Fq14& q) {
    r = fq14_sqr(a);
    q = fq14_sqr(c);
}
#else
// Montgomery product a b 2^-392 (mod q), product scanning.  Limbs: 14 max(a_i) max(b_j) + 2^60 < 2^64.
// Result limbs < 2^28 (top limb: what is left), value < a b / 2^392 + q.
__device__ __forceinline__ Fq14 fq14_mul(const Fq14& a, const Fq14& b) {
    uint32_t m[14];
    Fq14 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) fr9_mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int j = 0; j < k; j++) fr9_mad_k(acc, m[j], fq14_q(k - j));
        m[k] = ((uint32_t)acc * FQ14_QINV) & M28;
        fr9_mad_k(acc, m[k], fq14_q(0));
        acc >>= 28;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) fr9_mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int j = k - 13; j < 14; j++) fr9_mad_k(acc, m[j], fq14_q(k - j));
        r.l[k - 14] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    return r;
}

// a^2 2^-392 (mod q): the symmetric terms once, against the doubled limbs.  Limbs of a < 2^30.
__device__ __forceinline__ Fq14 fq14_sqr(const Fq14& a) {
    uint32_t m[14], a2[14];
#pragma unroll
    for (int i = 0; i < 14; i++) a2[i] = a.l[i] << 1;
    Fq14 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
#pragma unroll
        for (int i = (k < 14 ? 0 : k - 13); 2 * i < k; i++) fr9_mad(acc, a2[i], a.l[k - i]);
        if ((k & 1) == 0) fr9_mad(acc, a.l[k >> 1], a.l[k >> 1]);
        if (k < 14) {
#pragma unroll
            for (int j = 0; j < k; j++) fr9_mad_k(acc, m[j], fq14_q(k - j));
            m[k] = ((uint32_t)acc * FQ14_QINV) & M28;
            fr9_mad_k(acc, m[k], fq14_q(0));
        } else {
#pragma unroll
            for (int j = k - 13; j < 14; j++) fr9_mad_k(acc, m[j], fq14_q(k - j));
            r.l[k - 14] = (uint32_t)acc & M28;
        }
        acc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    return r;
}

// Two independent products / squares with their instruction streams interleaved (as fr9_mul2): two accumulator chains per wave,
// so the multiplier stays busy at the two waves per SIMD the G1 kernels run (their 14-limb operands leave no room for more).
// Measured (scripts/ubench/fq14_test.hip, MI355X): 66-69 G products/s against 57 G for the 12 x 32 form -- both forms are bound
// by v_mad_u64_u32 issue (392 against 288 + 288 add-with-carry), so the gain is 1.2x, not the 1.35x the cycle counts suggested.
__device__ __forceinline__ void fq14_mul2(const Fq14& a, const Fq14& b, const Fq14& c, const Fq14& d, Fq14& r, Fq14& q) {
    uint32_t m[14], n[14];
    uint64_t acc = 0, bcc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { fr9_mad(acc, a.l[i], b.l[k - i]); fr9_mad(bcc, c.l[i], d.l[k - i]); }
#pragma unroll
        for (int j = 0; j < k; j++) { fr9_mad_k(acc, m[j], fq14_q(k - j)); fr9_mad_k(bcc, n[j], fq14_q(k - j)); }
        m[k] = ((uint32_t)acc * FQ14_QINV) & M28;
        n[k] = ((uint32_t)bcc * FQ14_QINV) & M28;
        fr9_mad_k(acc, m[k], fq14_q(0));
        fr9_mad_k(bcc, n[k], fq14_q(0));
        acc >>= 28;
        bcc >>= 28;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) { fr9_mad(acc, a.l[i], b.l[k - i]); fr9_mad(bcc, c.l[i], d.l[k - i]); }
#pragma unroll
        for (int j = k - 13; j < 14; j++) { fr9_mad_k(acc, m[j], fq14_q(k - j)); fr9_mad_k(bcc, n[j], fq14_q(k - j)); }
        r.l[k - 14] = (uint32_t)acc & M28;
        q.l[k - 14] = (uint32_t)bcc & M28;
        acc >>= 28;
        bcc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    q.l[13] = (uint32_t)bcc;
}
__device__ __forceinline__ void fq14_sqr2(const Fq14& a, const Fq14& c, Fq14& r, Fq14& q) {
    uint32_t m[14], n[14], a2[14], c2[14];
#pragma unroll
    for (int i = 0; i < 14; i++) { a2[i] = a.l[i] << 1; c2[i] = c.l[i] << 1; }
    uint64_t acc = 0, bcc = 0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
#pragma unroll
        for (int i = (k < 14 ? 0 : k - 13); 2 * i < k; i++) { fr9_mad(acc, a2[i], a.l[k - i]); fr9_mad(bcc, c2[i], c.l[k - i]); }
        if ((k & 1) == 0) { fr9_mad(acc, a.l[k >> 1], a.l[k >> 1]); fr9_mad(bcc, c.l[k >> 1], c.l[k >> 1]); }
        if (k < 14) {
#pragma unroll
            for (int j = 0; j < k; j++) { fr9_mad_k(acc, m[j], fq14_q(k - j)); fr9_mad_k(bcc, n[j], fq14_q(k - j)); }
            m[k] = ((uint32_t)acc * FQ14_QINV) & M28;
            n[k] = ((uint32_t)bcc * FQ14_QINV) & M28;
            fr9_mad_k(acc, m[k], fq14_q(0));
            fr9_mad_k(bcc, n[k], fq14_q(0));
        } else {
#pragma unroll
            for (int j = k - 13; j < 14; j++) { fr9_mad_k(acc, m[j], fq14_q(k - j)); fr9_mad_k(bcc, n[j], fq14_q(k - j)); }
            r.l[k - 14] = (uint32_t)acc & M28;
            q.l[k - 14] = (uint32_t)bcc & M28;
        }
        acc >>= 28;
        bcc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    q.l[13] = (uint32_t)bcc;
}

#endif  // GM_FQ14_NO_ASM_BLOCKS

__device__ __forceinline__ Fq14 fq14_add(const Fq14& a, const Fq14& b) {
    Fq14 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <int N>
__device__ __forceinline__ Fq14 fq14_shl(const Fq14& a) {
    Fq14 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] << N;
    return r;
}
// a - b + 4 q / 16 q, limb-wise (b normalised, below 3.99 q / 15.99 q)
__device__ __forceinline__ Fq14 fq14_sub4(const Fq14& a, const Fq14& b) {
    Fq14 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + fq14_bias4(i) - b.l[i];
    return r;
}
__device__ __forceinline__ Fq14 fq14_sub16(const Fq14& a, const Fq14& b) {
    Fq14 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + fq14_bias16(i) - b.l[i];
    return r;
}

// H = a + 4 q - b (below 6 q): can it be 0 (mod q)?  Only as j q, j <= 5: compare the low 28 bits (limb 0 carries them alone).
// No false negative; a false positive once in 2^25.4 (the caller then takes the exact 12 x 32 path).
__device__ __forceinline__ bool fq14_maybe_zero(const Fq14& h) {
    bool z = false;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) z |= ((h.l[0] - j * fq14_q(0)) & M28) == 0;
    return z;
}

}  // namespace gm
