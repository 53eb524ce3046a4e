// Sumcheck objects on the device: the `Sumcheckable::{unipoly, bind, final_evals}` seam
// (/root/reference/src/cleanup/protocols/sumchecks/vecvec_eq.rs:218-225) for
//   DenseDeg2SumcheckObjectSO       dense_eq.rs:61-173          (degree-2 layer function, eq factored out)
//   VecVecDeg2SumcheckObjectSO      vecvec_eq.rs:72-398         (same over ragged rows, then handover to dense)
//   DenseSumcheckObjectSO           sumcheck.rs:237-347         (generic degree D; with EqWrapper/GammaWrapper :706-829)
//
// What runs where.  Per round the device does the two data-parallel passes: the round sums (every pair
// of every column is read once, the layer function is evaluated at "1" and "2", weighted by eq and the
// gamma powers and reduced to two field elements) and the fold.  The host only does the O(1) scalar
// tail (pads, multiplier, from12 / interpolation, claim update) with the same Fr code.
//
// Values.  The reference rewrites polynomials into "21 form" (make_21: p[2i] <- 2 p[2i+1] - p[2i]) and binds
// with bind_21.  Both are exact field identities of the plain pair (p0, p1): value-at-2 = 2 p1 - p0 and
// bind_21 = p1 + (t-1)(p2 - p1) = p0 + t (p1 - p0).  The device keeps plain form, computes value-at-2 in
// registers and folds with the plain formula: every round polynomial, folded polynomial and final evaluation
// is the same canonical field element as in the reference (sums in a field do not depend on their order).
#include <atomic>
#include <immintrin.h>
#include <chrono>
#include <condition_variable>

#include "internal.hpp"
#include "ragged.hip.h"
#include "fr9.hip.h"
#include "vecvec.hpp"

namespace gm {

// ------------------------------------------------------------------------------------------ host math
Fr eq_bind_factor(const Fr& q, const Fr& t) {
    // 1 - q - t + 2 q t   (dense_eq.rs:100, vecvec.rs:122)
    return fr_add(fr_sub(fr_sub(fr_one(), q), t), fr_dbl(fr_mul(q, t)));
}

// Lagrange basis on the nodes 0..n-1 in coefficient form: coeffs[k] = sum_i M[i][k] * evals[i].
// The nodes are fixed, so the matrix (and its n field inversions) is computed once per n.
static const std::vector<std::vector<Fr>>& lagrange_matrix(int n) {
    static thread_local std::vector<std::vector<std::vector<Fr>>> cache(9);
    if (n < 1 || n > 8) n = 8;
    std::vector<std::vector<Fr>>& M = cache[n];
    if (!M.empty()) return M;
    M.assign(n, std::vector<Fr>(n, fr_zero()));
    for (int i = 0; i < n; i++) {
        std::vector<Fr> num(1, fr_one());
        Fr den = fr_one();
        for (int j = 0; j < n; j++) {
            if (j == i) continue;
            std::vector<Fr> nx(num.size() + 1, fr_zero());  // num *= (x - j)
            const Fr fj = fr_from_u64((uint64_t)j);
            for (size_t k = 0; k < num.size(); k++) {
                nx[k + 1] = fr_add(nx[k + 1], num[k]);
                nx[k] = fr_sub(nx[k], fr_mul(fj, num[k]));
            }
            num.swap(nx);
            const Fr d = (i > j) ? fr_from_u64((uint64_t)(i - j)) : fr_neg(fr_from_u64((uint64_t)(j - i)));
            den = fr_mul(den, d);
        }
        const Fr di = fr_inv(den);
        for (int k = 0; k < n; k++) M[i][k] = fr_mul(num[k], di);
    }
    return M;
}

// inverses of 2 and 6 (Montgomery), once per thread
static const Fr& inv_small(int which) {
    static thread_local Fr v[2];
    static thread_local bool have = false;
    if (!have) { v[0] = fr_inv(fr_from_u64(2)); v[1] = fr_inv(fr_from_u64(6)); have = true; }
    return v[which];
}

std::vector<Fr> unipoly_from_evals(const std::vector<Fr>& evals) {
    const int n = (int)evals.size();
    if (n == 4) {
        // the degree-3 case of every eq-factored round (from12): Newton / finite differences instead of the 4 x 4 matrix --
        // 3 products by constants where the matrix spends 16; the same coefficients (unique interpolation, exact arithmetic):
        //   c3 = (e3 - 3 e2 + 3 e1 - e0) / 6,  c2 = (-e3 + 4 e2 - 5 e1 + 2 e0) / 2,  c1 = (2 e3 - 9 e2 + 18 e1 - 11 e0) / 6,  c0 = e0
        const Fr &e0 = evals[0], &e1 = evals[1], &e2 = evals[2], &e3 = evals[3];
        auto x2 = [](const Fr& a) { return fr_dbl(a); };
        auto x3 = [](const Fr& a) { return fr_add(fr_dbl(a), a); };
        const Fr d1 = fr_sub(e1, e0), d2 = fr_sub(e2, e1), d3 = fr_sub(e3, e2);          // first differences
        const Fr s1 = fr_sub(d2, d1), s2 = fr_sub(d3, d2);                                // second differences
        const Fr t3 = fr_sub(s2, s1);                                                     // third difference = 6 c3
        const Fr c3 = fr_mul(t3, inv_small(1));
        // p(x) = e0 + d1 x + s1 x (x - 1) / 2 + t3 x (x - 1) (x - 2) / 6
        //      = e0 + (d1 - s1 / 2 + t3 / 3) x + (s1 / 2 - t3 / 2) x^2 + (t3 / 6) x^3
        const Fr c2 = fr_mul(fr_sub(s1, t3), inv_small(0));
        const Fr c1 = fr_add(fr_sub(d1, fr_mul(s1, inv_small(0))), x2(c3));              // t3 / 3 = 2 c3
        (void)x3;
        return std::vector<Fr>{e0, c1, c2, c3};
    }
    const std::vector<std::vector<Fr>>& M = lagrange_matrix(n);
    std::vector<Fr> coeffs(n, fr_zero());
    for (int i = 0; i < n; i++)
        for (int k = 0; k < n; k++) coeffs[k] = fr_add(coeffs[k], fr_mul(M[i][k], evals[i]));
    return coeffs;
}

Fr evaluate_univar(const std::vector<Fr>& c, const Fr& x) {
    Fr r = fr_zero();
    for (int i = (int)c.size() - 1; i >= 0; i--) r = fr_add(fr_mul(r, x), c[i]);
    return r;
}

// inverses of (1 - q) for all coordinates of a point with one field inversion (Montgomery's trick);
// a coordinate equal to 1 yields 0 here (the reference would panic on `inverse().unwrap()`, vecvec_eq.rs:205)
std::vector<Fr> batch_inv_one_minus(const std::vector<Fr>& pt) {
    const size_t n = pt.size();
    std::vector<Fr> v(n), pre(n + 1, fr_one()), out(n, fr_zero());
    for (size_t i = 0; i < n; i++) {
        v[i] = fr_sub(fr_one(), pt[i]);
        pre[i + 1] = fr_is_zero(v[i]) ? pre[i] : fr_mul(pre[i], v[i]);
    }
    Fr inv = fr_inv(pre[n]);
    for (size_t i = n; i-- > 0;) {
        if (fr_is_zero(v[i])) continue;
        out[i] = fr_mul(inv, pre[i]);
        inv = fr_mul(inv, v[i]);
    }
    return out;
}

std::vector<Fr> from12_inv(const Fr& p1, const Fr& p2, const Fr& eq1, const Fr& inv_eq0, const Fr& prev_claim) {
    const Fr eq0 = fr_sub(fr_one(), eq1);
    const Fr eq2 = fr_sub(fr_dbl(eq1), eq0);
    const Fr eq3 = fr_sub(fr_dbl(eq2), eq1);
    const Fr prod1 = fr_mul(p1, eq1);
    const Fr prod0 = fr_sub(prev_claim, prod1);
    const Fr p0 = fr_mul(prod0, inv_eq0);
    const Fr p3 = fr_add(fr_sub(fr_sub(fr_add(fr_dbl(p2), p2), fr_dbl(p1)), p1), p0);
    std::vector<Fr> ev = {prod0, prod1, fr_mul(p2, eq2), fr_mul(p3, eq3)};
    return unipoly_from_evals(ev);
}

std::vector<Fr> from12(const Fr& p1, const Fr& p2, const Fr& eq1, const Fr& prev_claim) {
    const Fr eq0 = fr_sub(fr_one(), eq1);
    const Fr eq2 = fr_sub(fr_dbl(eq1), eq0);
    const Fr eq3 = fr_sub(fr_dbl(eq2), eq1);
    const Fr prod1 = fr_mul(p1, eq1);
    const Fr prod0 = fr_sub(prev_claim, prod1);
    const Fr p0 = fr_mul(prod0, fr_inv(eq0));  // undefined in the reference if eq0 == 0 (unwrap panics)
    const Fr p3 = fr_add(fr_sub(fr_sub(fr_add(fr_dbl(p2), p2), fr_dbl(p1)), p1), p0);
    std::vector<Fr> ev = {prod0, prod1, fr_mul(p2, eq2), fr_mul(p3, eq3)};
    return unipoly_from_evals(ev);
}

Fr eq_sum_host(const Fr* pt, uint32_t n, uint64_t k) {
    if (k >= (1ull << n)) return fr_one();
    Fr mult = fr_one(), acc = fr_zero();
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t left = k >> (n - i - 1);
        const Fr prev = mult;
        if (left == 1) {
            mult = fr_mul(mult, pt[i]);
            acc = fr_add(acc, fr_sub(prev, mult));
        } else {
            mult = fr_mul(mult, fr_sub(fr_one(), pt[i]));
        }
        k -= left << (n - i - 1);
    }
    return acc;
}

// ------------------------------------------------------------------------------------------ wait bound
// Kernels that wait on the device for the host's next challenge (k_fold_gate, k_tail_rounds) and the host loops that wait for
// their results give up after this long (gm_set_wait_timeout_ms; default 20 s).  Device side: wall_clock64() ticks (100 MHz).
// (wait_timeout_ms / wait_timeout_ticks / wait_timeout_host: internal.hpp -- the shared-memory communicator waits by the same bound)

// ------------------------------------------------------------------------------------------ kernels
#define SC_THREADS 256

// Coherent 32-byte accesses as two 16-byte instructions: sc1 = device-coherent (another CU / XCD wrote or will read the bytes),
// sc0 sc1 = system-coherent (pinned host memory).  They replace release / acquire fences around hand-offs: on gfx950 an agent- or
// system-scope release writes the whole L2 back (buffer_wbl2) and an acquire invalidates it -- after a fold has left megabytes of
// dirty lines that is microseconds per round.  Stores are left in flight: drain with coh_drain() before raising a flag / counter.
typedef uint32_t gm_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void coh_store_dev(Fr* p, const Fr& v) {
    const gm_u4 lo = {v.l[0], v.l[1], v.l[2], v.l[3]}, hi = {v.l[4], v.l[5], v.l[6], v.l[7]};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(p), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ void coh_store_sys(Fr* p, const Fr& v) {
    const gm_u4 lo = {v.l[0], v.l[1], v.l[2], v.l[3]}, hi = {v.l[4], v.l[5], v.l[6], v.l[7]};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1" ::"v"(p), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ void coh_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ Fr coh_load_dev(const Fr* p) {
    gm_u4 lo, hi;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
    Fr r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w; r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}
__device__ __forceinline__ Fr coh_load_sys(const Fr* p) {
    gm_u4 lo, hi;
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
    Fr r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w; r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}

// Block-wide sum of NACC field elements per thread, then the cross-block sum inside the same launch: a round is ONE kernel + ONE
// host poll.  Grids of <= 512 blocks add their block sums limb by limb into 64-bit accumulators with no-return atomics and the block
// that arrives last (a relaxed device counter behind drained write-through accesses -- no cache-wide fences) swaps the accumulators
// out and reduces them mod p; larger grids store their partials and the last block adds them up.  Either way it writes the NACC
// results straight into pinned host memory.  Sums of canonical values: the order of the blocks does not matter.
struct FinishCtx {
    Fr* partial;         // gridDim.x * gridDim.y rows of NACC elements
    uint32_t* counter;   // zero between launches (the last block resets it)
    Fr* out;             // pinned host memory (device-visible); element 7 doubles as the sequence slot
    uint32_t seq;        // written (system scope) after the results: the host polls it instead of a stream sync
    unsigned long long* acc;   // 3 x 8 limb accumulators (64-bit, one 128-byte line each), zero between launches; grids of <= 512 blocks
                               // add their block sums into them with atomics instead of storing partials (see k_stage's exchange)
    uint32_t raw;              // 1: `out` is read by the host (RoundScratch::finish_seq), which takes the 64-bit limb sums as they are and
                               // reduces them mod p itself (format word 1 next to the sequence number); 0: `out` receives field elements
};
#define FINISH_ATOMIC_MAX_BLOCKS 512u
// V = sum of <= 512 canonical field elements given as eight 64-bit limb sums (low words lo, bits 32.. hi): V mod p
GM_HD Fr limb_sums_mod_p(const uint32_t* lo8, const uint32_t* hi8) {
    Fr lo, hi, c32;
#pragma unroll
    for (int l = 0; l < 8; l++) { lo.l[l] = lo8[l]; hi.l[l] = hi8[l]; }
    // 2^32 in Montgomery form (2^32 R mod p): hi < 2^233 < p is a canonical operand, lo < 2^256 < 2.21 p needs two conditional subtractions
    c32.l[0] = 0xcaaf6b13u; c32.l[1] = 0x355094eau; c32.l[2] = 0x69a568efu; c32.l[3] = 0xf6b10cb3u;
    c32.l[4] = 0x40cc3869u; c32.l[5] = 0xe2c926a6u; c32.l[6] = 0xed269aadu; c32.l[7] = 0x736a6d3bu;
    return fr_add(fr_reduce_once(fr_reduce_once(lo)), fr_mul(hi, c32));
}

// sum of `v` over the 64 lanes of the wave; the result is valid in lane 0
__device__ __forceinline__ Fr wave_sum(Fr v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        Fr t;
#pragma unroll
        for (int l = 0; l < 8; l++) t.l[l] = __shfl_down(v.l[l], d, 64);
        v = fr_add(v, t);
    }
    return v;
}

// The same sum WITHOUT the modular additions: the sixteen 16-bit halves of the eight limbs summed over the wave as plain integers
// (each below 2^22), by DPP adds -- four row shifts, two row broadcasts, no LDS traffic, no carry chains, no conditional subtractions
// (wave_sum costs ~1 800 cycles of a lone wave, this ~400).  The cross-block accumulators take integer limb sums anyway
// (limb_sums_mod_p reduces them once, in the block that arrives last), so a block never needs its own sum reduced.  Result: wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_acc(uint32_t x) {
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ void wave_half_sums(const Fr& v, uint32_t* out16) {
    uint32_t h[16];
#pragma unroll
    for (int l = 0; l < 8; l++) { h[2 * l] = v.l[l] & 0xffffu; h[2 * l + 1] = v.l[l] >> 16; }
    // row_shr:1, 2, 4, 8: lane 15 of every row of 16 holds the row's sum; row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3:
    // lane 63 holds the wave's
#pragma unroll
    for (int k = 0; k < 16; k++) h[k] = dpp_acc<0x111, 0xf>(h[k]);
#pragma unroll
    for (int k = 0; k < 16; k++) h[k] = dpp_acc<0x112, 0xf>(h[k]);
#pragma unroll
    for (int k = 0; k < 16; k++) h[k] = dpp_acc<0x114, 0xf>(h[k]);
#pragma unroll
    for (int k = 0; k < 16; k++) h[k] = dpp_acc<0x118, 0xf>(h[k]);
#pragma unroll
    for (int k = 0; k < 16; k++) h[k] = dpp_acc<0x142, 0xa>(h[k]);
#pragma unroll
    for (int k = 0; k < 16; k++) h[k] = dpp_acc<0x143, 0xc>(h[k]);
#pragma unroll
    for (int k = 0; k < 16; k++) out16[k] = (uint32_t)__builtin_amdgcn_readlane((int)h[k], 63);
}

// block-wide sums of acc[0..NACC): wave shuffles, then one pass over the SC_THREADS / 64 wave totals.
// Valid in thread a (a < NACC) as the return value; one barrier.
template <int NACC>
__device__ __forceinline__ Fr block_sum(Fr* acc, Fr (*red)[NACC]) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        const Fr w = wave_sum(acc[a]);
        if (lane == 0) red[wave][a] = w;
    }
    __syncthreads();
    Fr tot = fr_zero();
    if (threadIdx.x < NACC) {
#pragma unroll
        for (int w = 0; w < SC_THREADS / 64; w++) tot = fr_add(tot, red[w][threadIdx.x]);
    }
    return tot;
}

template <int NACC>
__device__ __forceinline__ void block_reduce_finish(Fr* acc, const FinishCtx& fc) {
    __shared__ Fr red[SC_THREADS / 64][NACC];
    __shared__ uint32_t is_last;
    const uint32_t nblk = gridDim.x * gridDim.y;
    const uint32_t bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (fc.acc && nblk <= FINISH_ATOMIC_MAX_BLOCKS) {
        // small and medium grids: no second pass over stored partials (they are what a latency-bound round waits for)
        __shared__ uint32_t s_lo[8 * NACC], s_hi[8 * NACC];
        __shared__ uint32_t half[SC_THREADS / 64][NACC][16];
        {
            const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                uint32_t hs[16];
                wave_half_sums(acc[a], hs);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 16; k++) half[wave][a][k] = hs[k];
                }
            }
        }
        __syncthreads();
        if (fc.raw && NACC <= 3) {
            // ONE trip to the L2 (as k_stage's exchange): a returning add per (sum, limb) that also counts the contributors in the
            // accumulator's top bits; the block whose add arrives last at an accumulator forwards that total to the host -- 8 bytes,
            // self-validating: a 12-bit tag of the launch's sequence number (never 0) over the 52-bit total -- and leaves a zero behind.
            // No drain, no arrival counter, no swap by a last block, no sequence word: the host waits for the 8 NACC tagged values
            // (RoundScratch::finish_seq, which zeroes them once read) and reduces mod p.
            if (threadIdx.x < 8 * NACC) {
                const uint32_t a = threadIdx.x >> 3, l = threadIdx.x & 7;
                unsigned long long v = 0;
#pragma unroll
                for (int w = 0; w < SC_THREADS / 64; w++) v += (unsigned long long)half[w][a][2 * l] + ((unsigned long long)half[w][a][2 * l + 1] << 16);
                const unsigned long long prev = __hip_atomic_fetch_add(fc.acc + threadIdx.x * 16, v + (1ull << 52), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(prev >> 52) == nblk - 1) {
                    const unsigned long long tot = (prev + v) & ((1ull << 52) - 1);
                    const unsigned long long tag = (unsigned long long)(fc.seq % 4095u + 1u) << 52;
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(fc.out) + threadIdx.x, tag | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(fc.acc + threadIdx.x * 16, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
        if (threadIdx.x < 8 * NACC) {   // one lane per (sum, limb): the block's integer limb sum (< 2^42) into the launch's accumulator
            const uint32_t a = threadIdx.x >> 3, l = threadIdx.x & 7;
            unsigned long long v = 0;
#pragma unroll
            for (int w = 0; w < SC_THREADS / 64; w++) v += (unsigned long long)half[w][a][2 * l] + ((unsigned long long)half[w][a][2 * l + 1] << 16);
            (void)__hip_atomic_fetch_add(fc.acc + threadIdx.x * 16, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            coh_drain();
        }
        __syncthreads();   // this block's additions have been performed before the counter moves
        if (threadIdx.x == 0) {
            const uint32_t prev = __hip_atomic_fetch_add(fc.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            is_last = (prev == nblk - 1) ? 1u : 0u;
        }
        __syncthreads();
        if (!is_last) return;
        if (threadIdx.x < 8 * NACC) {
            const unsigned long long v = __hip_atomic_exchange(fc.acc + threadIdx.x * 16, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_lo[threadIdx.x] = (uint32_t)v;
            s_hi[threadIdx.x] = (uint32_t)(v >> 32);
        }
        __syncthreads();
        if (threadIdx.x < NACC) {
            coh_store_sys(fc.out + threadIdx.x, limb_sums_mod_p(s_lo + 8 * threadIdx.x, s_hi + 8 * threadIdx.x));
            coh_drain();
        }
        __syncthreads();  // the results have reached host memory before the sequence word is written
        if (threadIdx.x == 0) {
            __hip_atomic_store(fc.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(fc.out + 7), (unsigned long long)fc.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    {
        const Fr tot = block_sum<NACC>(acc, red);
        // hand-off without cache-wide fences (see the coherent helpers above): write-through stores, drained, then the counter
        if (threadIdx.x < NACC) { coh_store_dev(fc.partial + (uint64_t)bid * NACC + threadIdx.x, tot); coh_drain(); }
    }
    __syncthreads();  // all partial stores of this block have completed before the counter moves
    if (threadIdx.x == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(fc.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == nblk - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
    Fr s2[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        Fr s = fr_zero();
        for (uint32_t b2 = threadIdx.x; b2 < nblk; b2 += SC_THREADS) s = fr_add(s, coh_load_dev(fc.partial + (uint64_t)b2 * NACC + a));
        s2[a] = s;
    }
    {
        const Fr tot = block_sum<NACC>(s2, red);
        if (threadIdx.x < NACC) { coh_store_sys(fc.out + threadIdx.x, tot); coh_drain(); }
    }
    __syncthreads();  // the results have reached host memory before the sequence word is written
    if (threadIdx.x == 0) {
        __hip_atomic_store(fc.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(fc.out + 7), (unsigned long long)fc.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct SmallVals {
    Fr v[48];
};
__global__ void k_upload_small(SmallVals sv, int n, Fr* __restrict__ dst) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) fr_store(dst + i, sv.v[i]);
}

// Sharded rounds with a device-side collective (gm_comm::all_gather_dev): every rank's round kernel leaves its partial sums in a
// device slot (4 field elements), the slots are all-gathered on the prover's stream, and this one wave adds the `world` parts mod p
// and reports ONCE to pinned host memory -- one host hop per round instead of D2H + host gather + H2D + D2H + a stream sync.
__global__ void __launch_bounds__(64) k_sum_ranks(const Fr* __restrict__ all, uint32_t world, int nacc, Fr* __restrict__ h_out, uint32_t seq) {
    if ((int)threadIdx.x < nacc) {
        Fr s = fr_zero();
        for (uint32_t r = 0; r < world; r++) s = fr_add(s, fr_load(all + (size_t)r * 4 + threadIdx.x));
        coh_store_sys(h_out + threadIdx.x, s);
        coh_drain();
    }
    __syncthreads();   // the sums have reached host memory before the sequence word is written
    if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(h_out + 7), (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One segment of the gamma-combined layer function at the "1" point (h = 0: p1) or the "2" point (h = 1: 2 p1 - p0)
// of the pair starting at cell0:  sum_{o in segment} gamma^o f_o(.)
__device__ __forceinline__ Fr eval_seg(const Seg& g, const ColPtrs& cols, const Fr* __restrict__ gp, uint64_t cell0, int h) {
    Fr v[6], o[4];
#pragma unroll
    for (int q = 0; q < 6; q++)
        if (q < g.n_in) {
            const Fr p1 = fr_load(cols.p[g.in[q]] + cell0 + 1);
            v[q] = h ? fr_sub(fr_dbl(p1), fr_load(cols.p[g.in[q]] + cell0)) : p1;
        }
    prim_exec(g.prim, v, o);
    Fr A = fr_zero();
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q < g.n_out) {
            const int oc = g.out0 + q;
            A = fr_add(A, oc == 0 ? o[q] : fr_mul(fr_load(gp + oc), o[q]));
        }
    return A;
}

// Deg-2 round sums with eq factored out.
//   dense  (dense_eq.rs:121-139):  S_h = sum_i eq[i] * A_h(i)
//   VecVec (vecvec_eq.rs:320-361): S_h = sum_r coef[r] sum_idx eq_row[idx] * A_h(r, idx), plus acc[2] = the tail
//          weight W = sum_r coef[r] * (1 - sum_{idx < seg_r} eq_row[idx])  (get_trailing_sum, vecvec.rs:144-146)
// SPLIT = false: one thread per pair walks all segments and both points (large layers: every element is loaded once).
// SPLIT = true : blockIdx.y = 2 * segment + h; one thread per (pair, segment, point).  Small layers are latency bound
//                (a lone wave needs ~1 us per field multiplication), so the serial chain per thread is what matters.
struct VVArgs {
    const uint32_t* off;
    uint32_t nrows;
    const Fr* row_coef;
    const Fr* eq_prefix;
    const uint32_t* coarse;   // row of cell c << GM_COARSE_SHIFT for every c (gm_vv::coarse), or nullptr: full binary search
};

template <bool VECVEC, bool SPLIT>
__global__ void __launch_bounds__(SC_THREADS) k_round_deg2(SegPlan sp, ColPtrs cols, const Fr* __restrict__ eq,
                                                            const Fr* __restrict__ gp, uint64_t npairs_dense, VVArgs vv,
                                                            FinishCtx fc) {
    constexpr int NACC = VECVEC ? 3 : 2;
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    if (VECVEC && blockIdx.y == 0) {
        for (uint32_t r = blockIdx.x * SC_THREADS + threadIdx.x; r < vv.nrows; r += gridDim.x * SC_THREADS) {
            const uint32_t seg = (vv.off[r + 1] - vv.off[r]) >> 1;
            acc[2] = fr_add(acc[2], fr_mul(fr_load(vv.row_coef + r), fr_sub(fr_one(), fr_load(vv.eq_prefix + seg))));
        }
    }
    const uint64_t npairs = VECVEC ? (uint64_t)(vv.off[vv.nrows] >> 1) : npairs_dense;
    for (uint64_t base = (uint64_t)blockIdx.x * SC_THREADS; base < npairs; base += (uint64_t)gridDim.x * SC_THREADS) {
        const uint64_t i = base + threadIdx.x;
        const bool valid = i < npairs;
        Fr w;
        if (VECVEC) {
            const uint32_t cell0 = (uint32_t)(2 * i);
            const uint64_t last_pair = (base + SC_THREADS - 1 < npairs) ? base + SC_THREADS - 1 : npairs - 1;
            // small (split) launches are latency bound: bracket the block's rows once instead of log2(nrows) dependent
            // loads per thread; large launches hide that latency behind other waves and must not pay the barriers
            uint32_t r;
            if (SPLIT && vv.coarse) {  // the coarse table brackets the row in one load (block-uniform condition)
                if (!valid) continue;
                r = find_row_coarse(vv.off, vv.nrows, vv.coarse, cell0);
            } else if (SPLIT && (uint64_t)gridDim.x * SC_THREADS >= npairs) {  // single pass (block-uniform condition)
                r = find_row_span(vv.off, vv.nrows, cell0, valid, (uint32_t)(2 * base), (uint32_t)(2 * last_pair));
                if (!valid) continue;
            } else {
                if (!valid) continue;
                r = find_row(vv.off, vv.nrows, cell0);
            }
            w = fr_mul(fr_load(eq + ((cell0 - vv.off[r]) >> 1)), fr_load(vv.row_coef + r));
        } else {
            if (!valid) continue;
            w = fr_load(eq + i);
        }
        if (SPLIT) {
            const int sgi = blockIdx.y >> 1, h = blockIdx.y & 1;
            const Seg g = sp.seg[sgi];
            acc[h] = fr_add(acc[h], fr_mul(eval_seg(g, cols, gp, 2 * i, h), w));
        } else {
            Fr A1 = fr_zero(), A2 = fr_zero();
            for (int sgi = 0; sgi < sp.nseg; sgi++) {
                const Seg g = sp.seg[sgi];
                A1 = fr_add(A1, eval_seg(g, cols, gp, 2 * i, 0));
                A2 = fr_add(A2, eval_seg(g, cols, gp, 2 * i, 1));
            }
            acc[0] = fr_add(acc[0], fr_mul(A1, w));
            acc[1] = fr_add(acc[1], fr_mul(A2, w));
        }
    }
    block_reduce_finish<NACC>(acc, fc);
}

// ------------------------------------------------------------------------------------------ pre-enqueued folds (small rounds)
// A small round costs ~22 us of kernel time but ~45 us of wall time: after the host has the round sums it still has to
// launch the fold and the next round kernel, and each launch takes ~5 us of API time plus ~4 us until the GPU starts it.
// Small rounds therefore enqueue the fold of round r and the round kernel of round r + 1 BEFORE the challenge t_r exists:
// a gate kernel in front of the fold waits (bounded) for the host to publish t_r in pinned memory.  The launch latency is
// spent while round r's kernel is still running.  The wait is bounded (gm_set_wait_timeout_ms, default 20 s); on timeout the fold reports through a
// status word and exits, so a host that never answers cannot wedge the GPU.
// One wave (the gate) polls the host's ticket word -- thousands of fold blocks polling over PCIe would queue behind each
// other's reads -- and copies the challenge into device memory; the fold behind it in the stream is an ordinary kernel.
__global__ void __launch_bounds__(64) k_fold_gate(const Fr* __restrict__ t_slot, const uint32_t* __restrict__ ticket_word, uint32_t ticket,
                                                  uint32_t* __restrict__ status, Fr* __restrict__ d_t, uint64_t timeout_ticks) {
    if (threadIdx.x != 0) return;
    const uint64_t t_begin = wall_clock64();
    for (uint32_t it = 0;; it++) {
        if ((it & 255u) == 255u && wall_clock64() - t_begin > timeout_ticks) break;
        // no acquire (it would invalidate the caches on every poll): the host stores t, then the ticket (x86 store order), and both
        // are read straight from host memory, the challenge after the ticket has been seen, as two 16-byte loads (one round trip)
        const uint32_t f = __hip_atomic_load(ticket_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(f - ticket) >= 0) {
            *d_t = coh_load_sys(t_slot);
            return;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // the fold runs on a stale t; the host reports the error
}

__global__ void __launch_bounds__(256) k_dense_fold_dev(ColPtrs in, ColPtrsMut out, uint64_t n_out, const Fr* __restrict__ d_t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const Fr t = fr_load(d_t);
    const Fr* src = in.p[blockIdx.y];
    const Fr p0 = fr_load(src + 2 * i), p1 = fr_load(src + 2 * i + 1);
    fr_store(out.p[blockIdx.y] + i, fr_add(p0, fr_mul(t, fr_sub(p1, p0))));
}

// Final evaluations of a finished sumcheck: element 0 of every column, gathered by one wave straight into pinned host
// memory, then a sequence number the host polls (k separate 32-byte copies cost ~20 us each).
__global__ void __launch_bounds__(64) k_gather_finals(ColPtrs cols, int k, Fr* __restrict__ h_out, uint32_t* __restrict__ h_seq, uint32_t seq) {
    for (int i = threadIdx.x; i < k; i += 64) coh_store_sys(h_out + i, fr_load(cols.p[i]));
    coh_drain();       // the values have reached host memory ...
    __syncthreads();   // ... for every lane, before the sequence word is written (no cache-wide fence)
    if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------ persistent stage kernel
// Small rounds are latency-bound: as separate dispatches a round costs a round kernel (21-29 us), a gate (5-9 us) and a fold
// (5-8 us).  ONE launch of k_stage runs every small round of a layer instead:
//   * the "thin" rounds of a VecVec sumcheck -- the rounds left once every row is down to 0 or 2 cells (rows halve every round
//     and are re-padded to even length, vecvec.rs:420-441, so the last ~7 of the 19 sparse rounds of a bucket-sum layer work on
//     one pair per row): thread = row, the pair lives in registers, a fold re-pads in registers;
//   * bind_into_dense (vecvec_eq.rs:157-175): row -> one dense element, regrouped into pairs through LDS;
//   * the whole dense stage (13 rounds at config B) -- or all rounds of a dense object that starts small (triangle layers).
// Grid: x = 2 * segment + evaluation point, y = slice of 256 rows / dense elements.  Per round a block (i) evaluates its
// segment of the layer function on its pairs, reduces over the block and adds its sum into the round's limb accumulators; the
// block that arrives last reduces them mod p and writes ONE report (three self-validating values) into pinned host memory (the host
// does the O(1) scalar tail: from12, transcript, challenge), (ii) waits for the challenge -- in device memory the host writes
// directly over the large BAR, or, without one, block (0, 0) polls the host's ticket over PCIe and relays it through device
// memory -- (iii) folds its own inputs and regroups the pairs through LDS.  Slices never exchange data until each is
// down to one element; then every slice hands its element to slice 0 through device memory (release / acquire around a
// counter) and slice 0 finishes alone.  All blocks are co-resident (<= 512 blocks of 256 threads), so waiting is safe; every
// wait is bounded (gm_set_wait_timeout_ms): on timeout a block flags `status` and leaves.
// Coherent 32-byte accesses as two 16-byte instructions.  Relaxed atomic dword accesses would do, but the compiler drains the
// memory pipeline (s_waitcnt vmcnt(0)) after every one of them: eight dependent round trips per field element (measured: 6.6 us
// to publish two elements).  sc1 = device-coherent (another CU / XCD wrote or will read the bytes), sc0 sc1 = system-coherent
// (pinned host memory).  Stores are left in flight: drain with coh_drain() before raising the flag / counter.
// Self-validating messages between the stage kernel and the host (and between its blocks): 16-byte chunks of 12 data bytes +
// the 4-byte sequence number of the round.  Every chunk is one store instruction and is validated on its own, so the writer
// needs no ordering between stores (no drain, no separate flag) and the reader gets data and flag in ONE round trip.
__device__ __forceinline__ void chunk_store_sys(uint32_t* p, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t seq) {
    const gm_u4 v = {a0, a1, a2, seq};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void chunk_store_dev(uint32_t* p, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t seq) {
    const gm_u4 v = {a0, a1, a2, seq};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// a field element as three chunks (12 + 12 + 8 bytes) at p[0..12)
__device__ __forceinline__ void fr_chunks_store_sys(uint32_t* p, const Fr& v, uint32_t seq) {
    chunk_store_sys(p, v.l[0], v.l[1], v.l[2], seq);
    chunk_store_sys(p + 4, v.l[3], v.l[4], v.l[5], seq);
    chunk_store_sys(p + 8, v.l[6], v.l[7], 0u, seq);
}
__device__ __forceinline__ void fr_chunks_store_dev(uint32_t* p, const Fr& v, uint32_t seq) {
    chunk_store_dev(p, v.l[0], v.l[1], v.l[2], seq);
    chunk_store_dev(p + 4, v.l[3], v.l[4], v.l[5], seq);
    chunk_store_dev(p + 8, v.l[6], v.l[7], 0u, seq);
}
// load the three chunks of a field element; 1: every chunk carries `want`, 2: every chunk carries `alt`, 0: neither (yet)
template <bool SYS>
__device__ __forceinline__ int fr_chunks_load(const uint32_t* p, Fr* out, uint32_t want, uint32_t alt) {
    gm_u4 c0, c1, c2;
    if (SYS)
        asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc0 sc1\n\t"
                     "global_load_dwordx4 %2, %3, off offset:32 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(c0), "=&v"(c1), "=&v"(c2) : "v"(p) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
                     "global_load_dwordx4 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(c0), "=&v"(c1), "=&v"(c2) : "v"(p) : "memory");
    out->l[0] = c0.x; out->l[1] = c0.y; out->l[2] = c0.z; out->l[3] = c1.x; out->l[4] = c1.y; out->l[5] = c1.z; out->l[6] = c2.x; out->l[7] = c2.y;
    if (c0.w == want && c1.w == want && c2.w == want) return 1;
    if (c0.w == alt && c1.w == alt && c2.w == alt) return 2;
    return 0;
}

// The gate of a pre-enqueued fold inside the fold itself (small folds, when the host can write challenges into device memory -- the
// large BAR, see TailStage / RoundScratch::bar): lane 0 of every workgroup polls the challenge's self-validating chunks in DEVICE
// memory (no PCIe read per poll, so hundreds of pollers are fine), the workgroup takes it from LDS.  Saves the gate launch and the
// kernel boundary between gate and fold (~3-5 us per pre-enqueued round).  Timeout: status word, stale challenge, host reports.
struct GateArgs {
    const uint32_t* bar_slot;   // nullptr: the challenge comes through d_t (k_fold_gate ran in front of this kernel)
    uint32_t ticket;
    uint32_t* status;
    uint64_t timeout_ticks;
};
__device__ __forceinline__ Fr gated_challenge(const GateArgs& g, const Fr* __restrict__ d_t) {
    if (!g.bar_slot) return fr_load(d_t);
    __shared__ Fr t_sh;
    if (threadIdx.x == 0) {
        Fr t = fr_zero();
        const uint64_t t_begin = wall_clock64();
        bool good = false;
        for (uint32_t it = 0;; it++) {
            if ((it & 255u) == 255u && wall_clock64() - t_begin > g.timeout_ticks) break;
            if (fr_chunks_load<true>(g.bar_slot, &t, g.ticket, g.ticket) == 1) { good = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!good && blockIdx.x == 0 && blockIdx.y == 0) __hip_atomic_store(g.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        t_sh = t;
    }
    __syncthreads();
    return t_sh;
}

#define STAGE_MAX_BLOCKS 512
#define STAGE_MAX_SLICES 64
#define STAGE_MAX_ROUNDS 16
struct PadCols {
    Fr v[16];
};
struct StageArgs {
    uint32_t nrows;                    // thin phase: stored rows (each of 0 or 2 cells); 0 = no thin phase
    uint32_t n_elems;                  // dense elements (a power of two): 2^col_logsize, or the dense object's current length
    int n_thin, n_dense;               // rounds of each phase
    const uint32_t* off;               // thin: row offsets
    const Fr* row_coef;                // thin: eq(point[0..col_logsize], row)
    const Fr* thin_eq[12];             // thin round q: entry 0 of that round's level of the row eq sequence
    const Fr* eq[STAGE_MAX_ROUNDS];    // dense round q: eq table indexed by the global pair index
    PadCols row_pad, col_pad;          // thin only
    uint32_t* h_rep;                   // pinned: the round's report at 96 (round & 1) words: 24 self-validating chunks, one per 64-bit limb
                                       // sum (8 limbs each of the sum at point 1, at point 2, of the tail weight): the host reduces mod p
    uint32_t* d_round_cnt;             // (unused since the accumulators count their own contributors)
    Fr* h_finals;                      // pinned: what is left of column c after the last round, 32 slots per column
    uint32_t* h_fin_seq;               // pinned: word [segment][slice] = ticket0 + rounds once that block's part is written
    const uint32_t* h_tkt;             // pinned: the host publishes t_r as three chunks tagged ticket0 + r at 12 (r & 1) words
    uint32_t* h_status;
    uint32_t* d_relay;                 // device: the relayed challenge of round r as three chunks at 12 (r & 1) words
    const uint32_t* d_tkt_bar;         // fine-grained DEVICE memory the host writes t_r into directly (large BAR), same layout as h_tkt; nullptr:
                                       // block 0 polls h_tkt over PCIe and relays through d_relay
    Fr* d_xbuf;                        // device: hand-over of the slices' last elements, [gridDim.x][6][STAGE_MAX_SLICES]
    uint32_t* d_merge;                 // device: one arrival counter per blockIdx.x (zeroed before the launch)
    uint32_t ticket0;
    uint64_t timeout_ticks;
    unsigned long long* d_acc;         // device: per round 3 sums x 8 limb accumulators (64-bit, one 128-byte line each), zero between launches
    uint32_t* d_arrive;                // device: residency barrier -- cumulative count of blocks that have started (bit 31: a launch gave up)
    uint32_t arrive_target;            // ... value it has once every block of THIS launch is resident
    uint64_t resident_ticks;           // ... how long a block waits for the others before the launch is abandoned (100 MHz ticks)
    uint64_t* d_dbg;                   // development aid: phase time stamps (nullptr normally)
    uint32_t par_fold;                 // dense rounds: fold by (input, pair) lanes instead of by pair lanes (GM_STAGE_PAR_FOLD=0: off)
};

// The stage kernel's products.  k_stage<3> is 51 000 instructions (400 KB) with every product inlined, and blocks of different segments
// / parts share an instruction cache; ONE out-of-line copy of the multiplier (-DGM_STAGE_MUL_OUTLINE: a call costs a handful of
// instructions against the product's 302) was measured in round 4 and changes nothing: image part 49.1-49.9 ms against 48.8-50.3
// inlined (best of 8, three alternating runs, same box) -- the instruction cache is not what a stage round waits for.  Inlined it stays.
#ifdef GM_STAGE_MUL_OUTLINE
__device__ __attribute__((noinline)) Fr fr_mul_s(Fr a, Fr b) { return fr_mul(a, b); }
#else
__device__ __forceinline__ Fr fr_mul_s(const Fr& a, const Fr& b) { return fr_mul(a, b); }
#endif
__device__ __forceinline__ Fr fr_mul_by_d_s(const Fr& x) { return fr_mul_s(x, fr_coeff_d()); }

// one part of a split segment (segfn.hip.h): its share of sum_o gamma^o f_o; G(k, x) = gamma^(out0 + k) x
__device__ __forceinline__ Fr stage_part_eval(int prim, int out0, const Fr* v, const Fr* __restrict__ gp) {
    auto G = [&](int k, const Fr& x) -> Fr { return out0 + k == 0 ? x : fr_mul_s(fr_load(gp + out0 + k), x); };
    auto x5 = [](const Fr& x) -> Fr { return fr_add(fr_dbl(fr_dbl(x)), x); };   // -a x, a = -5
    switch (prim) {
        case FN_P_PROJ_L1_A: return fr_mul_s(v[0], fr_add(G(0, v[2]), x5(G(2, v[1]))));          // v0, v3, v4
        case FN_P_PROJ_L1_B: return fr_mul_s(v[0], fr_add(G(1, v[1]), G(2, v[2])));              // v1, v3, v4
        case FN_P_PROJ_L1_C: return G(3, fr_mul_s(v[0], v[1]));                                  // v2, v5
        case FN_P_PROJ_L2_A: return fr_add(G(0, fr_mul_s(fr_add(v[0], v[1]), v[2])), G(3, fr_mul_s(v[0], v[1])));   // v0, v1, v3
        case FN_P_PROJ_L2_B: return fr_mul_s(v[1], fr_add(G(1, v[0]), G(2, v[1])));              // v2, v3
        case FN_P_PROJ_L3_A: return G(0, fr_mul_s(fr_sub(v[1], fr_mul_by_d_s(v[2])), v[0]));       // v0, v2, v3
        case FN_P_PROJ_L3_B: return G(1, fr_mul_s(fr_add(v[1], fr_mul_by_d_s(v[2])), v[0]));       // v1, v2, v3
        case FN_P_PROJ_L3_C: {                                                                 // v2, v3
            const Fr dxy = fr_mul_by_d_s(v[1]);
            return G(2, fr_mul_s(fr_sub(v[0], dxy), fr_add(v[0], dxy)));
        }
        case FN_P_AFF_L1_A: return fr_mul_s(v[0], fr_add(G(0, v[2]), x5(G(2, v[1]))));           // v0, v2, v3
        case FN_P_AFF_L1_B: return fr_mul_s(v[0], fr_add(G(1, v[1]), G(2, v[2])));               // v1, v2, v3
        case FN_P_LOGUP_A: return fr_mul_s(v[2], fr_add(G(0, v[0]), G(1, v[1])));                // v0, v1, v3
        case FN_P_LOGUP_B: return G(0, fr_mul_s(v[0], v[1]));                                    // v1, v2
        case FN_P_AFF_L3_A: return G(0, fr_mul_s(fr_sub(fr_one(), fr_mul_by_d_s(v[1])), v[0]));    // v0, v2
        case FN_P_AFF_L3_B: return G(1, fr_mul_s(fr_add(fr_one(), fr_mul_by_d_s(v[1])), v[0]));    // v1, v2
        default: {                                                                             // FN_P_AFF_L3_C: v2
            const Fr dxy = fr_mul_by_d_s(v[0]);
            return G(2, fr_mul_s(fr_sub(fr_one(), dxy), fr_add(fr_one(), dxy)));
        }
    }
}

// the primitives with at most three inputs (the narrow instance of the stage kernel: the wide arms of prim_exec would only cost it
// registers); outputs as prim_exec
__device__ __forceinline__ void prim_exec3(int id, const Fr* a, Fr* o) {
    switch (id) {
        case FN_AFF_L2: aff_l2(a, o); break;
        case FN_AFF_L3: aff_l3(a, o); break;
        case FN_ID: o[0] = a[0]; break;
        case FN_BITCHECK: o[0] = fr_sub(fr_sqr(a[0]), a[0]); break;
        case FN_PT_BIT_CHOICE: {
            o[0] = fr_mul(a[0], a[1]);
            o[1] = fr_add(fr_mul(a[0], fr_sub(a[2], fr_one())), fr_one());
        } break;
        case FN_ADD_INVERSES: {
            o[0] = fr_add(a[0], a[1]);
            o[1] = fr_mul(a[0], a[1]);
        } break;
        default: break;
    }
}
__host__ __device__ constexpr bool prim_fits3(int prim) {
    return prim >= 64 || prim == FN_AFF_L2 || prim == FN_AFF_L3 || prim == FN_ID || prim == FN_BITCHECK || prim == FN_PT_BIT_CHOICE ||
           prim == FN_ADD_INVERSES;
}

// sum_{o in segment} gamma^o f_o(v) for one pair at evaluation point h
template <int MAXIN>
__device__ __forceinline__ Fr stage_eval(const Seg& g, const Fr* p0, const Fr* p1, const Fr* __restrict__ gp, int h) {
    Fr v[MAXIN], o[4];
#pragma unroll
    for (int q = 0; q < MAXIN; q++) {
        v[q] = p1[q];
        if (h && q < g.n_in) v[q] = fr_sub(fr_dbl(p1[q]), p0[q]);   // a lone wave: skip the inputs the segment does not have
    }
    if (prim_is_part(g.prim)) return stage_part_eval(g.prim, g.out0, v, gp);
    if (MAXIN == 3) prim_exec3(g.prim, v, o); else prim_exec(g.prim, v, o);
    Fr A = fr_zero();
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q < g.n_out) {
            const int oc = g.out0 + q;
            A = fr_add(A, oc == 0 ? o[q] : fr_mul_s(fr_load(gp + oc), o[q]));
        }
    return A;
}

// MAXIN = inputs a segment may have: 6 (any plan) or 3 (every segment of the plan is a part or a narrow primitive -- all the
// split twisted-Edwards layers): the narrow instance keeps half the state in registers and in LDS.
template <int MAXIN>
__global__ void __launch_bounds__(256) k_stage(SegPlan sp, ColPtrs cols, const Fr* __restrict__ gp, StageArgs a) {
    __shared__ Fr xch[MAXIN][256];
    // the UNFOLDED pairs of a dense round, staged while the block waits for the challenge: the fold is then one product per (input, pair)
    // lane -- every lane of the block busy for one product -- instead of n_in dependent-issue products per pair lane (a lone wave per SIMD
    // issues ~one instruction per 7 cycles: three products of a fold are ~3.6 us, one is ~1.2)
    __shared__ Fr pre0[MAXIN][128], pre1[MAXIN][128];
    __shared__ uint32_t half[4][2][16];
    __shared__ Fr ts;
    __shared__ int ok;
    const int sgi = blockIdx.x >> 1, h = blockIdx.x & 1;
    const Seg g = sp.seg[sgi];
    const uint32_t slice = blockIdx.y, nsl = gridDim.y;
    const uint32_t i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const uint32_t blk = slice * gridDim.x + blockIdx.x;
    Fr p0[MAXIN], p1[MAXIN];
#pragma unroll
    for (int q = 0; q < MAXIN; q++) { p0[q] = fr_zero(); p1[q] = fr_zero(); }
    uint32_t round = 0;   // rounds done by this launch so far: the ticket of round r is ticket0 + r
    // development aid (GM_STAGE_DEBUG=1): wall-clock stamps (100 MHz) of the phases of every round, blocks 0 and 1
#define STAGE_STAMP(k_) do { if (a.d_dbg && i == 0 && blk < 2) a.d_dbg[((size_t)blk * 32 + round) * 8 + (k_)] = wall_clock64(); } while (0)

    // Residency barrier.  The blocks of a launch wait for each other every round, so all of them must be resident at once.  The
    // host only launches what fits an EMPTY device (StageSlots), but kernels of other streams or processes may hold compute units:
    // every block counts itself in and waits -- briefly -- until the whole grid has; the block whose patience runs out flips the
    // abandon bit (compare-and-swap against the count it saw: "everyone is here" and "abandoned" exclude each other), reports status 2
    // and everybody leaves, including blocks that only become resident later.  The host then runs the layer's rounds as ordinary
    // kernels (StageRun::sums -> GM_STAGE_NOT_RESIDENT).  Nothing has been written by then: the columns are read-only here.
    if (i == 0) {
        int good = 0;
        uint32_t v = __hip_atomic_fetch_add(a.d_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        const uint64_t t_begin = wall_clock64();
        for (uint32_t it = 0;; it++) {
            if (v & 0x80000000u) break;
            if (v >= a.arrive_target) { good = 1; break; }
            if ((it & 31u) == 31u && wall_clock64() - t_begin > a.resident_ticks) {
                uint32_t seen = v;
                if (__hip_atomic_compare_exchange_strong(a.d_arrive, &seen, v | 0x80000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.h_status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                v = seen;
                continue;
            }
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(a.d_arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ok = good;
    }
    __syncthreads();
    if (!ok) return;

    // Sum (s0, s1) over all reporting blocks and hand the round's sums to the host, then wait for that round's challenge;
    // false = give up.  Blocks publish their partial in device memory; the block that arrives last (agent-scope release /
    // acquire around the round's own counter) adds them up per evaluation point and writes ONE report to pinned host memory:
    // with up to 192 blocks, per-block reports cost ~45 us per round in PCIe write transactions alone.
    uint32_t stage_np = 0;   // > 0: exchange() stages this many pairs (p0 / p1 of the calling lane) before it waits
    auto exchange = [&](Fr s0, Fr s1, bool with_w, uint32_t nrep) -> bool {
        STAGE_STAMP(1);
        {   // integer limb sums over the wave (see wave_half_sums): the block's share goes into the accumulators unreduced
            uint32_t hs[16];
            wave_half_sums(s0, hs);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 16; k++) half[wave][0][k] = hs[k];
            }
            if (with_w) {
                wave_half_sums(s1, hs);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 16; k++) half[wave][1][k] = hs[k];
                }
            }
        }
        __syncthreads();
        const uint32_t want = a.ticket0 + round;
        STAGE_STAMP(2);
        // Cross-block sum in ONE trip to the L2: every block adds the integer limb sums of its share to the round's 64-bit accumulators
        // (one 128-byte line each: the atomics of a value spread over L2 channels) with RETURNING adds that also count the
        // contributors in the accumulator's top bits (limb sums stay below 2^49; the count sits from bit 52).  The block whose add
        // arrives last at an accumulator -- the returned count says so; it may be a different block for every accumulator -- knows that
        // accumulator's total, forwards it to the host as one self-validating chunk and leaves a zero behind for the next launch.  The
        // host waits for all chunks of the round and reduces L + 2^32 H mod p (~0.1 us).  Sums of integers: the same field elements as
        // a chain of fr_add.  (Before: no-return adds, a drain, an arrival counter, and the last block swapping every accumulator out:
        // three dependent trips; before that, stored partials and a second pass: ~4.5 us per round.)
        unsigned long long* accb = a.d_acc + (size_t)round * (3 * 8 * 16);
        if (i < 16) {   // wave 0: lanes 0-7 the limbs of the evaluation-point sum, lanes 8-15 those of the tail weight
            const uint32_t sel = i >> 3, l = i & 7;
            if (sel == 0 || with_w) {
                unsigned long long v = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) v += (unsigned long long)half[w][sel][2 * l] + ((unsigned long long)half[w][sel][2 * l + 1] << 16);
                const uint32_t slot = sel ? 16u + l : (blockIdx.x & 1u) * 8u + l;
                // contributors: the blocks of this evaluation point (half of the reporting ones); the tail weight comes from segment 0's
                const uint32_t expected = sel ? nsl : (nrep >> 1);
                const unsigned long long prev = __hip_atomic_fetch_add(accb + slot * 16, v + (1ull << 52), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(prev >> 52) == expected - 1) {
                    const unsigned long long tot = (prev + v) & ((1ull << 52) - 1);
                    // report slot of this round: rewritten two rounds later, after the host has read it
                    chunk_store_sys(a.h_rep + 96 * (round & 1) + 4 * slot, (uint32_t)tot, (uint32_t)(tot >> 32), 0u, want);
                    __hip_atomic_store(accb + slot * 16, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        STAGE_STAMP(3);
        if (stage_np && i < stage_np) {   // off the critical path: the sums are on their way, the challenge is not back yet
#pragma unroll
            for (int q = 0; q < MAXIN; q++)
                if (q < g.n_in) { pre0[q][i] = p0[q]; pre1[q][i] = p1[q]; }
        }
        STAGE_STAMP(4);
        if (i == 0) {
            int good = 0;
            Fr t = fr_zero();
            const uint64_t t_begin = wall_clock64();
            if (a.d_tkt_bar) {
                // the host stores the challenge straight into device memory: every block polls it there -- no PCIe read, no relay hop
                const uint32_t* src = a.d_tkt_bar + 12 * (round & 1);
                for (uint32_t it = 0;; it++) {
                    if ((it & 255u) == 255u && wall_clock64() - t_begin > a.timeout_ticks) break;
                    if (fr_chunks_load<true>(src, &t, want, a.ticket0 + 0x4000u)) { good = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!good && blk == 0) __hip_atomic_store(a.h_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (blk == 0) {
                const uint32_t* src = a.h_tkt + 12 * (round & 1);
                for (uint32_t it = 0;; it++) {   // polling over PCIe until the bound: a slow (interpreted, traced) transcript is fine
                    if ((it & 255u) == 255u && wall_clock64() - t_begin > a.timeout_ticks) break;
                    if (fr_chunks_load<true>(src, &t, want, a.ticket0 + 0x4000u)) { good = 1; break; }   // ticket0 + 0x4000: the host lets every wait through
                    __builtin_amdgcn_s_sleep(1);
                }
                if (good) {
                    // slot round & 1 of the relay: a block still reading t_(r-1) is never overwritten (t_(r+1) needs every block's round-r+1 sum)
                    fr_chunks_store_dev(a.d_relay + 12 * (round & 1), t, want);
                } else {
                    __hip_atomic_store(a.h_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    fr_chunks_store_dev(a.d_relay + 12 * (round & 1), t, 0xffffffffu);   // let the others go
                }
            } else {
                const uint32_t* src = a.d_relay + 12 * (round & 1);
                for (uint32_t it = 0;; it++) {   // outlasts block 0's wait (4x); block 0 releases the others when it gives up
                    if ((it & 1023u) == 1023u && wall_clock64() - t_begin > 4 * a.timeout_ticks) break;
                    const int f = fr_chunks_load<false>(src, &t, want, 0xffffffffu);
                    if (f == 2) break;
                    if (f == 1) { good = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            ts = t;
            ok = good;
        }
        __syncthreads();
        STAGE_STAMP(5);
        return ok != 0;
    };

    uint32_t np;   // pairs this block holds in the dense phase
    // ------------------------------------------------------------------ thin phase: thread = row
    if (a.n_thin > 0) {
        const uint32_t r = slice * 256 + i;
        bool have = false;
        Fr coef = fr_zero();
        if (r < a.nrows) {
            const uint32_t c0 = a.off[r];
            have = a.off[r + 1] != c0;
            coef = fr_load(a.row_coef + r);
            if (have) {
#pragma unroll
                for (int q = 0; q < MAXIN; q++)
                    if (q < g.n_in) { p0[q] = fr_load(cols.p[g.in[q]] + c0); p1[q] = fr_load(cols.p[g.in[q]] + c0 + 1); }
            }
        }
        for (int tr = 0; tr < a.n_thin; tr++, round++) {
            STAGE_STAMP(0);
            // The weight of a row's pair in thin round tr is coef[r] e0, e0 = entry 0 of that round's eq level: ONE scalar per round.  The
            // host multiplies the round's sums by it (StageRun::thin_e0) -- on the device it was a product per row and round on every
            // block's critical path.  Likewise the tail weight W = sum_r coef[r] (1 - sum_{idx < seg_r} eq[idx]) (get_trailing_sum,
            // vecvec.rs:144-146) = C_all - e0 C_have with C_have = sum of coef over the rows that hold a pair -- the same rows in every
            // thin round (a folded row is re-padded to one pair): reported ONCE, in the launch's first round, with no product at all.
            Fr acc = fr_zero(), accw = fr_zero();
            if (have) acc = fr_mul_s(stage_eval<MAXIN>(g, p0, p1, gp, h), coef);
            const bool with_c = blockIdx.x == 0 && tr == 0;
            if (with_c && have) accw = coef;
            if (!exchange(acc, accw, with_c, nsl * gridDim.x)) return;
            if (have) {   // bind_21 on a row of one pair: [p0 + t (p1 - p0), row_pad]
                const Fr t = ts;
#pragma unroll
                for (int q = 0; q < MAXIN; q++)
                    if (q < g.n_in) { p0[q] = fr_add(p0[q], fr_mul_s(t, fr_sub(p1[q], p0[q]))); p1[q] = a.row_pad.v[g.in[q]]; }
            }
        }
        // bind_into_dense: the last fold above left the row's value in p0; absent cells are row_pad, absent rows col_pad
#pragma unroll
        for (int q = 0; q < MAXIN; q++)
            if (q < g.n_in) xch[q][i] = have ? p0[q] : (r < a.nrows ? a.row_pad.v[g.in[q]] : a.col_pad.v[g.in[q]]);
        __syncthreads();
        np = (a.n_elems < 256 ? a.n_elems : 256u) >> 1;
        if (i < np) {
#pragma unroll
            for (int q = 0; q < MAXIN; q++)
                if (q < g.n_in) { p0[q] = xch[q][2 * i]; p1[q] = xch[q][2 * i + 1]; }
        }
        __syncthreads();
    } else {
        np = (a.n_elems < 256 ? a.n_elems : 256u) >> 1;
        if (i < np) {
            const uint64_t gi = (uint64_t)slice * np + i;
#pragma unroll
            for (int q = 0; q < MAXIN; q++)
                if (q < g.n_in) { p0[q] = fr_load(cols.p[g.in[q]] + 2 * gi); p1[q] = fr_load(cols.p[g.in[q]] + 2 * gi + 1); }
        }
    }
    // ------------------------------------------------------------------ dense phase: thread = pair
    uint32_t my_slice = slice;   // 0 after the slices have merged
    bool merged = false;
    uint32_t n_left = np;        // elements this block holds after the last fold of the launch
    for (int dr = 0; dr < a.n_dense; dr++, round++) {
        STAGE_STAMP(0);
        Fr acc = fr_zero();
        if (i < np) acc = fr_mul_s(stage_eval<MAXIN>(g, p0, p1, gp, h), fr_load(a.eq[dr] + (uint64_t)my_slice * np + i));
        const bool pf = a.par_fold && np <= 128;
        stage_np = pf ? np : 0;
        if (!exchange(acc, fr_zero(), false, (merged || nsl == 1) ? gridDim.x : nsl * gridDim.x)) return;
        stage_np = 0;
        const Fr t = ts;
        if (pf) {
            // (the barrier at the end of exchange() orders the staging stores before these loads)
            const uint32_t nf = (uint32_t)g.n_in * np;
            for (uint32_t T = i; T < nf; T += 256) {
                const uint32_t q = T / np, j = T - q * np;
                const Fr a0 = pre0[q][j], a1 = pre1[q][j];
                xch[q][j] = fr_add(a0, fr_mul_s(t, fr_sub(a1, a0)));
            }
        } else if (i < np) {
#pragma unroll
            for (int q = 0; q < MAXIN; q++)
                if (q < g.n_in) xch[q][i] = fr_add(p0[q], fr_mul_s(t, fr_sub(p1[q], p0[q])));
        }
        __syncthreads();
        n_left = np;
        if (dr + 1 == a.n_dense) break;   // the folded elements stay in xch[q][0 .. n_left): exported below
        if (np > 1) {
            np >>= 1;
            if (i < np) {
#pragma unroll
                for (int q = 0; q < MAXIN; q++)
                    if (q < g.n_in) { p0[q] = xch[q][2 * i]; p1[q] = xch[q][2 * i + 1]; }
            }
            __syncthreads();
        } else {
            // one element per slice left: hand it to slice 0 (device memory, release / acquire around the arrival counter)
            Fr* xb = a.d_xbuf + (size_t)blockIdx.x * 6 * STAGE_MAX_SLICES;
            if (i < (uint32_t)g.n_in) {
                coh_store_dev(xb + (size_t)i * STAGE_MAX_SLICES + slice, xch[i][0]);
            }
            coh_drain();
            __syncthreads();
            if (i == 0) __hip_atomic_fetch_add(a.d_merge + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (slice != 0) return;
            if (i == 0) {
                int good = 0;
                const uint64_t t_begin = wall_clock64();
                for (uint32_t it = 0;; it++) {
                    if ((it & 1023u) == 1023u && wall_clock64() - t_begin > 4 * a.timeout_ticks) break;
                    if (__hip_atomic_load(a.d_merge + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nsl) { good = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!good) __hip_atomic_store(a.h_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else __hip_atomic_store(a.d_merge + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
                ok = good;
            }
            __syncthreads();
            if (!ok) return;
            my_slice = 0;
            merged = true;
            np = nsl >> 1;
            if (i < np) {
#pragma unroll
                for (int q = 0; q < MAXIN; q++)
                    if (q < g.n_in) {
                        // handed-over bytes: system-coherent loads (another CU wrote them)
                        const Fr* src = xb + (size_t)q * STAGE_MAX_SLICES + 2 * i;
                        p0[q] = coh_load_dev(src);
                        p1[q] = coh_load_dev(src + 1);
                    }
            }
            __syncthreads();
        }
    }
    // What is left of every input column of this segment goes to pinned host memory: the single final evaluation when the launch
    // ran every round, or the last <= 32 elements per column when the host finishes the sumcheck itself (rounds of <= 16 pairs
    // take the host a few microseconds each; here each costs a full round trip).  Element e of column c at h_finals[c * 32 + e].
    if (h == 0) {
        for (uint32_t e = i; e < n_left * (uint32_t)g.n_in; e += 256) {
            const uint32_t q = e / n_left, j = e % n_left;
            coh_store_sys(a.h_finals + (size_t)g.in[q] * 32 + (size_t)my_slice * n_left + j, xch[q][j]);
        }
        coh_drain();
        __syncthreads();
        if (i == 0) __hip_atomic_store(a.h_fin_seq + (size_t)sgi * STAGE_MAX_SLICES + my_slice, a.ticket0 + round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void __launch_bounds__(256) k_dense_fold_gated(ColPtrs in, ColPtrsMut out, uint64_t n_out, GateArgs g) {
    const Fr t = gated_challenge(g, nullptr);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const Fr* src = in.p[blockIdx.y];
    const Fr p0 = fr_load(src + 2 * i), p1 = fr_load(src + 2 * i + 1);
    fr_store(out.p[blockIdx.y] + i, fr_add(p0, fr_mul(t, fr_sub(p1, p0))));
}

// ------------------------------------------------------------------------------------------ lean large-round kernels
// The generic kernels above keep every segment's inputs, outputs and both evaluation points live at once (256 VGPRs plus
// AGPR spills: one wave per SIMD, ~45 G mul/s).  Large rounds of single-primitive layers -- every bintree layer and every
// gen-1 layer -- go through these instead: the primitive is a template parameter, each output is folded into the gamma
// combination as soon as it exists, and the evaluation points are walked by a rolled loop that reloads the pair (an L2
// hit), so the live state is one set of inputs.  Field sums are exact: same round polynomials, bit for bit.
#define LEAN_AFF_L1_BC 100  // StackedAlgFn(affine_l1, RepeatedAlgFn(BitCheckFn, 2)): the first bintree layer (bintree_add.rs:258-285)

__device__ __forceinline__ constexpr int lean_n_in(int prim) {
    return prim == FN_AFF_L1 ? 4 : prim == FN_AFF_L2 ? 3 : prim == FN_AFF_L3 ? 3 : prim == FN_PROJ_L1 ? 6 : prim == FN_PROJ_L2 ? 4
         : prim == FN_PROJ_L3 ? 4 : prim == FN_PT_BIT_CHOICE ? 3 : prim == LEAN_AFF_L1_BC ? 6 : prim == FN_ADD_INVERSES ? 2
         : prim == FN_LOGUP_LAYER ? 4 : 0;
}

// sum_o gamma^o f_o(v) for the whole (single-primitive) layer function; g[o] = gamma^o on the device
template <int PRIM>
__device__ __forceinline__ Fr lean_gamma_eval(const Fr* v, const Fr* __restrict__ g) {
    // the same re-association as lean_gamma_eval9 (every gamma power multiplies an input once): exact identities
    if (PRIM == FN_AFF_L1 || PRIM == LEAN_AFF_L1_BC) {
        const Fr g2 = fr_load(g + 2);
        const Fr g2v2 = fr_mul(g2, v[2]);
        const Fr t1 = fr_add(v[3], fr_add(fr_dbl(fr_dbl(g2v2)), g2v2));                   // v3 + 5 g2 v2   (-a = 5)
        const Fr t2 = fr_add(fr_mul(fr_load(g + 1), v[2]), fr_mul(g2, v[3]));
        Fr A = fr_add(fr_mul(v[0], t1), fr_mul(v[1], t2));
        if (PRIM == LEAN_AFF_L1_BC) {
            A = fr_add(A, fr_mul(fr_load(g + 3), fr_sub(fr_sqr(v[4]), v[4])));
            A = fr_add(A, fr_mul(fr_load(g + 4), fr_sub(fr_sqr(v[5]), v[5])));
        }
        return A;
    } else if (PRIM == FN_AFF_L2) {
        Fr A = fr_add(v[0], v[1]);
        A = fr_add(A, fr_mul(fr_load(g + 1), v[2]));
        return fr_add(A, fr_mul(fr_mul(fr_load(g + 2), v[0]), v[1]));
    } else if (PRIM == FN_AFF_L3 || PRIM == FN_PROJ_L3) {
        const Fr dxy = fr_mul_by_d(v[PRIM == FN_AFF_L3 ? 2 : 3]);
        const Fr base = PRIM == FN_AFF_L3 ? fr_one() : v[2];
        const Fr m = fr_sub(base, dxy), q = fr_add(base, dxy);
        const Fr A = fr_mul(m, fr_add(v[0], fr_mul(fr_load(g + 2), q)));
        return fr_add(A, fr_mul(fr_load(g + 1), fr_mul(q, v[1])));
    } else if (PRIM == FN_PROJ_L1) {
        const Fr g2 = fr_load(g + 2);
        const Fr g2v3 = fr_mul(g2, v[3]);
        const Fr t1 = fr_add(v[4], fr_add(fr_dbl(fr_dbl(g2v3)), g2v3));                   // v4 + 5 g2 v3
        const Fr t2 = fr_add(fr_mul(fr_load(g + 1), v[3]), fr_mul(g2, v[4]));
        const Fr A = fr_add(fr_mul(v[0], t1), fr_mul(v[1], t2));
        return fr_add(A, fr_mul(fr_load(g + 3), fr_mul(v[2], v[5])));
    } else if (PRIM == FN_PROJ_L2) {
        const Fr u = fr_add(fr_add(v[0], v[1]), fr_add(fr_mul(fr_load(g + 1), v[2]), fr_mul(fr_load(g + 2), v[3])));
        return fr_add(fr_mul(v[3], u), fr_mul(fr_load(g + 3), fr_mul(v[0], v[1])));
    } else if (PRIM == FN_ADD_INVERSES) {
        return fr_add(fr_add(v[0], v[1]), fr_mul(fr_mul(fr_load(g + 1), v[0]), v[1]));
    } else if (PRIM == FN_LOGUP_LAYER) {
        return fr_add(fr_mul(v[3], fr_add(v[0], fr_mul(fr_load(g + 1), v[1]))), fr_mul(v[1], v[2]));
    } else {  // FN_PT_BIT_CHOICE: (b, x, y) -> (b x, b (y - 1) + 1): b (x + g1 (y - 1)) + g1
        const Fr g1 = fr_load(g + 1);
        return fr_add(fr_mul(v[0], fr_add(v[1], fr_mul(g1, fr_sub(v[2], fr_one())))), g1);
    }
}

struct LeanCols {
    const Fr* p[7];  // inputs of the primitive in order (+ the eq column for the generic object)
};

// deg-2 round sums with eq factored out (same contract as k_round_deg2<VECVEC, false>)
template <int PRIM, bool VECVEC>
__global__ void __launch_bounds__(SC_THREADS) k_round_deg2_lean(LeanCols cols, const Fr* __restrict__ eq, const Fr* __restrict__ gp,
                                                                 uint64_t npairs_dense, VVArgs vv, FinishCtx fc) {
    constexpr int NACC = VECVEC ? 3 : 2;
    constexpr int NI = lean_n_in(PRIM);
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    if (VECVEC) {
        for (uint32_t r = blockIdx.x * SC_THREADS + threadIdx.x; r < vv.nrows; r += gridDim.x * SC_THREADS) {
            const uint32_t seg = (vv.off[r + 1] - vv.off[r]) >> 1;
            acc[2] = fr_add(acc[2], fr_mul(fr_load(vv.row_coef + r), fr_sub(fr_one(), fr_load(vv.eq_prefix + seg))));
        }
    }
    const uint64_t npairs = VECVEC ? (uint64_t)(vv.off[vv.nrows] >> 1) : npairs_dense;
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS) {
        Fr w;
        if (VECVEC) {
            const uint32_t cell0 = (uint32_t)(2 * i);
            // 13 dependent loads of a full binary search per pair stall the few resident waves; the coarse table brackets the
            // row to the rows that intersect one 256-cell block (one or two for long rows)
            const uint32_t r = vv.coarse ? find_row_coarse(vv.off, vv.nrows, vv.coarse, cell0) : find_row(vv.off, vv.nrows, cell0);
            w = fr_mul(fr_load(eq + ((cell0 - vv.off[r]) >> 1)), fr_load(vv.row_coef + r));
        } else {
            w = fr_load(eq + i);
        }
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            Fr v[NI];
#pragma unroll
            for (int q = 0; q < NI; q++) {
                const Fr p1 = fr_load(cols.p[q] + 2 * i + 1);
                v[q] = h ? fr_sub(fr_dbl(p1), fr_load(cols.p[q] + 2 * i)) : p1;
            }
            const Fr t = fr_mul(lean_gamma_eval<PRIM>(v, gp), w);
            if (h == 0) acc[0] = fr_add(acc[0], t); else acc[1] = fr_add(acc[1], t);
        }
    }
    block_reduce_finish<NACC>(acc, fc);
}

// The medium sparse rounds of a single-primitive layer (<= 2^14 pairs: latency-bound, one (pair, evaluation point) per thread as
// k_round_deg2<true, true>, blockIdx.y = the point) with the re-associated gamma combination instead of prim_exec + one product per
// output: PROJ_L1 7 + 2 multiplications on a thread's critical path instead of 9 + 3 + 2.
template <int PRIM>
__global__ void __launch_bounds__(SC_THREADS) k_round_deg2_lean_split(LeanCols cols, const Fr* __restrict__ eq, const Fr* __restrict__ gp, VVArgs vv,
                                                                       FinishCtx fc) {
    constexpr int NI = lean_n_in(PRIM);
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    if (blockIdx.y == 0) {
        for (uint32_t r = blockIdx.x * SC_THREADS + threadIdx.x; r < vv.nrows; r += gridDim.x * SC_THREADS) {
            const uint32_t seg = (vv.off[r + 1] - vv.off[r]) >> 1;
            acc[2] = fr_add(acc[2], fr_mul(fr_load(vv.row_coef + r), fr_sub(fr_one(), fr_load(vv.eq_prefix + seg))));
        }
    }
    const uint64_t npairs = (uint64_t)(vv.off[vv.nrows] >> 1);
    const int h = blockIdx.y & 1;
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS) {
        const uint32_t cell0 = (uint32_t)(2 * i);
        const uint32_t r = vv.coarse ? find_row_coarse(vv.off, vv.nrows, vv.coarse, cell0) : find_row(vv.off, vv.nrows, cell0);
        const Fr w = fr_mul(fr_load(eq + ((cell0 - vv.off[r]) >> 1)), fr_load(vv.row_coef + r));
        Fr v[NI];
#pragma unroll
        for (int q = 0; q < NI; q++) {
            const Fr p1 = fr_load(cols.p[q] + 2 * i + 1);
            v[q] = h ? fr_sub(fr_dbl(p1), fr_load(cols.p[q] + 2 * i)) : p1;
        }
        acc[h] = fr_add(acc[h], fr_mul(lean_gamma_eval<PRIM>(v, gp), w));
    }
    block_reduce_finish<3>(acc, fc);
}

// ---- the same large rounds in the 9 x 29-bit form (fr9.hip.h).  Inputs are loaded raw (the limbs of the stored value: domain
// 256, S = 1 -- no conversion); every term of these layer functions is a product of exactly two inputs, so all of them land in
// domain 251, the gamma powers are loaded in domain 261 (shifted loads: g x term stays in 251), the weight brings the
// accumulators to domain 241 (VecVec: eq x coef) or 246 (dense), and ONE multiplication by 2^276 / 2^271 per thread at the end
// returns to the stored form.  Same field values as k_round_deg2_lean, bit for bit (the sums are canonicalised before the block
// reduction).  Bounds per line: L = limb bound, S = value / p (2^261 / p = 70.66).
// Measured at config B (bench.py, ms per proof over the large launches, 9 x 29 vs 8 x 32): PROJ_L1 6.03 vs 6.25, PROJ_L2 5.10 vs
// 5.22, PROJ_L3 4.65 vs 5.12, AFF_L1+BITCHECK 4.42 vs 5.13, AFF_L3 2.88 vs 3.26; gen-1 at 2^20 points: 353 vs 368 ms.
// GM_LEAN_FR9=0 switches the form off (A/B measurements).
__host__ __device__ constexpr bool lean9_has(int prim) {
    return prim == FN_PROJ_L1 || prim == FN_PROJ_L2 || prim == FN_PROJ_L3 || prim == FN_AFF_L1 || prim == LEAN_AFF_L1_BC || prim == FN_AFF_L3 ||
           prim == FN_AFF_L2 || prim == FN_PT_BIT_CHOICE || prim == FN_ADD_INVERSES || prim == FN_LOGUP_LAYER;
}
// AFF_L2 and ADD_INVERSES have terms of degree one (a + b): those stay in domain 256, and their product term is brought there by
// taking ONE factor from a shifted load (domain 261): (g v0) v1s is 261 + 256 - 261 = 256, times 261, minus 261 = 256.  Their
// accumulators therefore sit five binary places higher than the others'.
__host__ __device__ constexpr bool lean9_terms_256(int prim) { return prim == FN_AFF_L2 || prim == FN_ADD_INVERSES; }
// ld(q): input q at the evaluation point -- L 2^29, S <= 10, domain 256; loaded (and, at the second point, formed from the pair)
// when the formula first needs it, in an order that keeps at most three inputs live: nine registers per value is what pushed the
// six-input primitives to 256 VGPRs when all inputs were loaded up front.  Result: domain 251, L <= 5 2^29, S <= 30.
// lds(q): the same input from shifted loads (domain 261, L 2^29 with a top limb below 2^29.9, S <= 128)
// Both evaluation points of a pair side by side (a: the point "1" = p1, b: the point "2" = 2 p1 - p0): every operation of the layer
// function is applied to both, the two products of a step as ONE interleaved instruction stream (fr9_mul2: the multiply-add that
// consumes its predecessor's result costs a wait state, alternating two independent chains hides every one of them).  A pair is
// then loaded ONCE per round kernel instead of once per evaluation point (profiles/r02: 317 MB of HBM traffic per launch for 155 MB
// of pairs).  Limb / value bounds per operation are those of the scalar helpers, component-wise.
struct Fr9x2 {
    Fr9 a, b;
    __device__ __forceinline__ Fr9x2() {}
    __device__ __forceinline__ Fr9x2(const Fr9& x, const Fr9& y) : a(x), b(y) {}
    __device__ __forceinline__ explicit Fr9x2(const Fr9& c) : a(c), b(c) {}
};
__device__ __forceinline__ Fr9x2 fr9_mul(const Fr9x2& x, const Fr9x2& y) { Fr9x2 r; fr9_mul2(x.a, y.a, x.b, y.b, r.a, r.b); return r; }
__device__ __forceinline__ Fr9x2 fr9_mul(const Fr9& c, const Fr9x2& y) { Fr9x2 r; fr9_mul2(c, y.a, c, y.b, r.a, r.b); return r; }
__device__ __forceinline__ Fr9x2 fr9_mul(const Fr9x2& x, const Fr9& c) { Fr9x2 r; fr9_mul2(x.a, c, x.b, c, r.a, r.b); return r; }
__device__ __forceinline__ Fr9x2 fr9_sqr(const Fr9x2& x) { return Fr9x2(fr9_sqr(x.a), fr9_sqr(x.b)); }
__device__ __forceinline__ Fr9x2 fr9_add(const Fr9x2& x, const Fr9x2& y) { return Fr9x2(fr9_add(x.a, y.a), fr9_add(x.b, y.b)); }
__device__ __forceinline__ Fr9x2 fr9_add(const Fr9x2& x, const Fr9& c) { return Fr9x2(fr9_add(x.a, c), fr9_add(x.b, c)); }
__device__ __forceinline__ Fr9x2 fr9_add(const Fr9& c, const Fr9x2& x) { return Fr9x2(fr9_add(c, x.a), fr9_add(c, x.b)); }
__device__ __forceinline__ Fr9x2 fr9_mul5(const Fr9x2& x) { return Fr9x2(fr9_mul5(x.a), fr9_mul5(x.b)); }
__device__ __forceinline__ Fr9x2 fr9_norm(const Fr9x2& x) { return Fr9x2(fr9_norm(x.a), fr9_norm(x.b)); }
__device__ __forceinline__ Fr9x2 fr9_sub8(const Fr9x2& x, const Fr9x2& y) { return Fr9x2(fr9_sub8(x.a, y.a), fr9_sub8(x.b, y.b)); }
__device__ __forceinline__ Fr9x2 fr9_sub8(const Fr9x2& x, const Fr9& c) { return Fr9x2(fr9_sub8(x.a, c), fr9_sub8(x.b, c)); }
__device__ __forceinline__ Fr9x2 fr9_sub8(const Fr9& c, const Fr9x2& x) { return Fr9x2(fr9_sub8(c, x.a), fr9_sub8(c, x.b)); }

// Round 3: the gamma combination is re-associated so that every gamma power multiplies an INPUT once and the products of two inputs
// are shared -- e.g. PROJ_L1 = v0 (v4 + 5 g2 v3) + v1 (g1 v3 + g2 v4) + g3 v2 v5: 7 multiplications instead of 8, PROJ_L2 =
// v3 (v0 + v1 + g1 v2 + g2 v3) + g3 v0 v1: 5 instead of 7.  Exact field identities: the same round sums, bit for bit.
template <int PRIM, typename LD, typename LDS>
__device__ __forceinline__ auto lean_gamma_eval9(const LD& ld, const LDS& lds, const Fr* __restrict__ g) -> decltype(ld(0)) {
    typedef decltype(ld(0)) V;   // Fr9: one evaluation point; Fr9x2: both points of the pair side by side (k_round_deg2_lean9x2)
    // bounds: ld(q) L 2^29, S <= 10, domain 256; a gamma power g_k = fr9_load(g + k) L 2^29, S 32, domain 261; g_k x input: domain 256,
    // S 32 x 10 / 70.66 + 1 = 5.53; a product needs 9 L_a L_b + 2^62 < 2^64, i.e. L_a L_b <= 2^60.4
    if (PRIM == FN_ADD_INVERSES) {
        // v0 + v1 + g1 v0 v1, every term in domain 256
        const V v0 = ld(0);
        const V t = fr9_mul(fr9_mul(fr9_load(g + 1), v0), lds(1));                     // S 5.5, then 5.5 x 128 / 70.66 + 1 = 11
        return fr9_add(fr9_add(v0, ld(1)), t);                                           // L 3 2^29, S 31
    } else if (PRIM == FN_AFF_L2) {
        // v0 + v1 + g1 v2 + g2 v0 v1, every term in domain 256
        const V v0 = ld(0);
        V A = fr9_mul(fr9_mul(fr9_load(g + 2), v0), lds(1));                           // S 11
        A = fr9_add(A, fr9_mul(fr9_load(g + 1), ld(2)));                                 // S 5.5
        return fr9_add(fr9_add(A, v0), ld(1));                                           // L 4 2^29, S 36.5
    } else if (PRIM == FN_PT_BIT_CHOICE) {
        // b x + g1 (b (y - 1) + 1) = b (x + g1 (y - 1)) + g1: two products per point; g1 in domain 251 is loop-invariant
        const Fr9 g1 = fr9_load(g + 1);
        const V ym1 = fr9_norm(fr9_sub8(ld(2), fr9_one256()));                            // S 18, domain 256
        const V u = fr9_add(ld(1), fr9_mul(g1, ym1));                                    // g1 (y - 1): S 9.2; sum L 2^30, S 19.2
        return fr9_add(fr9_mul(ld(0), u), fr9_mul(g1, fr9_one251()));                    // S 3.7 + 1.0: L 2 2^29, domain 251
    } else if (PRIM == FN_LOGUP_LAYER) {
        // a d + b c + g1 b d = d (a + g1 b) + b c: three products
        const V v1 = ld(1);
        const V u = fr9_add(ld(0), fr9_mul(fr9_load(g + 1), v1));                        // L 2^30, S 15.5, domain 256
        return fr9_add(fr9_mul(ld(3), u), fr9_mul(v1, ld(2)));                           // S 3.2 + 2.4: L 2 2^29, domain 251
    } else if (PRIM == FN_AFF_L1 || PRIM == LEAN_AFF_L1_BC) {
        // v0 v3 + g1 v2 v1 + g2 (v1 v3 + 5 v0 v2) = v0 (v3 + 5 g2 v2) + v1 (g1 v2 + g2 v3) [+ g3 (v4^2 - v4) + g4 (v5^2 - v5)]
        V A;
        {
            const V v2 = ld(2), v3 = ld(3);
            const Fr9 g2 = fr9_load(g + 2);
            const V t1 = fr9_norm(fr9_add(v3, fr9_mul5(fr9_mul(g2, v2))));               // 5 x 5.53 + 10: S 37.7, normalised: L 2^29
            const V t2 = fr9_add(fr9_mul(fr9_load(g + 1), v2), fr9_mul(g2, v3));         // L 2^30, S 11.1
            A = fr9_mul(ld(0), t1);                                                       // S 6.3, domain 251
            A = fr9_add(A, fr9_mul(ld(1), t2));                                           // S 2.6: L 2 2^29, S 8.9
        }
        if (PRIM == LEAN_AFF_L1_BC) {
            // b^2 - b = b (b - 1): (b - 1 + 8 p) normalised has S 18, the product S 3.5, times gamma S 2.6
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const V b = ld(4 + k);
                A = fr9_add(A, fr9_mul(fr9_load(g + 3 + k), fr9_mul(b, fr9_norm(fr9_sub8(b, fr9_one256())))));
            }
        }
        return A;                                                                         // L <= 4 2^29, S <= 14.1
    } else if (PRIM == FN_AFF_L3 || PRIM == FN_PROJ_L3) {
        // m v0 + g1 q v1 + g2 m q = m (v0 + g2 q) + g1 (q v1),  m = base - d xy, q = base + d xy
        const V dxy = fr9_mul(ld(PRIM == FN_AFF_L3 ? 2 : 3), fr9_coeff_d());             // d in domain 261: dxy in 256, S 1.14
        const V base = PRIM == FN_AFF_L3 ? V(fr9_one256()) : ld(2);
        const V m = fr9_norm(fr9_sub8(base, dxy));                                        // S 18
        const V q = fr9_add(base, dxy);                                                   // L 2^30, S 11.2
        const V u = fr9_add(ld(0), fr9_mul(fr9_load(g + 2), q));                          // g2 q: 2^29 x 2^30, S 6.1, domain 256; sum L 2^30, S 16.1
        V A = fr9_mul(m, u);                                                              // S 5.1, domain 251
        return fr9_add(A, fr9_mul(fr9_load(g + 1), fr9_mul(q, ld(1))));                   // q v1: S 2.6 -> 2.2; L 2 2^29, S 7.3
    } else if (PRIM == FN_PROJ_L1) {
        // v0 v4 + g1 v3 v1 + g2 (v1 v4 + 5 v0 v3) + g3 v2 v5 = v0 (v4 + 5 g2 v3) + v1 (g1 v3 + g2 v4) + g3 v2 v5
        V A;
        {
            const V v3 = ld(3), v4 = ld(4);
            const Fr9 g2 = fr9_load(g + 2);
            const V t1 = fr9_norm(fr9_add(v4, fr9_mul5(fr9_mul(g2, v3))));               // S 37.7, normalised: L 2^29
            const V t2 = fr9_add(fr9_mul(fr9_load(g + 1), v3), fr9_mul(g2, v4));         // L 2^30, S 11.1
            A = fr9_mul(ld(0), t1);                                                       // S 6.3, domain 251
            A = fr9_add(A, fr9_mul(ld(1), t2));                                           // S 2.6
        }
        return fr9_add(A, fr9_mul(fr9_load(g + 3), fr9_mul(ld(2), ld(5))));               // S 2.1: L 3 2^29, S 11
    } else {  // FN_PROJ_L2: (v0 + v1) v3 + g1 v2 v3 + g2 v3^2 + g3 v0 v1 = v3 (v0 + v1 + g1 v2 + g2 v3) + g3 v0 v1
        const V v0 = ld(0), v1 = ld(1);
        V A;
        {
            const V v3 = ld(3);
            const V u = fr9_add(fr9_add(v0, v1), fr9_add(fr9_mul(fr9_load(g + 1), ld(2)), fr9_mul(fr9_load(g + 2), v3)));   // L 4 2^29 = 2^31, S 31.1
            A = fr9_mul(v3, u);                                                           // 9 x 2^29 x 2^31 + 2^62 = 2^63.7; S 5.4, domain 251
        }
        return fr9_add(A, fr9_mul(fr9_load(g + 3), fr9_mul(v0, v1)));                     // S 2.1: L 2 2^29, S 7.5
    }
}

template <int PRIM, bool VECVEC>
__global__ void __launch_bounds__(SC_THREADS, 3) k_round_deg2_lean9(LeanCols cols, const Fr* __restrict__ eq, const Fr* __restrict__ gp,
                                                                  uint64_t npairs_dense, VVArgs vv, FinishCtx fc) {
    constexpr int NACC = VECVEC ? 3 : 2;
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    if (VECVEC) {
        for (uint32_t r = blockIdx.x * SC_THREADS + threadIdx.x; r < vv.nrows; r += gridDim.x * SC_THREADS) {
            const uint32_t seg = (vv.off[r + 1] - vv.off[r]) >> 1;
            acc[2] = fr_add(acc[2], fr_mul(fr_load(vv.row_coef + r), fr_sub(fr_one(), fr_load(vv.eq_prefix + seg))));
        }
    }
    Fr9 a0 = fr9_zero(), a1 = fr9_zero();   // domain 241 (VecVec) / 246 (dense); normalised, S grows by <= 1.5 per pair
    uint32_t it = 0;
    const uint64_t npairs = VECVEC ? (uint64_t)(vv.off[vv.nrows] >> 1) : npairs_dense;
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS, it++) {
        Fr9 w;
        if (VECVEC) {
            const uint32_t cell0 = (uint32_t)(2 * i);
            const uint32_t r = vv.coarse ? find_row_coarse(vv.off, vv.nrows, vv.coarse, cell0) : find_row(vv.off, vv.nrows, cell0);
            w = fr9_mul(fr9_load_raw(eq + ((cell0 - vv.off[r]) >> 1)), fr9_load_raw(vv.row_coef + r));   // domain 251, S 1.02
        } else {
            w = fr9_load_raw(eq + i);                                                                      // domain 256, S 1
        }
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            auto ld = [&](int q) -> Fr9 {
                const Fr9 p1 = fr9_load_raw(cols.p[q] + 2 * i + 1);
                if (!h) return p1;
                const Fr9 p0 = fr9_load_raw(cols.p[q] + 2 * i);
                return fr9_norm(fr9_sub8(fr9_add(p1, p1), p0));         // 2 p1 - p0 + 8 p: S 10
            };
            auto lds = [&](int q) -> Fr9 {
                const Fr9 p1 = fr9_load(cols.p[q] + 2 * i + 1);         // the limbs of 32 X: domain 261, S 32
                if (!h) return p1;
                const Fr9 p0 = fr9_load(cols.p[q] + 2 * i);
                return fr9_norm(fr9_sub64(fr9_add(p1, p1), p0));        // 2 p1 - p0 + 64 p: S 128, top limb < 2^29.9
            };
            const Fr9 t = fr9_mul(lean_gamma_eval9<PRIM>(ld, lds, gp), w);   // L <= 5 2^29 x 2^29; S <= 37 x 1.02 / 70.66 + 1 = 1.6
            if (h == 0) a0 = fr9_norm(fr9_add(a0, t)); else a1 = fr9_norm(fr9_add(a1, t));
        }
        if ((it & 15u) == 15u) {   // S <= 16 x 3 + 2: back below 2 (times one in domain 261 keeps the domain)
            a0 = fr9_mul(a0, fr9_one());
            a1 = fr9_mul(a1, fr9_one());
        }
    }
    // back to the stored form: domain 241 / 246 times 2^276 / 2^271 (domain-free integers) = domain 256; S <= 50 / 70.66 + 1 < 2
    // (terms in domain 256 -- lean9_terms_256 -- leave the accumulators five places higher: 2^271 / 2^266)
    const Fr9 K = lean9_terms_256(PRIM) ? (VECVEC ? fr9_two271() : fr9_two266()) : (VECVEC ? fr9_two276() : fr9_two271());
    acc[0] = fr9_to_raw(fr9_mul(a0, K));
    acc[1] = fr9_to_raw(fr9_mul(a1, K));
    block_reduce_finish<NACC>(acc, fc);
}

// The same round sums with every pair loaded ONCE: both evaluation points go through the layer function side by side (Fr9x2).
// Twice the live values of k_round_deg2_lean9 (two waves per SIMD instead of three), in exchange for half the loads and two
// interleaved multiplier chains.  Same field values, bit for bit.  GM_LEAN_X2=0 goes back to the one-point-at-a-time kernel.
template <int PRIM, bool VECVEC>
__global__ void __launch_bounds__(SC_THREADS, 2) k_round_deg2_lean9x2(LeanCols cols, const Fr* __restrict__ eq, const Fr* __restrict__ gp,
                                                                    uint64_t npairs_dense, VVArgs vv, FinishCtx fc) {
    constexpr int NACC = VECVEC ? 3 : 2;
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    if (VECVEC) {
        for (uint32_t r = blockIdx.x * SC_THREADS + threadIdx.x; r < vv.nrows; r += gridDim.x * SC_THREADS) {
            const uint32_t seg = (vv.off[r + 1] - vv.off[r]) >> 1;
            acc[2] = fr_add(acc[2], fr_mul(fr_load(vv.row_coef + r), fr_sub(fr_one(), fr_load(vv.eq_prefix + seg))));
        }
    }
    Fr9x2 a = Fr9x2(fr9_zero());   // domain 241 (VecVec) / 246 (dense); normalised, S grows by <= 1.5 per pair
    uint32_t it = 0;
    const uint64_t npairs = VECVEC ? (uint64_t)(vv.off[vv.nrows] >> 1) : npairs_dense;
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS, it++) {
        Fr9 w;
        if (VECVEC) {
            const uint32_t cell0 = (uint32_t)(2 * i);
            const uint32_t r = vv.coarse ? find_row_coarse(vv.off, vv.nrows, vv.coarse, cell0) : find_row(vv.off, vv.nrows, cell0);
            w = fr9_mul(fr9_load_raw(eq + ((cell0 - vv.off[r]) >> 1)), fr9_load_raw(vv.row_coef + r));   // domain 251, S 1.02
        } else {
            w = fr9_load_raw(eq + i);                                                                      // domain 256, S 1
        }
        auto ld = [&](int q) -> Fr9x2 {
            const Fr9 p0 = fr9_load_raw(cols.p[q] + 2 * i), p1 = fr9_load_raw(cols.p[q] + 2 * i + 1);
            return Fr9x2(p1, fr9_norm(fr9_sub8(fr9_add(p1, p1), p0)));      // p1 | 2 p1 - p0 + 8 p: S 10
        };
        auto lds = [&](int q) -> Fr9x2 {
            const Fr9 p0 = fr9_load(cols.p[q] + 2 * i), p1 = fr9_load(cols.p[q] + 2 * i + 1);   // the limbs of 32 X: domain 261, S 32
            return Fr9x2(p1, fr9_norm(fr9_sub64(fr9_add(p1, p1), p0)));    // 2 p1 - p0 + 64 p: S 128, top limb < 2^29.9
        };
        const Fr9x2 t = fr9_mul(lean_gamma_eval9<PRIM>(ld, lds, gp), w);   // L <= 5 2^29 x 2^29; S <= 37 x 1.02 / 70.66 + 1 = 1.6
        a = fr9_norm(fr9_add(a, t));
        if ((it & 15u) == 15u) a = fr9_mul(a, fr9_one());   // S <= 16 x 3 + 2: back below 2 (times one in domain 261 keeps the domain)
    }
    const Fr9 K = lean9_terms_256(PRIM) ? (VECVEC ? fr9_two271() : fr9_two266()) : (VECVEC ? fr9_two276() : fr9_two271());
    a = fr9_mul(a, K);
    acc[0] = fr9_to_raw(a.a);
    acc[1] = fr9_to_raw(a.b);
    block_reduce_finish<NACC>(acc, fc);
}

// The medium sparse rounds (<= 2^14 pairs: latency-bound) of a single-primitive layer in the same form: one (pair, evaluation point) per
// thread, blockIdx.y = the point; the tail weight (a loop over the ROWS) runs in workgroup row y = 2 of its own instead of ahead of the
// pairs in row 0, whose threads were the launch's critical path.  Same sums as k_round_deg2_lean_split, bit for bit.
template <int PRIM>
__global__ void __launch_bounds__(SC_THREADS) k_round_deg2_lean9_split(LeanCols cols, const Fr* __restrict__ eq, const Fr* __restrict__ gp,
                                                                        VVArgs vv, FinishCtx fc) {
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    if (blockIdx.y == 2) {
        for (uint32_t r = blockIdx.x * SC_THREADS + threadIdx.x; r < vv.nrows; r += gridDim.x * SC_THREADS) {
            const uint32_t seg = (vv.off[r + 1] - vv.off[r]) >> 1;
            acc[2] = fr_add(acc[2], fr_mul(fr_load(vv.row_coef + r), fr_sub(fr_one(), fr_load(vv.eq_prefix + seg))));
        }
        block_reduce_finish<3>(acc, fc);
        return;
    }
    const bool h = blockIdx.y == 1;
    Fr9 a = fr9_zero();   // domain 241; normalised, S grows by <= 1.5 per pair
    uint32_t it = 0;
    const uint64_t npairs = (uint64_t)(vv.off[vv.nrows] >> 1);
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS, it++) {
        const uint32_t cell0 = (uint32_t)(2 * i);
        const uint32_t r = vv.coarse ? find_row_coarse(vv.off, vv.nrows, vv.coarse, cell0) : find_row(vv.off, vv.nrows, cell0);
        const Fr9 w = fr9_mul(fr9_load_raw(eq + ((cell0 - vv.off[r]) >> 1)), fr9_load_raw(vv.row_coef + r));   // domain 251, S 1.02
        auto ld = [&](int q) -> Fr9 {
            const Fr9 p1 = fr9_load_raw(cols.p[q] + 2 * i + 1);
            if (!h) return p1;
            return fr9_norm(fr9_sub8(fr9_add(p1, p1), fr9_load_raw(cols.p[q] + 2 * i)));   // 2 p1 - p0 + 8 p: S 10
        };
        auto lds = [&](int q) -> Fr9 {
            const Fr9 p1 = fr9_load(cols.p[q] + 2 * i + 1);                                 // the limbs of 32 X: domain 261, S 32
            if (!h) return p1;
            return fr9_norm(fr9_sub64(fr9_add(p1, p1), fr9_load(cols.p[q] + 2 * i)));      // S 128, top limb < 2^29.9
        };
        const Fr9 t = fr9_mul(lean_gamma_eval9<PRIM>(ld, lds, gp), w);
        a = fr9_norm(fr9_add(a, t));
        if ((it & 15u) == 15u) a = fr9_mul(a, fr9_one());
    }
    a = fr9_mul(a, lean9_terms_256(PRIM) ? fr9_two271() : fr9_two276());
    acc[h ? 1 : 0] = fr9_to_raw(a);
    block_reduce_finish<3>(acc, fc);
}

// generic degree-3 round of F = eq * GammaWrapper(f) (same contract as k_round_generic<3, false>, kind 0); cols.p[NI] = eq
template <int PRIM>
__global__ void __launch_bounds__(SC_THREADS) k_round_generic3_lean(LeanCols cols, const Fr* __restrict__ gp, uint64_t npairs,
                                                                     FinishCtx fc) {
    constexpr int NI = lean_n_in(PRIM);
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS) {
#pragma unroll 1
        for (int s = 0; s < 3; s++) {
            // argument at evaluation point s + 1:  p1 + s (p1 - p0)
            Fr v[NI + 1];
#pragma unroll
            for (int q = 0; q <= NI; q++) {
                const Fr p1 = fr_load(cols.p[q] + 2 * i + 1);
                if (s == 0) v[q] = p1;
                else {
                    const Fr d = fr_sub(p1, fr_load(cols.p[q] + 2 * i));
                    v[q] = fr_add(p1, s == 1 ? d : fr_dbl(d));
                }
            }
            const Fr t = fr_mul(lean_gamma_eval<PRIM>(v, gp), v[NI]);
            if (s == 0) acc[0] = fr_add(acc[0], t);
            else if (s == 1) acc[1] = fr_add(acc[1], t);
            else acc[2] = fr_add(acc[2], t);
        }
    }
    block_reduce_finish<3>(acc, fc);
}

// Prod3Fn rounds (pushforward.rs:27-49; the combined sumcheck of the pushforward argument): acc[s] += a b c at p1 + s (p1 - p0)
__global__ void __launch_bounds__(SC_THREADS) k_round_prod3_lean(LeanCols cols, uint64_t npairs, FinishCtx fc) {
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS) {
#pragma unroll 1
        for (int s = 0; s < 3; s++) {
            Fr v[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const Fr p1 = fr_load(cols.p[q] + 2 * i + 1);
                if (s == 0) v[q] = p1;
                else {
                    const Fr d = fr_sub(p1, fr_load(cols.p[q] + 2 * i));
                    v[q] = fr_add(p1, s == 1 ? d : fr_dbl(d));
                }
            }
            const Fr t = fr_mul(fr_mul(v[0], v[1]), v[2]);
            if (s == 0) acc[0] = fr_add(acc[0], t);
            else if (s == 1) acc[1] = fr_add(acc[1], t);
            else acc[2] = fr_add(acc[2], t);
        }
    }
    block_reduce_finish<3>(acc, fc);
}

// FoldedProdAlgFn rounds (multiopen_reduction.rs:13-42): F = sum_i gamma^i a_i e_i over nargs (polynomial, eq table) pairs,
// degree 2: acc[s] += F(p1 + s (p1 - p0)), s = 0, 1.  cols.p[i] = a_i, cols.p[nargs + i] = e_i.
struct FoldedCols {
    const Fr* p[16];
};
__global__ void __launch_bounds__(SC_THREADS) k_round_folded_prod(FoldedCols cols, int nargs, const Fr* __restrict__ gp, uint64_t npairs,
                                                                   FinishCtx fc) {
    Fr acc[2] = {fr_zero(), fr_zero()};
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS) {
#pragma unroll 1
        for (int q = 0; q < nargs; q++) {
            const Fr a0 = fr_load(cols.p[q] + 2 * i), a1 = fr_load(cols.p[q] + 2 * i + 1);
            const Fr e0 = fr_load(cols.p[nargs + q] + 2 * i), e1 = fr_load(cols.p[nargs + q] + 2 * i + 1);
            Fr t0 = fr_mul(a1, e1);
            Fr t1 = fr_mul(fr_sub(fr_dbl(a1), a0), fr_sub(fr_dbl(e1), e0));
            if (q) {
                const Fr g = fr_load(gp + q);
                t0 = fr_mul(t0, g);
                t1 = fr_mul(t1, g);
            }
            acc[0] = fr_add(acc[0], t0);
            acc[1] = fr_add(acc[1], t1);
        }
    }
    block_reduce_finish<2>(acc, fc);
}

// inclusive->exclusive prefix sums of a (short) eq level: prefix[0] = 0, prefix[k] = sum_{i<k} v[i]; single block
__global__ void __launch_bounds__(SC_THREADS) k_prefix_sums(const Fr* __restrict__ v, uint32_t n, Fr* __restrict__ prefix) {
    __shared__ Fr part[SC_THREADS];
    __shared__ Fr carry;
    if (threadIdx.x == 0) { carry = fr_zero(); fr_store(prefix, fr_zero()); }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += SC_THREADS) {
        const uint32_t i = base + threadIdx.x;
        part[threadIdx.x] = (i < n) ? fr_load(v + i) : fr_zero();
        __syncthreads();
        for (uint32_t s = 1; s < SC_THREADS; s <<= 1) {
            Fr t = (threadIdx.x >= s) ? part[threadIdx.x - s] : fr_zero();
            __syncthreads();
            part[threadIdx.x] = fr_add(part[threadIdx.x], t);
            __syncthreads();
        }
        if (i < n) fr_store(prefix + i + 1, fr_add(carry, part[threadIdx.x]));
        __syncthreads();
        if (threadIdx.x == SC_THREADS - 1) carry = fr_add(carry, part[threadIdx.x]);
        __syncthreads();
    }
}

// Prefix sums of every level of the padded eq sequence in one launch, without a scan.  Level i has
// len_i = (i <= padded ? 1 : 2^(i - padded)) entries at offset off_i = sum_{j<i} len_j; its len_i + 1 prefix sums go to
// prefix + off_i + i.  An eq level splits every entry of the level below into two that add up to it exactly
// (next[2j] = w - r w, next[2j+1] = r w, utils.rs:222-250), so
//   P_i[2t] = P_{i-1}[t],  P_i[2t+1] = P_{i-1}[t] + E_i[2t]   =>   P_i[n] = sum_{b : bit b of n set} E_{i-b}[(n >> b) - 1]
// -- the same field elements as the running sums of vecvec.rs:101-109.  blockIdx.y = level.
__global__ void __launch_bounds__(SC_THREADS) k_prefix_sums_levels(const Fr* __restrict__ seq, Fr* __restrict__ prefix,
                                                                    uint32_t padded, uint32_t nlevels_minus1) {
    const uint32_t lv = blockIdx.y;
    if (lv > nlevels_minus1) return;
    const uint32_t j = lv > padded ? lv - padded : 0;
    const uint32_t len = 1u << j;
    const uint32_t n = blockIdx.x * SC_THREADS + threadIdx.x;
    if (n > len) return;
    // offset of level l >= padded: padded - 1 + 2^(l - padded); of level l < padded: l
    const uint64_t off_lv = lv >= padded ? (uint64_t)padded - 1 + len : lv;
    Fr s = fr_zero();
    if (lv <= padded) {
        if (n) s = fr_load(seq + off_lv);
    } else {
        for (uint32_t b = 0; b <= j; b++)
            if ((n >> b) & 1u) {
                const uint64_t off_b = (uint64_t)padded - 1 + (1u << (j - b));
                s = fr_add(s, fr_load(seq + off_b + ((n >> b) - 1)));
            }
    }
    fr_store(prefix + off_lv + lv + n, s);
}

// VecVec fold: out row = pad2(len/2) cells, cell p < len/2 = p0 + t (p1 - p0), the extra cell = row pad
// (bind_21, vecvec.rs:420-441); blockIdx.y = column
__global__ void __launch_bounds__(SC_THREADS) k_vv_fold(ColPtrs in, ColPtrsMut out, const uint32_t* __restrict__ off_in,
                                                         const uint32_t* __restrict__ off_out, uint32_t nrows, Fr t,
                                                         PadCols pad, const Fr* __restrict__ d_t, int ncols,
                                                         const uint32_t* __restrict__ coarse_out, GateArgs g) {
    if (g.bar_slot) t = gated_challenge(g, nullptr);   // pre-enqueued small fold: the gate is in here
    else if (d_t) t = fr_load(d_t);  // pre-enqueued fold: the challenge arrives through the gate kernel (k_fold_gate)
    const uint32_t j = blockIdx.x * SC_THREADS + threadIdx.x;
    const uint32_t total = off_out[nrows];
    uint32_t r;
    if (coarse_out) {   // the coarse table of the output layout: one load instead of the block's two 13-step searches (small folds are all latency)
        if (j >= total) return;
        r = find_row_coarse(off_out, nrows, coarse_out, j);
    } else {
        r = find_row_block(off_out, nrows, j, j < total, total);
        if (j >= total) return;
    }
    const uint32_t p = j - off_out[r];
    const uint32_t in0 = off_in[r], half = (off_in[r + 1] - in0) >> 1;
    // two columns per thread (grid y = ceil(k / 2)): four loads in flight per thread, one row lookup for both
    const int c0 = 2 * blockIdx.y, c1 = c0 + 1;
    const bool has1 = c1 < ncols;
    if (p < half) {
        const Fr a0 = fr_load(in.p[c0] + in0 + 2 * p), a1 = fr_load(in.p[c0] + in0 + 2 * p + 1);
        Fr b0 = a0, b1 = a1;
        if (has1) { b0 = fr_load(in.p[c1] + in0 + 2 * p); b1 = fr_load(in.p[c1] + in0 + 2 * p + 1); }
        fr_store(out.p[c0] + j, fr_add(a0, fr_mul(t, fr_sub(a1, a0))));
        if (has1) fr_store(out.p[c1] + j, fr_add(b0, fr_mul(t, fr_sub(b1, b0))));
    } else {
        fr_store(out.p[c0] + j, pad.v[c0]);
        if (has1) fr_store(out.p[c1] + j, pad.v[c1]);
    }
}

// bind_into_dense (vecvec_eq.rs:157-175): rows of 0 or 2 cells -> one dense value per row
__global__ void __launch_bounds__(SC_THREADS) k_vv_fold_to_dense(ColPtrs in, ColPtrsMut out, const uint32_t* __restrict__ off_in,
                                                                  uint32_t nrows, uint32_t nrows_dense, Fr t,
                                                                  PadCols row_pad, PadCols col_pad) {
    const uint32_t r = blockIdx.x * SC_THREADS + threadIdx.x;
    if (r >= nrows_dense) return;
    const int c = blockIdx.y;
    Fr v;
    if (r >= nrows) v = col_pad.v[c];
    else {
        const uint32_t in0 = off_in[r], len = off_in[r + 1] - in0;
        if (len == 0) v = row_pad.v[c];
        else {
            const Fr p0 = fr_load(in.p[c] + in0), p1 = fr_load(in.p[c] + in0 + 1);
            v = fr_add(p0, fr_mul(t, fr_sub(p1, p0)));
        }
    }
    fr_store(out.p[c] + r, v);
}

// Generic dense round (sumcheck.rs:283-316) for F = EqWrapper(GammaWrapper(f)) [kind 0: last column is eq]
// or Prod3 [kind 1]: acc[s] += F(p1 + s * (p1 - p0)), s = 0..D-1.
// SPLIT: blockIdx.y = D * segment + s (kind 1: = s); one thread per (pair, segment, evaluation point).
template <int D, bool SPLIT>
__global__ void __launch_bounds__(SC_THREADS) k_round_generic(int kind, SegPlan sp, ColPtrs cols, int ncols,
                                                               const Fr* __restrict__ gp, uint64_t npairs, FinishCtx fc) {
    Fr acc[D];
#pragma unroll
    for (int s = 0; s < D; s++) acc[s] = fr_zero();
    for (uint64_t i = (uint64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * SC_THREADS) {
        const int s_lo = SPLIT ? (int)(blockIdx.y % D) : 0, s_hi = SPLIT ? s_lo + 1 : D;
        if (kind == 1) {
            Fr a[3], d[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const Fr p0 = fr_load(cols.p[q] + 2 * i), p1 = fr_load(cols.p[q] + 2 * i + 1);
                a[q] = p1;
                d[q] = fr_sub(p1, p0);
            }
            for (int s = 0; s < s_hi; s++) {
                if (s) {
#pragma unroll
                    for (int q = 0; q < 3; q++) a[q] = fr_add(a[q], d[q]);
                }
                if (s >= s_lo) acc[s] = fr_add(acc[s], fr_mul(fr_mul(a[0], a[1]), a[2]));
            }
        } else {
            const Fr e0 = fr_load(cols.p[ncols - 1] + 2 * i), e1 = fr_load(cols.p[ncols - 1] + 2 * i + 1);
            const Fr ed = fr_sub(e1, e0);
            Fr G[D];
#pragma unroll
            for (int s = 0; s < D; s++) G[s] = fr_zero();
            const int g_lo = SPLIT ? (int)(blockIdx.y / D) : 0, g_hi = SPLIT ? g_lo + 1 : sp.nseg;
            for (int sg = g_lo; sg < g_hi; sg++) {
                const Seg g = sp.seg[sg];
                Fr a[6], d[6];
#pragma unroll
                for (int q = 0; q < 6; q++)
                    if (q < g.n_in) {
                        const Fr p0 = fr_load(cols.p[g.in[q]] + 2 * i), p1 = fr_load(cols.p[g.in[q]] + 2 * i + 1);
                        a[q] = p1;
                        d[q] = fr_sub(p1, p0);
                    }
                for (int s = 0; s < s_hi; s++) {
                    if (s) {
#pragma unroll
                        for (int q = 0; q < 6; q++)
                            if (q < g.n_in) a[q] = fr_add(a[q], d[q]);
                    }
                    if (s < s_lo) continue;
                    Fr o[4];
                    prim_exec(g.prim, a, o);
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (q < g.n_out) {
                            const int oc = g.out0 + q;
                            G[s] = fr_add(G[s], oc == 0 ? o[q] : fr_mul(fr_load(gp + oc), o[q]));
                        }
                }
            }
            Fr e = e1;
            for (int s = 0; s < s_hi; s++) {
                if (s) e = fr_add(e, ed);
                if (s >= s_lo) acc[s] = fr_add(acc[s], fr_mul(G[s], e));
            }
        }
    }
    block_reduce_finish<D>(acc, fc);
}

}  // namespace gm

using namespace gm;

// ============================================================================================ objects
static constexpr uint32_t SC_MAX_BLOCKS = 4096;

struct gm_sc {
    virtual ~gm_sc() {}
    virtual int32_t unipoly(std::vector<Fr>* coeffs) = 0;
    virtual int32_t bind(const Fr& t) = 0;
    virtual int32_t final_evals(std::vector<Fr>* out) = 0;
    virtual Fr claim() const = 0;
    hipStream_t stream = nullptr;
};

namespace {

// a field element as three self-validating 16-byte chunks (12 data bytes + tag), each ONE aligned store (see fr_chunks_load)
static void write_chunks16(volatile uint32_t* p, const Fr& t, uint32_t tag) {
    alignas(16) uint32_t c[12] = {t.l[0], t.l[1], t.l[2], tag, t.l[3], t.l[4], t.l[5], tag, t.l[6], t.l[7], 0u, tag};
    for (int j = 0; j < 3; j++) {
        const __m128i vv = _mm_load_si128(reinterpret_cast<const __m128i*>(c + 4 * j));
        _mm_store_si128(reinterpret_cast<__m128i*>(const_cast<uint32_t*>(p) + 4 * j), vv);
    }
}
// 4 KiB of fine-grained DEVICE memory the host can store into (large BAR), or nullptr: GM_STAGE_BAR=0, no large BAR, or the probe
// store did not arrive.  The host writes challenges there so that waiting kernels poll device memory instead of host memory.
static uint32_t* alloc_host_writable_device_words() {
    static const bool off = [] { const char* e = getenv("GM_STAGE_BAR"); return e && e[0] == '0'; }();
    int large_bar = 0, dev_ = 0;
    (void)hipGetDevice(&dev_);
    uint32_t* ret = nullptr;
    if (!off && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev_) == hipSuccess && large_bar) {
        uint32_t* p = nullptr;
        if (hipExtMallocWithFlags((void**)&p, 4096, hipDeviceMallocFinegrained) == hipSuccess && p) {
            // (the NULL stream only, not the device: another rank-thread of this process may have a kernel in flight that waits for
            // a challenge its host will publish only after an exchange with THIS thread)
            bool good = hipMemset(p, 0, 4096) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
            if (good) {
                reinterpret_cast<volatile uint32_t*>(p)[1000] = 0x5eed1234u;   // probe: a host store the device must see
                _mm_sfence();
                uint32_t back = 0;
                good = hipMemcpy(&back, p + 1000, 4, hipMemcpyDeviceToHost) == hipSuccess && back == 0x5eed1234u;
                reinterpret_cast<volatile uint32_t*>(p)[1000] = 0;
                _mm_sfence();
            }
            if (good) ret = p; else (void)hipFree(p);
        }
    }
    (void)hipGetLastError();
    return ret;
}

struct RoundScratch {
    DevBuf partial;
    // the arrival counter (+ the device copy of a pre-enqueued fold's challenge) and the limb accumulators: a few hundred bytes that
    // every kernel leaves at zero -- kept per (host thread, device) for the life of the process instead of allocated and memset per
    // object (two fill launches per layer, ~9 us of a late layer's ~45 us of set-up).  Keyed by the STREAM as well: objects that share
    // a stream (a VecVec object and its dense stage, the two lock-step objects of the pushforward argument) run their kernels one
    // after the other; a thread that drives two streams gets two sets.
    struct Persistent { void* p = nullptr; uint32_t* bar = nullptr; };
    struct CounterView { void* p = nullptr; } counter, accbuf;
    uint32_t* bar = nullptr;   // host-writable device words (4 challenge slots of 12 words) of this object's set, or nullptr
    std::shared_ptr<Persistent> pset;   // keeps the set out of the recycling below while this object lives
    const uint32_t* bar_slot(uint32_t round) const { return bar ? bar + 12 * (round & 3) : nullptr; }
    // Sets per (host thread, device), keyed by stream.  A caller that opens a fresh stream per proof must not grow this without
    // bound nor stall the device at every new stream: a new set is zeroed IN STREAM ORDER (no device synchronisation); the one 4 KiB
    // page of host-writable device memory per (thread, device) -- its probe is the only synchronising step, once -- is cut into
    // sixteen 256-byte slices, one per set (later sets run without: bar == nullptr, the pinned-memory path); and once MAX_SETS streams
    // have been seen, the sets no live object refers to are recycled for new streams behind ONE hipDeviceSynchronize (their last
    // kernels may still be running on streams this thread can no longer name).
    static constexpr size_t SET_BYTES = 256 + 24 * 128;
    static constexpr size_t MAX_SETS = 64;
    struct PerDev {
        uint32_t* bar_page = nullptr;
        bool bar_tried = false;
        uint32_t bar_slices_used = 0;
        std::map<hipStream_t, std::shared_ptr<Persistent>> sets;
        std::vector<std::shared_ptr<Persistent>> spare;
    };
    static int32_t persistent(hipStream_t stream, std::shared_ptr<Persistent>* keep) {
        static thread_local std::map<int, PerDev> per_dev;
        int dev = 0;
        (void)hipGetDevice(&dev);
        GM_REQUIRE(dev >= 0 && dev < GM_MAX_DEVICES, "device id %d: the per-device tables of this library hold %d devices", dev, GM_MAX_DEVICES);
        PerDev& pd = per_dev[dev];
        auto it = pd.sets.find(stream);
        if (it != pd.sets.end()) { *keep = it->second; return GM_OK; }
        if (pd.sets.size() >= MAX_SETS && pd.spare.empty()) {
            std::vector<hipStream_t> idle;
            for (auto& kv : pd.sets) if (kv.second.use_count() == 1) idle.push_back(kv.first);
            if (!idle.empty()) {
                GM_HIP(hipDeviceSynchronize());
                for (hipStream_t st : idle) {
                    GM_HIP(hipMemset(pd.sets[st]->p, 0, SET_BYTES));   // (kernels leave the words at zero; a killed launch may not have)
                    pd.spare.push_back(pd.sets[st]);
                    pd.sets.erase(st);
                }
            }
        }
        std::shared_ptr<Persistent> e;
        if (!pd.spare.empty()) {
            e = pd.spare.back();
            pd.spare.pop_back();
        } else {
            e = std::make_shared<Persistent>();
            GM_HIP(hipMalloc(&e->p, SET_BYTES));
            GM_HIP(hipMemsetAsync(e->p, 0, SET_BYTES, stream));   // every use of the set is on this stream: ordered behind the zeros
            if (!pd.bar_tried) {
                pd.bar_tried = true;
                pd.bar_page = alloc_host_writable_device_words();
            }
            if (pd.bar_page && pd.bar_slices_used < 16) e->bar = pd.bar_page + 64 * pd.bar_slices_used++;
        }
        pd.sets[stream] = e;
        *keep = e;
        return GM_OK;
    }
    Fr* h_result = nullptr;  // pinned, device-visible
    bool own_pinned = false;
    int32_t init(hipStream_t s) {
        int32_t rc = partial.alloc((size_t)SC_MAX_BLOCKS * 3 * sizeof(Fr));
        if (rc) return rc;
        // counter: [0] last-block counter, [64..96) the device copy of a pre-enqueued fold's challenge; accbuf: the limb accumulators
        // of block_reduce_finish (3 sums x 8 limbs, one 128-byte line each)
        rc = persistent(s, &pset);
        if (rc) return rc;
        counter.p = pset->p;
        accbuf.p = static_cast<char*>(pset->p) + 256;
        bar = pset->bar;
        if (shared_pinned()) {
            h_result = shared_pinned();
            own_pinned = false;
        } else {
            GM_HIP(hipHostMalloc((void**)&h_result, 16 * sizeof(Fr), hipHostMallocCoherent | hipHostMallocMapped));
            memset(h_result, 0, 16 * sizeof(Fr));
            own_pinned = true;
        }
        return GM_OK;
    }
    ~RoundScratch() {
        if (h_result && own_pinned) (void)hipHostFree(h_result);
    }
    static uint32_t& seq_counter() {
        static thread_local uint32_t c = 0;
        return c;
    }
    uint32_t expect = 0;
    FinishCtx ctx() {
        expect = ++seq_counter();
        if (expect == 0) expect = ++seq_counter();
        return FinishCtx{partial.fr(), reinterpret_cast<uint32_t*>(counter.p), h_result, expect,
                         reinterpret_cast<unsigned long long*>(accbuf.p), 1u};
    }
    // ---- sharded rounds over a device-side collective (see k_sum_ranks)
    DevBuf xslot, xall;
    static bool dev_exchange(const Shard& sh) {
        static const bool off = [] { const char* e = getenv("GM_SHARD_HOST_EXCHANGE"); return e && e[0] == '1'; }();   // A/B switch
        return sh.comm && sh.comm->all_gather_dev && !off;
    }
    // the round kernel reports into this rank's device slot instead of pinned memory
    int32_t ctx_dev(const Shard& sh, FinishCtx* fc) {
        if (!xslot.p) {
            int32_t rc = xslot.alloc(8 * sizeof(Fr));
            if (rc) return rc;
            rc = xall.alloc((size_t)sh.world * 4 * sizeof(Fr));
            if (rc) return rc;
        }
        *fc = ctx();
        fc->out = xslot.fr();
        fc->raw = 0;   // k_sum_ranks adds field elements
        return GM_OK;
    }
    // all-gather of the slots + the one-wave sum, both on the prover's stream; finish() then sees the SUMS in pinned memory
    int32_t exchange(const Shard& sh, int nacc, hipStream_t s) {
        const int32_t rc = sh.comm->all_gather_dev(sh.comm->ctx, xslot.p, xall.p, 4 * sizeof(Fr), s);
        if (rc) return set_err(GM_ERR_STATE, "gm_comm all_gather_dev failed with %d", rc);
        hipLaunchKernelGGL(k_sum_ranks, dim3(1), dim3(64), 0, s, xall.fr(), sh.world, nacc, h_result, expect);
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    // The launch writes `nacc` results and then the sequence number into the pinned buffer.  Poll the sequence
    // slot (a PCIe write lands in ~2 us; hipStreamSynchronize costs 10-20 us per round); fall back to the stream
    // synchronisation if it has not shown up after a bounded spin.
    int32_t finish(int nacc, hipStream_t s, Fr* out) { return finish_seq(expect, nacc, s, out); }
    // pinned layout (16 elements): [0..6] round results, [7] result sequence number, [8..11] challenge slots of pre-enqueued
    // folds, [12] ticket word the waiting folds watch (l[0]) and their status word (l[1])
    Fr* t_slot(uint32_t round) const { return h_result + 8 + (round & 3); }
    uint32_t* ticket_word() const { return reinterpret_cast<uint32_t*>(h_result + 12); }
    static uint32_t& ticket_counter() {
        static thread_local uint32_t c = 0;
        return c;
    }
    void publish(uint32_t round, const Fr& t, uint32_t ticket) {
        if (bar) {   // a fold with its gate inside may be polling device memory (gated_challenge)
            write_chunks16(bar + 12 * (round & 3), t, ticket);
            _mm_sfence();
        }
        *t_slot(round) = t;
        std::atomic_thread_fence(std::memory_order_release);
        *reinterpret_cast<volatile uint32_t*>(ticket_word()) = ticket;
    }
    // small pre-enqueued folds carry their gate inside (no k_fold_gate launch) when the host can write into device memory
    // (every workgroup of such a fold spins until the challenge arrives.  Ranks of a sharded proof that SHARE a device -- rehearsals,
    // tests -- wait for each other's round sums: four ranks' spinning folds of 512 workgroups each fill the device and keep the round
    // kernel of the rank everybody waits for from being scheduled.  Sharded objects therefore keep the inside gate to folds of <= 64
    // workgroups; a rank with a device of its own loses a gate launch on the larger ones.)
    GateArgs gate_in_fold(uint32_t round, uint32_t ticket, uint64_t fold_blocks, bool sharded = false) const {
        static const bool off = [] { const char* e = getenv("GM_FOLD_GATE_INSIDE"); return e && e[0] == '0'; }();
        GateArgs g;
        g.bar_slot = (!off && bar && fold_blocks <= (sharded ? 64u : 512u)) ? bar + 12 * (round & 3) : nullptr;
        g.ticket = ticket;
        g.status = const_cast<uint32_t*>(reinterpret_cast<volatile uint32_t*>(ticket_word())) + 1;
        g.timeout_ticks = wait_timeout_ticks();
        return g;
    }
    // can_sync = false: a pre-enqueued fold is waiting in the stream for a challenge the host has not published yet, so a
    // stream synchronisation would wait for it; keep polling (bounded by wall time) instead
    // Two result formats (block_reduce_finish): grids of <= 512 blocks send 8 nacc tagged 64-bit limb sums (tag = want % 4095 + 1 in
    // the top 12 bits; no sequence word) which are reduced here and zeroed once read (a stale value can then never carry a valid tag);
    // larger grids send nacc field elements and then the sequence word.
    bool tagged_sums(uint32_t want, int nacc, Fr* out) {
        volatile unsigned long long* raw = reinterpret_cast<volatile unsigned long long*>(h_result);
        const unsigned long long tag = (unsigned long long)(want % 4095u + 1u);
        if (nacc > 3) return false;
        unsigned long long v[24];
        for (int i = 0; i < 8 * nacc; i++) {
            v[i] = raw[i];
            if ((v[i] >> 52) != tag) return false;
        }
        for (int a = 0; a < nacc; a++) {
            uint32_t lo[8], hi[8];
            for (int l = 0; l < 8; l++) { const unsigned long long t = v[8 * a + l] & ((1ull << 52) - 1); lo[l] = (uint32_t)t; hi[l] = (uint32_t)(t >> 32); }
            out[a] = limb_sums_mod_p(lo, hi);
        }
        for (int i = 0; i < 8 * nacc; i++) raw[i] = 0;
        return true;
    }
    int32_t finish_seq(uint32_t want, int nacc, hipStream_t s, Fr* out, bool can_sync = true) {
        volatile uint32_t* slot = reinterpret_cast<volatile uint32_t*>(h_result + 7);
        bool seen = false;
        for (int spin = 0; spin < 200000; spin++) {
            if (tagged_sums(want, nacc, out)) return GM_OK;
            if (*slot == want) { seen = true; break; }
            __builtin_ia32_pause();
        }
        if (!seen && !can_sync) {
            const auto t0 = std::chrono::steady_clock::now();
            while (!seen && std::chrono::steady_clock::now() - t0 < wait_timeout_host()) {
                for (int spin = 0; spin < 10000 && !seen; spin++) {
                    if (tagged_sums(want, nacc, out)) return GM_OK;
                    if (*slot == want) seen = true;
                    __builtin_ia32_pause();
                }
            }
            if (!seen) return set_err(GM_ERR_STATE, "round kernel result did not arrive within %u ms (gm_set_wait_timeout_ms)", wait_timeout_ms().load());
        }
        if (!seen) {
            GM_HIP(hipStreamSynchronize(s));
            if (tagged_sums(want, nacc, out)) return GM_OK;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int a = 0; a < nacc; a++) out[a] = h_result[a];
        return GM_OK;
    }
};

// pinned staging of k_gather_finals: one per host thread and device, kept for the life of the process
struct FinalsStage {
    Fr* h = nullptr;
    uint32_t seq = 0;
};
static int32_t gather_finals(const Fr* const* cur, int k, hipStream_t s, std::vector<Fr>* out) {
    static thread_local FinalsStage per_dev[GM_MAX_DEVICES];   // per (host thread, device), as TailStage
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    GM_REQUIRE(dev_ >= 0 && dev_ < GM_MAX_DEVICES, "device id %d: the per-device tables of this library hold %d devices", dev_, GM_MAX_DEVICES);
    FinalsStage& st = per_dev[dev_];
    if (!st.h) {
        GM_HIP(hipHostMalloc((void**)&st.h, (GM_MAX_COLS + 1) * sizeof(Fr), hipHostMallocCoherent | hipHostMallocMapped));
        memset(st.h, 0, (GM_MAX_COLS + 1) * sizeof(Fr));
    }
    GM_REQUIRE(k >= 0 && k <= GM_MAX_COLS, "too many columns");
    out->resize(k);
    if (k == 0) return GM_OK;
    ColPtrs cp;
    for (int i = 0; i < k; i++) cp.p[i] = cur[i];
    if (++st.seq == 0) ++st.seq;
    uint32_t* h_seq = reinterpret_cast<uint32_t*>(st.h + GM_MAX_COLS);
    hipLaunchKernelGGL(k_gather_finals, dim3(1), dim3(64), 0, s, cp, k, st.h, h_seq, st.seq);
    GM_LAUNCH_CHECK();
    volatile uint32_t* slot = h_seq;
    bool seen = false;
    for (int spin = 0; spin < 200000; spin++) {
        if (*slot == st.seq) { seen = true; break; }
        __builtin_ia32_pause();
    }
    if (!seen) GM_HIP(hipStreamSynchronize(s));
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < k; i++) (*out)[i] = st.h[i];
    return GM_OK;
}

// pinned staging + device-side counters of k_stage: one per host thread and device, kept for the life of the process
struct TailStage {
    char* base = nullptr;
    uint32_t* rep() const { return reinterpret_cast<uint32_t*>(base); }            // 2 report slots of 96 words (3 sums x 8 limb chunks)
    uint32_t* tkt() const { return rep() + 192; }                                  // 2 challenge slots of 12 words (3 chunks)
    uint32_t* status() const { return rep() + 224; }
    Fr* finals() const { return reinterpret_cast<Fr*>(base + 1024); }              // 32 slots per column: what the launch leaves of it
    uint32_t* fin_seq() const { return reinterpret_cast<uint32_t*>(finals() + 32 * GM_MAX_COLS); }   // [segment][slice]
    uint32_t counter = 0;
    // device state of the launches of this host thread on this device (never memset: the kernel leaves its counters at zero, tags
    // are unique)
    uint32_t* d_state = nullptr;
    uint32_t* tkt_bar = nullptr;            // 256 bytes of fine-grained device memory the host writes challenges into (large BAR), or nullptr
    bool tkt_bar_tried = false;
    unsigned long long* d_acc = nullptr;   // the rounds' limb accumulators (k_stage's exchange): 32 rounds x 24 lines of 128 bytes
    static constexpr size_t ACC_BYTES = (size_t)32 * 24 * 128;
    bool d_state_dirty = true;
    uint32_t arrive_total = 0;   // blocks counted in by the residency barrier since the state was last zeroed
    static constexpr size_t BYTES = 1024 + 32 * GM_MAX_COLS * sizeof(Fr) + (size_t)GM_MAX_SEGS * STAGE_MAX_SLICES * 4;
    static constexpr size_t STATE_BYTES = 256 + 2 * GM_MAX_SEGS * 4 + 64;   // relay | round counters | merge counters | residency word
    static constexpr size_t ARRIVE_WORD = (256 + 2 * GM_MAX_SEGS * 4) / 4;
};
// one per (host thread, device): a thread that drives several GPUs (gm_set_device between calls) gets device-side counters and
// pinned staging of its own on each of them
static int32_t tail_stage(TailStage** out) {
    static thread_local TailStage per_dev[GM_MAX_DEVICES];
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    GM_REQUIRE(dev_ >= 0 && dev_ < GM_MAX_DEVICES, "device id %d: the per-device tables of this library hold %d devices", dev_, GM_MAX_DEVICES);
    TailStage& st = per_dev[dev_];
    if (!st.base) {
        GM_HIP(hipHostMalloc((void**)&st.base, TailStage::BYTES, hipHostMallocCoherent | hipHostMallocMapped));
        memset(st.base, 0, TailStage::BYTES);
    }
    *out = &st;
    return GM_OK;
}

static bool stage_enabled() {
    // the stage kernel is itself a pre-enqueued mechanism: GM_SC_NO_PIPELINE=1 (plain rounds: kernel, sync, fold) switches it off too
    static const bool v = [] {
        const char* e = getenv("GM_SC_NO_TAIL");
        const char* p = getenv("GM_SC_NO_PIPELINE");
        return !(e && e[0] == '1') && !(p && p[0] == '1');
    }();
    return v;
}
static bool spin_for(volatile uint32_t* slot, uint32_t want) {
    for (int spin = 0; spin < 400000; spin++) {
        if (*slot == want) return true;
        __builtin_ia32_pause();
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::steady_clock::now() - t0 < wait_timeout_host() + std::chrono::milliseconds(200))
        for (int spin = 0; spin < 10000; spin++) {
            if (*slot == want) return true;
            __builtin_ia32_pause();
        }
    return false;
}

// The last rounds of a dense stage are a handful of pairs: a round trip through the device costs ~20 us each, the host does
// them in a few microseconds.  How many trailing rounds the host takes: as many as fit ~600 field multiplications in total
// (per pair: two evaluations of the layer function + the gamma combination), at most 5, and at least one round stays on the device.
static int stage_host_rounds(const SegPlan& sp, int n_dense) {
    static const int cap = [] { const char* e = getenv("GM_SC_HOST_ROUNDS"); return e ? atoi(e) : 5; }();
    int mul_per_pair = 0;
    for (int sg = 0; sg < sp.nseg; sg++) mul_per_pair += 2 * (6 + sp.seg[sg].n_out);
    int h = 0;
    while (h < cap && h + 1 < n_dense && mul_per_pair * ((2 << h) - 1) <= 600) h++;
    return h;
}

// Co-residency budget of k_stage.  Its blocks wait for each other, so all of them must be resident at once: the budget is what
// the device can hold (occupancy of k_stage x compute units, queried once per device), a launch larger than that is not made
// (StageRun::fits), and host threads that prove concurrently on one device share it -- a launch waits until the blocks of the
// launches in flight plus its own fit (two 512-block launches from two threads would otherwise each hold part of the chip and
// spin until their time-outs).  A thread may WAIT for its share only while none of its own kernels is waiting for it (a
// pre-enqueued fold's gate spins on a CU until this thread publishes the challenge: waiting then could keep the holder's blocks
// from ever becoming resident -- a cycle); launches made while such a gate is in the stream only TRY, and the layer goes on with
// ordinary round kernels when the device is busy.
struct StageSlots {
    std::mutex mu;
    std::condition_variable cv;
    // budget in units of 1 / (pw pn) of a compute unit, pw / pn = workgroups of the wide / narrow instance a CU holds: a wide
    // workgroup costs pn units, a narrow one pw, the device has CUs * pw * pn
    uint32_t capacity[GM_MAX_DEVICES] = {0}, in_flight[GM_MAX_DEVICES] = {0}, cost_wide[GM_MAX_DEVICES] = {0}, cost_narrow[GM_MAX_DEVICES] = {0};
    static StageSlots& get() { static StageSlots s; return s; }
    // (ids >= GM_MAX_DEVICES never get here: no sumcheck object can be created on such a device, RoundScratch::persistent refuses)
    static int device() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < GM_MAX_DEVICES) ? d : 0; }
    uint32_t cap(int dev) {
        std::lock_guard<std::mutex> g(mu);
        if (!capacity[dev]) {
            int pw = 0, pn = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pw, k_stage<6>, 256, 0) != hipSuccess || pw < 1) pw = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pn, k_stage<3>, 256, 0) != hipSuccess || pn < 1) pn = 1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
            (void)hipGetLastError();
            cost_wide[dev] = (uint32_t)pn;
            cost_narrow[dev] = (uint32_t)pw;
            capacity[dev] = (uint32_t)cus * (uint32_t)pw * (uint32_t)pn;
        }
        return capacity[dev];
    }
    uint32_t cost(int dev, uint32_t blocks, bool narrow) { (void)cap(dev); return blocks * (narrow ? cost_narrow[dev] : cost_wide[dev]); }
    bool acquire(int dev, uint32_t n) {
        const uint32_t c = cap(dev);
        std::unique_lock<std::mutex> g(mu);
        return cv.wait_for(g, wait_timeout_host(), [&] { return in_flight[dev] + n <= c; }) ? (in_flight[dev] += n, true) : false;
    }
    bool try_acquire(int dev, uint32_t n) {
        const uint32_t c = cap(dev);
        std::lock_guard<std::mutex> g(mu);
        if (in_flight[dev] + n > c) return false;
        in_flight[dev] += n;
        return true;
    }
    void release(int dev, uint32_t n) {
        { std::lock_guard<std::mutex> g(mu); in_flight[dev] -= n; }
        cv.notify_all();
    }
};
// every segment of the plan is a part or a narrow primitive: the narrow instance of the stage kernel serves it
static bool stage_plan_is_narrow(const SegPlan& sp) {
    for (int s = 0; s < sp.nseg; s++)
        if (sp.seg[s].n_in > 3 || !prim_fits3(sp.seg[s].prim)) return false;
    return true;
}
// StageRun::launch could not get its share of the device without waiting (internal; never leaves the library)
#define GM_STAGE_BUSY 1000
// StageRun::sums, first round of a launch: the grid did not become resident as a whole, the launch has left without touching anything
#define GM_STAGE_NOT_RESIDENT 1001

// process-wide tally (gm_sc_stage_counts: tests and benches check which path a proof took)
static std::atomic<uint64_t> g_stage_launched{0}, g_stage_left{0};

// One launch of k_stage seen from the host.  A VecVec object that enters its thin rounds creates it; the dense object it hands
// over to (bind_into_dense) keeps using the same launch.
struct StageRun {
    int slot_dev = 0;
    uint32_t slots_held = 0;
    bool launched = false;   // the kernel is (or was) in the stream: only then are there waits to release
    TailStage* st = nullptr;
    hipStream_t stream = nullptr;
    uint32_t ticket0 = 0;
    int n_thin = 0, n_dense = 0, nseg = 0;   // n_dense: dense rounds that run on the device
    uint32_t n_elems0 = 0;                // dense elements at the start of the dense phase
    int published = 0;                    // rounds whose challenge the host has published
    uint32_t gx = 0, nsl = 1;
    int merge_after = 0;                  // dense rounds 0..merge_after report from every slice, later ones from slice 0 only
    DevBuf xbuf, dbg;
    static bool debug() {
        static const bool v = [] { const char* e = getenv("GM_STAGE_DEBUG"); return e && e[0] == '1'; }();
        return v;
    }
    void dump_debug() {
        if (!dbg.p) return;
        std::vector<uint64_t> h(2 * 32 * 8);
        if (hipMemcpy(h.data(), dbg.p, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
        fprintf(stderr, "[k_stage %ux%u thin %d dense %d] host turn-around %.2f us per round (sums seen -> challenge written); per round, us since the round's start: block 0 | block 1\n",
                gx, nsl, n_thin, n_dense, host_n ? host_us / host_n : 0.0);
        for (int r = 0; r < total(); r++) {
            fprintf(stderr, "  r%02d", r);
            for (int b = 0; b < 2; b++) {
                const uint64_t* e = &h[((size_t)b * 32 + r) * 8];
                const uint64_t nxt = (r + 1 < total()) ? h[((size_t)b * 32 + r + 1) * 8] : e[5];
                fprintf(stderr, "  eval %5.1f red %5.1f pub %5.1f last %5.1f wait %5.1f | next-start %5.1f", (e[1] - e[0]) / 100.0,
                        (e[2] - e[1]) / 100.0, (e[3] - e[2]) / 100.0, (e[4] - e[3]) / 100.0, (e[5] - e[4]) / 100.0, (nxt - e[0]) / 100.0);
            }
            fprintf(stderr, "\n");
        }
    }
    int total() const { return n_thin + n_dense; }
    // geometry for `n_elems` dense elements; false = does not fit one launch
    static bool fits(int nseg_, uint64_t n_elems, int n_thin_, int n_dense_) {
        const uint64_t nsl_ = n_elems <= 256 ? 1 : n_elems / 256;
        return nseg_ >= 1 && nseg_ <= GM_MAX_SEGS && n_elems >= 2 && nsl_ <= STAGE_MAX_SLICES && 2ull * nseg_ * nsl_ <= STAGE_MAX_BLOCKS &&
               StageSlots::get().cost(StageSlots::device(), (uint32_t)(2ull * nseg_ * nsl_), false) <= StageSlots::get().cap(StageSlots::device()) &&
               n_thin_ <= 12 && n_dense_ >= 1 && n_dense_ <= STAGE_MAX_ROUNDS;
    }
    // a: geometry, data pointers, eq pointers and pads filled by the caller
    int32_t launch(const SegPlan& sp_in, const ColPtrs& cp, const Fr* d_gamma, StageArgs a, hipStream_t s, bool may_wait = true) {
        // term split (segfn.hip.h): more, shorter evaluation chains per round when the wider grid still fits the device
        static const bool no_split = [] { const char* e = getenv("GM_STAGE_SPLIT"); return e && e[0] == '0'; }();
        SegPlan sp_split;
        const bool split = !no_split && seg_plan_split_terms(sp_in, &sp_split) &&
                           fits(sp_split.nseg, a.n_elems, a.n_thin > 0 ? a.n_thin : 0, a.n_dense);
        const SegPlan& sp = split ? sp_split : sp_in;
        int32_t rc = tail_stage(&st);
        if (rc) return rc;
        stream = s;
        n_thin = a.n_thin; n_dense = a.n_dense; nseg = sp.nseg;
        n_elems0 = a.n_elems;
        gx = 2u * (uint32_t)sp.nseg;
        nsl = a.n_elems <= 256 ? 1 : a.n_elems / 256;
        const uint32_t np0 = (a.n_elems < 256 ? a.n_elems : 256u) >> 1;
        merge_after = 0;
        while ((1u << merge_after) < np0) merge_after++;
        // tickets are exact-match sequence numbers, unique per host thread: ticket0 + r for round r; ticket0 + 0x4000 releases
        // every wait of the launch (abort).  Launches are spaced 0x8000 apart.
        st->counter += 0x8000u;
        if (st->counter == 0 || st->counter > 0xffff0000u) st->counter = 0x8000u;
        ticket0 = st->counter;
        if (nsl > 1) {
            rc = xbuf.alloc((size_t)gx * 6 * STAGE_MAX_SLICES * sizeof(Fr));
            if (rc) return rc;
        }
        // small device state: [0, 128) relay, [128, 256) one arrival counter per round, [256, ..) one merge counter per
        // blockIdx.x.  It belongs to the host thread and is zeroed once: a launch leaves its counters at zero, relay tags are unique.
        // A launch that was aborted (time-out, failing transcript) may leave counters behind: the next launch zeroes again.
        if (!st->d_state) {
            GM_HIP(hipMalloc((void**)&st->d_state, TailStage::STATE_BYTES));
            GM_HIP(hipMalloc((void**)&st->d_acc, TailStage::ACC_BYTES));
            st->d_state_dirty = true;
        }
        if (!st->tkt_bar_tried) {
            // Challenges straight into device memory when the device exposes it to the host (large BAR) and a probe write arrives;
            // GM_STAGE_BAR=0 keeps the PCIe-poll + relay scheme (A/B, and the fall-back wherever the probe fails)
            st->tkt_bar_tried = true;
            st->tkt_bar = alloc_host_writable_device_words();
        }
        if (st->arrive_total > 0x40000000u) st->d_state_dirty = true;   // the cumulative arrival count stays far below bit 31
        if (st->d_state_dirty) {
            GM_HIP(hipMemsetAsync(st->d_state, 0, TailStage::STATE_BYTES, s));
            GM_HIP(hipMemsetAsync(st->d_acc, 0, TailStage::ACC_BYTES, s));
            st->d_state_dirty = false;
            st->arrive_total = 0;
        }
        void* state_p = st->d_state;
        a.h_rep = st->rep(); a.h_finals = st->finals(); a.h_fin_seq = st->fin_seq();
        a.h_tkt = st->tkt(); a.h_status = st->status();
        a.d_tkt_bar = st->tkt_bar;
        a.d_relay = reinterpret_cast<uint32_t*>(state_p);
        a.d_round_cnt = reinterpret_cast<uint32_t*>(state_p) + 32;
        a.d_merge = reinterpret_cast<uint32_t*>(state_p) + 64;
        a.d_arrive = reinterpret_cast<uint32_t*>(state_p) + TailStage::ARRIVE_WORD;
        a.d_acc = st->d_acc;
        a.d_xbuf = xbuf.fr();
        if (debug()) {
            rc = dbg.alloc(2 * 32 * 8 * 8);
            if (rc) return rc;
            GM_HIP(hipMemsetAsync(dbg.p, 0, 2 * 32 * 8 * 8, s));
            a.d_dbg = reinterpret_cast<uint64_t*>(dbg.p);
        }
        a.ticket0 = ticket0;
        a.timeout_ticks = wait_timeout_ticks();
        {
            static const bool pf_off = [] { const char* e = getenv("GM_STAGE_PAR_FOLD"); return e && e[0] == '0'; }();   // A/B
            a.par_fold = pf_off ? 0u : 1u;
        }
        // the staging outlives this launch (one per host thread): a timeout flagged by an earlier launch must not fail this one
        *reinterpret_cast<volatile uint32_t*>(st->status()) = 0;
        slot_dev = StageSlots::device();
        const bool narrow = stage_plan_is_narrow(sp);
        const uint32_t want = StageSlots::get().cost(slot_dev, gx * nsl, narrow);
        if (!may_wait) {
            // GM_STAGE_FORCE_BUSY=1 (tests): every launch that may only try finds the device busy
            static const bool force_busy = [] { const char* e = getenv("GM_STAGE_FORCE_BUSY"); return e && e[0] == '1'; }();
            if (force_busy || !StageSlots::get().try_acquire(slot_dev, want)) return GM_STAGE_BUSY;
        } else if (!StageSlots::get().acquire(slot_dev, want))
            return set_err(GM_ERR_STATE, "stage launch: the device stayed full of other threads' stage kernels (%u units wanted, %u in flight, "
                           "capacity %u; gm_set_wait_timeout_ms)", want, StageSlots::get().in_flight[slot_dev], StageSlots::get().cap(slot_dev));
        slots_held = want;
        {
            // GM_STAGE_FORCE_NONRESIDENT=1 (tests): the barrier waits for one block more than the grid has and gives up at once
            static const bool force_nr = [] { const char* e = getenv("GM_STAGE_FORCE_NONRESIDENT"); return e && e[0] == '1'; }();
            static const uint64_t res_ticks = [] { const char* e = getenv("GM_STAGE_RESIDENT_MS"); return (uint64_t)(e && atoi(e) > 0 ? atoi(e) : 10) * 100000ull; }();
            st->arrive_total += gx * nsl;
            a.arrive_target = st->arrive_total + (force_nr ? 1u : 0u);
            a.resident_ticks = force_nr ? 2000ull : res_ticks;
        }
        if (narrow) hipLaunchKernelGGL(k_stage<3>, dim3(gx, nsl), dim3(256), 0, s, sp, cp, d_gamma, a);
        else hipLaunchKernelGGL(k_stage<6>, dim3(gx, nsl), dim3(256), 0, s, sp, cp, d_gamma, a);
        GM_LAUNCH_CHECK();
        launched = true;
        published = 0;
        g_stage_launched++;
        return GM_OK;
    }
    // round r of the launch: wait for the reporting blocks and add their partials up: s[h] = sum over the (segment, h) blocks,
    // w = the tail weight (thin rounds)
    // one field element from three self-validating chunks; false while any chunk still carries another sequence number
    static bool read_chunks(const volatile uint32_t* p, uint32_t want, Fr* out) {
        if (p[3] != want || p[7] != want || p[11] != want) return false;
        std::atomic_thread_fence(std::memory_order_acquire);
        out->l[0] = p[0]; out->l[1] = p[1]; out->l[2] = p[2]; out->l[3] = p[4]; out->l[4] = p[5]; out->l[5] = p[6]; out->l[6] = p[8]; out->l[7] = p[9];
        std::atomic_thread_fence(std::memory_order_acquire);
        return p[3] == want && p[7] == want && p[11] == want;
    }
    int32_t sums(int r, Fr* s1, Fr* s2, Fr* w) {
        const uint32_t want = ticket0 + (uint32_t)r;
        const volatile uint32_t* rep = st->rep() + 96 * (r & 1);
        Fr v[3];
        // 24 chunks of {low word, high word, -, tag}: the limb sums of the three values; valid once every chunk carries the round's tag
        const int nchunks = (r == 0 && n_thin > 0) ? 24 : 16;   // the launch's first thin round also reports C_have (see k_stage)
        auto all = [&] {
            uint32_t lo[24] = {0}, hi[24] = {0};
            for (int c = 0; c < nchunks; c++)
                if (rep[4 * c + 3] != want) return false;
            std::atomic_thread_fence(std::memory_order_acquire);
            for (int c = 0; c < nchunks; c++) { lo[c] = rep[4 * c]; hi[c] = rep[4 * c + 1]; }
            std::atomic_thread_fence(std::memory_order_acquire);
            for (int c = 0; c < nchunks; c++)
                if (rep[4 * c + 3] != want) return false;
            for (int k = 0; k < 3; k++) v[k] = limb_sums_mod_p(lo + 8 * k, hi + 8 * k);
            return true;
        };
        bool seen = false;
        volatile uint32_t* stat = reinterpret_cast<volatile uint32_t*>(st->status());
        for (int spin = 0; spin < 400000 && !seen; spin++) {
            seen = all() || *stat != 0;
            if (!seen) __builtin_ia32_pause();
        }
        if (!seen) {
            const auto t0 = std::chrono::steady_clock::now();
            while (!seen && std::chrono::steady_clock::now() - t0 < wait_timeout_host() + std::chrono::milliseconds(200))
                for (int spin = 0; spin < 10000 && !seen; spin++) seen = all() || *stat != 0;
        }
        if (*stat == 2u && r == 0) {
            // the grid never became resident as a whole (other work on the device): the launch has left, nothing was written
            *stat = 0;
            st->d_state_dirty = true;
            published = total();   // no waiting block is left behind
            (void)hipStreamSynchronize(stream);
            if (slots_held) { StageSlots::get().release(slot_dev, slots_held); slots_held = 0; }
            g_stage_left++;
            return GM_STAGE_NOT_RESIDENT;
        }
        if (*stat) {
            *stat = 0;
            st->d_state_dirty = true;
            return set_err(GM_ERR_STATE, "the stage kernel timed out waiting for a challenge (gm_set_wait_timeout_ms)");
        }
        if (!seen) {
            st->d_state_dirty = true;
            return set_err(GM_ERR_STATE, "stage round result did not arrive in time (gm_set_wait_timeout_ms)");
        }
        if (r < n_thin) {
            // thin round: the device summed eval x coef; the round's eq scalar and the tail weight are applied here (see k_stage)
            if (r == 0) c_have = v[2];
            if ((size_t)r >= thin_e0.size()) return set_err(GM_ERR_STATE, "stage round %d has no eq scalar", r);
            const Fr e0 = thin_e0[r];
            v[0] = fr_mul(v[0], e0);
            v[1] = fr_mul(v[1], e0);
            v[2] = fr_sub(c_all, fr_mul(e0, c_have));
        }
        *s1 = v[0];
        *s2 = v[1];
        if (w) *w = v[2];
        if (debug()) t_sums = std::chrono::steady_clock::now();
        return GM_OK;
    }
    // thin rounds (see k_stage): entry 0 of every thin round's eq level, the sum of row_coef over this launch's rows, and -- from the
    // first round's report -- over the rows that hold a pair
    std::vector<Fr> thin_e0;
    Fr c_all = fr_zero(), c_have = fr_zero();
    // development aid (GM_STAGE_DEBUG=1): the host's turn-around, from a round's sums seen to its challenge written
    std::chrono::steady_clock::time_point t_sums;
    double host_us = 0;
    int host_n = 0;
    static void write_chunks(volatile uint32_t* p, const Fr& t, uint32_t tag) { write_chunks16(p, t, tag); }
    void publish(int r, const Fr& t) {
        if (debug()) { host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_sums).count(); host_n++; }
        if (st->tkt_bar) {
            write_chunks(st->tkt_bar + 12 * (r & 1), t, ticket0 + (uint32_t)r);
            _mm_sfence();   // device memory is mapped write-combining on the host: push the three stores out now
        } else {
            write_chunks(st->tkt() + 12 * (r & 1), t, ticket0 + (uint32_t)r);
        }
        published = r + 1;
    }
    // elements per column the launch leaves behind (1 = the final evaluations) and the slices that hold them
    uint32_t left_per_col() const { return n_elems0 >> n_dense; }
    // what the launch left of every column: cols[c][0 .. left_per_col())
    int32_t collect(int ncols, std::vector<std::vector<Fr>>* cols) {
        const uint32_t want = ticket0 + (uint32_t)total();
        const bool merged = nsl > 1 && n_dense - 1 > merge_after;
        const uint32_t slices = (nsl > 1 && !merged) ? nsl : 1;
        for (int sg = 0; sg < nseg; sg++)
            for (uint32_t sl = 0; sl < slices; sl++)
                if (!spin_for(st->fin_seq() + (size_t)sg * STAGE_MAX_SLICES + sl, want)) {
                    st->d_state_dirty = true;
                    return set_err(GM_ERR_STATE, "stage results did not arrive in time (gm_set_wait_timeout_ms)");
                }
        std::atomic_thread_fence(std::memory_order_acquire);
        const uint32_t n = left_per_col();
        cols->assign(ncols, std::vector<Fr>());
        for (int c = 0; c < ncols; c++) (*cols)[c].assign(st->finals() + (size_t)c * 32, st->finals() + (size_t)c * 32 + n);
        if (debug()) { (void)hipStreamSynchronize(stream); dump_debug(); }
        // the finals are written after a block's last wait: the launch no longer needs its share of the device
        if (slots_held) { StageSlots::get().release(slot_dev, slots_held); slots_held = 0; }
        return GM_OK;
    }
    ~StageRun() {
        if (launched && st && published < total()) {   // never leave waiting blocks behind: the release tag lets every wait of this launch through
            st->d_state_dirty = true;
            uint32_t* tk = st->tkt_bar ? st->tkt_bar : st->tkt();
            write_chunks(tk, fr_zero(), ticket0 + 0x4000u);
            write_chunks(tk + 12, fr_zero(), ticket0 + 0x4000u);
            _mm_sfence();
            (void)hipStreamSynchronize(stream);
        }
        // every block is past its last wait (the finals were collected, or the release tag let them through)
        if (slots_held) StageSlots::get().release(slot_dev, slots_held);
    }
};

// grid for a round: x over pairs (grid-stride beyond the cap), y = sub-units in split mode
static uint64_t sc_split_max_pairs() {
    static const uint64_t v = [] {
        const char* e = getenv("GM_SC_SPLIT_MAX_LOG");  // tuning knob (development): log2 of the largest split-mode round
        return e ? (1ull << atoi(e)) : (1ull << 14);
    }();
    return v;
}
#define SC_SPLIT_MAX_PAIRS sc_split_max_pairs()
static dim3 round_grid(uint64_t npairs, int ny) {
    uint64_t bx = (npairs + SC_THREADS - 1) / SC_THREADS;
    if (bx < 1) bx = 1;
    const uint64_t cap = SC_MAX_BLOCKS / (uint64_t)(ny > 0 ? ny : 1);
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)ny);
}

// lean-kernel dispatch (large rounds of single-primitive layers)
static int lean_prim_of(const SegPlan& sp) {
    if (sp.nseg == 1 && sp.seg[0].out0 == 0) {
        switch (sp.seg[0].prim) {
            case FN_AFF_L1: case FN_AFF_L2: case FN_AFF_L3: case FN_PROJ_L1: case FN_PROJ_L2: case FN_PROJ_L3:
            case FN_PT_BIT_CHOICE: case FN_ADD_INVERSES: case FN_LOGUP_LAYER:
                for (int q = 0; q < sp.seg[0].n_in; q++) if (sp.seg[0].in[q] != q) return 0;
                return sp.seg[0].prim;
            default: return 0;
        }
    }
    if (sp.nseg == 3 && sp.seg[0].prim == FN_AFF_L1 && sp.seg[1].prim == FN_BITCHECK && sp.seg[2].prim == FN_BITCHECK &&
        sp.seg[0].out0 == 0 && sp.seg[1].in[0] == 4 && sp.seg[2].in[0] == 5 && sp.seg[1].out0 == 3 && sp.seg[2].out0 == 4)
        return LEAN_AFF_L1_BC;
    return 0;
}

template <bool VECVEC>
static int32_t launch_deg2_lean(int prim, dim3 grid, hipStream_t s, const LeanCols& lc, const Fr* eq, const Fr* gp, uint64_t npairs,
                                const VVArgs& va, const FinishCtx& fc) {
#define GM_LEAN_CASE(P)                                                                                                  \
    case P: hipLaunchKernelGGL((k_round_deg2_lean<P, VECVEC>), grid, dim3(SC_THREADS), 0, s, lc, eq, gp, npairs, va, fc); break;
    static const bool use9 = [] { const char* e = getenv("GM_LEAN_FR9"); return !(e && e[0] == '0'); }();
    static const bool usex2 = [] { const char* e = getenv("GM_LEAN_X2"); return !(e && e[0] == '0'); }();
    if (use9 && usex2 && lean9_has(prim)) {
#define GM_LEAN9X2_CASE(P)                                                                                               \
    case P: hipLaunchKernelGGL((k_round_deg2_lean9x2<P, VECVEC>), grid, dim3(SC_THREADS), 0, s, lc, eq, gp, npairs, va, fc); break;
        switch (prim) {
            GM_LEAN9X2_CASE(FN_AFF_L1) GM_LEAN9X2_CASE(FN_AFF_L3) GM_LEAN9X2_CASE(FN_PROJ_L1) GM_LEAN9X2_CASE(FN_PROJ_L2)
            GM_LEAN9X2_CASE(FN_PROJ_L3) GM_LEAN9X2_CASE(LEAN_AFF_L1_BC) GM_LEAN9X2_CASE(FN_AFF_L2) GM_LEAN9X2_CASE(FN_PT_BIT_CHOICE)
            GM_LEAN9X2_CASE(FN_ADD_INVERSES) GM_LEAN9X2_CASE(FN_LOGUP_LAYER)
        }
#undef GM_LEAN9X2_CASE
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    if (use9 && lean9_has(prim)) {
#define GM_LEAN9_CASE(P)                                                                                                 \
    case P: hipLaunchKernelGGL((k_round_deg2_lean9<P, VECVEC>), grid, dim3(SC_THREADS), 0, s, lc, eq, gp, npairs, va, fc); break;
        switch (prim) {
            GM_LEAN9_CASE(FN_AFF_L1) GM_LEAN9_CASE(FN_AFF_L3) GM_LEAN9_CASE(FN_PROJ_L1) GM_LEAN9_CASE(FN_PROJ_L2)
            GM_LEAN9_CASE(FN_PROJ_L3) GM_LEAN9_CASE(LEAN_AFF_L1_BC) GM_LEAN9_CASE(FN_AFF_L2) GM_LEAN9_CASE(FN_PT_BIT_CHOICE)
            GM_LEAN9_CASE(FN_ADD_INVERSES) GM_LEAN9_CASE(FN_LOGUP_LAYER)
        }
#undef GM_LEAN9_CASE
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    switch (prim) {
        GM_LEAN_CASE(FN_AFF_L1) GM_LEAN_CASE(FN_AFF_L2) GM_LEAN_CASE(FN_AFF_L3) GM_LEAN_CASE(FN_PROJ_L1)
        GM_LEAN_CASE(FN_PROJ_L2) GM_LEAN_CASE(FN_PROJ_L3) GM_LEAN_CASE(FN_PT_BIT_CHOICE) GM_LEAN_CASE(LEAN_AFF_L1_BC)
        GM_LEAN_CASE(FN_ADD_INVERSES) GM_LEAN_CASE(FN_LOGUP_LAYER)
        default: return set_err(GM_ERR_STATE, "no lean kernel for primitive %d", prim);
    }
#undef GM_LEAN_CASE
    GM_LAUNCH_CHECK();
    return GM_OK;
}

static bool launch_deg2_lean_split(int prim, dim3 grid, hipStream_t s, const LeanCols& lc, const Fr* eq, const Fr* gp, const VVArgs& va,
                                   const FinishCtx& fc) {
    static const bool off = [] { const char* e = getenv("GM_LEAN_SPLIT"); return e && e[0] == '0'; }();
    if (off) return false;
    static const bool form9 = [] { const char* e = getenv("GM_LEAN_SPLIT9"); return !(e && e[0] == '0'); }();   // A/B: the 8 x 32 form
    if (form9) {
        const dim3 g3(grid.x, 3);   // row 2: the tail weight
#define GM_LS9_CASE(P) \
    case P: hipLaunchKernelGGL((k_round_deg2_lean9_split<P>), g3, dim3(SC_THREADS), 0, s, lc, eq, gp, va, fc); return true;
        switch (prim) {
            GM_LS9_CASE(FN_AFF_L1) GM_LS9_CASE(FN_AFF_L2) GM_LS9_CASE(FN_AFF_L3) GM_LS9_CASE(FN_PROJ_L1) GM_LS9_CASE(FN_PROJ_L2)
            GM_LS9_CASE(FN_PROJ_L3) GM_LS9_CASE(LEAN_AFF_L1_BC)
            default: break;
        }
#undef GM_LS9_CASE
    }
#define GM_LS_CASE(P) \
    case P: hipLaunchKernelGGL((k_round_deg2_lean_split<P>), grid, dim3(SC_THREADS), 0, s, lc, eq, gp, va, fc); return true;
    switch (prim) {
        GM_LS_CASE(FN_AFF_L1) GM_LS_CASE(FN_AFF_L2) GM_LS_CASE(FN_AFF_L3) GM_LS_CASE(FN_PROJ_L1) GM_LS_CASE(FN_PROJ_L2) GM_LS_CASE(FN_PROJ_L3)
        GM_LS_CASE(LEAN_AFF_L1_BC)
        default: return false;
    }
#undef GM_LS_CASE
}

static int32_t launch_generic3_lean(int prim, dim3 grid, hipStream_t s, const LeanCols& lc, const Fr* gp, uint64_t npairs,
                                    const FinishCtx& fc) {
#define GM_LEAN_CASE(P)                                                                                                  \
    case P: hipLaunchKernelGGL((k_round_generic3_lean<P>), grid, dim3(SC_THREADS), 0, s, lc, gp, npairs, fc); break;
    switch (prim) {
        GM_LEAN_CASE(FN_AFF_L1) GM_LEAN_CASE(FN_AFF_L2) GM_LEAN_CASE(FN_AFF_L3) GM_LEAN_CASE(FN_PROJ_L1)
        GM_LEAN_CASE(FN_PROJ_L2) GM_LEAN_CASE(FN_PROJ_L3) GM_LEAN_CASE(FN_PT_BIT_CHOICE)
        GM_LEAN_CASE(FN_ADD_INVERSES) GM_LEAN_CASE(FN_LOGUP_LAYER)
        default: return set_err(GM_ERR_STATE, "no lean kernel for primitive %d", prim);
    }
#undef GM_LEAN_CASE
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// ---- profiler (gm_sc_profile): the large round kernels timed with HIP events on their launch stream --------------------
// mode 1: every large (non-split) round-kernel launch is bracketed by two events; exact pair counts of sparse launches are
//         copied from the device (off[nrows], 4 bytes, asynchronous).  ~300 event records per proof at config B: cheap enough
//         to stay on during timed runs.
// mode 2: additionally every small round and every fold is accounted (algorithmic bytes only, no events).
// Rows are per kernel class; algorithmic bytes per SURVEY 8(d): a round kernel reads both cells of every pair of every input
// column (64 k bytes per pair; + 32 for the eq column of the generic object), a fold moves 96 bytes per output cell.
struct ScProf {
    int mode = 0;
    struct Rec {
        hipEvent_t e0, e1;
        int cls, k;
        uint64_t pairs;          // dense: exact; sparse: filled from *h_cells at read time
        uint32_t* h_cells;       // pinned slot receiving off[nrows] (cells), or nullptr
        int fr_mul_per_pair;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> free_events;
    uint32_t* h_slots = nullptr;  // pinned, SLOT_CAP words
    uint32_t n_slots = 0;
    static constexpr uint32_t SLOT_CAP = 8192;
    double small_round_bytes = 0, fold_bytes = 0, small_rounds = 0, folds = 0;
    hipEvent_t get_event() {
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
static ScProf& sc_prof() {
    static thread_local ScProf p;
    return p;
}
static const char* sc_class_name(int cls) {
    // cls = prim * 4 + variant; variant 0: k_round_deg2_lean dense, 1: k_round_deg2_lean VecVec, 2: k_round_generic3_lean, 3: k_round_prod3_lean
    static thread_local char buf[64];
    const int prim = cls >> 2, var = cls & 3;
    const char* pn = prim == FN_AFF_L1 ? "AFF_L1" : prim == FN_AFF_L2 ? "AFF_L2" : prim == FN_AFF_L3 ? "AFF_L3" : prim == FN_PROJ_L1 ? "PROJ_L1"
                   : prim == FN_PROJ_L2 ? "PROJ_L2" : prim == FN_PROJ_L3 ? "PROJ_L3" : prim == FN_PT_BIT_CHOICE ? "PT_BIT_CHOICE"
                   : prim == LEAN_AFF_L1_BC ? "AFF_L1+BITCHECK" : prim == FN_ADD_INVERSES ? "ADD_INVERSES" : prim == FN_LOGUP_LAYER ? "LOGUP_LAYER" : "?";
    if (var == 3) snprintf(buf, sizeof(buf), "k_round_prod3_lean");
    else if (var == 2) snprintf(buf, sizeof(buf), "k_round_generic3_lean<%s>", pn);
    else snprintf(buf, sizeof(buf), "k_round_deg2_lean<%s,%s>", pn, var == 1 ? "vecvec" : "dense");
    return buf;
}
// Fr multiplications of lean_gamma_eval<PRIM> (one evaluation of the gamma-combined layer function)
static int lean_eval_muls(int prim) {
    switch (prim) {   // after the round-3 re-association (lean_gamma_eval9)
        case FN_AFF_L1: return 5; case LEAN_AFF_L1_BC: return 9; case FN_AFF_L2: return 3; case FN_AFF_L3: return 5;
        case FN_PROJ_L1: return 7; case FN_PROJ_L2: return 5; case FN_PROJ_L3: return 5; case FN_ADD_INVERSES: return 2;
        case FN_LOGUP_LAYER: return 3; case FN_PT_BIT_CHOICE: return 2; default: return 0;
    }
}
// returns the record index to close with prof_end, or -1
static int prof_begin(hipStream_t s, int cls, int k, uint64_t pairs, const uint32_t* d_cells_word, int fr_mul_per_pair) {
    ScProf& p = sc_prof();
    if (p.mode < 1) return -1;
    ScProf::Rec r;
    r.e0 = p.get_event(); r.e1 = p.get_event();
    r.cls = cls; r.k = k; r.pairs = pairs; r.h_cells = nullptr; r.fr_mul_per_pair = fr_mul_per_pair;
    if (d_cells_word) {
        if (!p.h_slots && hipHostMalloc((void**)&p.h_slots, ScProf::SLOT_CAP * 4, hipHostMallocDefault) != hipSuccess) p.h_slots = nullptr;
        if (p.h_slots && p.n_slots < ScProf::SLOT_CAP) {
            r.h_cells = p.h_slots + p.n_slots++;
            (void)hipMemcpyAsync(r.h_cells, d_cells_word, 4, hipMemcpyDeviceToHost, s);
        }
    }
    (void)hipEventRecord(r.e0, s);
    p.recs.push_back(r);
    return (int)p.recs.size() - 1;
}
static void prof_end(hipStream_t s, int idx) {
    if (idx >= 0) (void)hipEventRecord(sc_prof().recs[idx].e1, s);
}
static inline void prof_small_round(double bytes) {
    ScProf& p = sc_prof();
    if (p.mode >= 2) { p.small_round_bytes += bytes; p.small_rounds += 1; }
}
static inline void prof_fold(double bytes) {
    ScProf& p = sc_prof();
    if (p.mode >= 2) { p.fold_bytes += bytes; p.folds += 1; }
}

// ---- columns with ping-pong fold buffers --------------------------------------------------------
struct FoldCols {
    int k = 0;
    std::vector<const Fr*> cur;            // current columns (input columns on round 0)
    std::vector<std::unique_ptr<DevBuf>> a, b;  // scratch: a holds len/2, b holds len/4
    bool cur_is_a = false, started = false;
    int32_t init(int k_, const Fr* const* in, uint64_t len_elems) {
        k = k_;
        cur.assign(in, in + k);
        for (int i = 0; i < k; i++) {
            a.emplace_back(new DevBuf());
            b.emplace_back(new DevBuf());
            int32_t rc = a.back()->alloc((size_t)(len_elems / 2 + 2) * sizeof(Fr));
            if (rc) return rc;
            rc = b.back()->alloc((size_t)(len_elems / 4 + 2) * sizeof(Fr));
            if (rc) return rc;
        }
        return GM_OK;
    }
    // destination columns of the next fold
    void next(std::vector<Fr*>* dst) {
        dst->resize(k);
        const bool to_a = !started || !cur_is_a;
        for (int i = 0; i < k; i++) (*dst)[i] = to_a ? a[i]->fr() : b[i]->fr();
    }
    void commit(const std::vector<Fr*>& dst) {
        cur_is_a = !started || !cur_is_a;
        started = true;
        for (int i = 0; i < k; i++) cur[i] = dst[i];
    }
};

// gamma powers on the device: g[o] = gamma^o   (make_gamma_pows, utils.rs:126-135)
// small host -> device uploads ride in kernel arguments: asynchronous, no staging copy, no sync
static int32_t upload_small(const Fr* v, size_t n, Fr* dst, hipStream_t s) {
    for (size_t base = 0; base < n; base += 48) {
        SmallVals sv;
        const int cnt = (int)((n - base < 48) ? n - base : 48);
        for (int i = 0; i < cnt; i++) sv.v[i] = v[base + i];
        hipLaunchKernelGGL(k_upload_small, dim3(1), dim3(64), 0, s, sv, cnt, dst + base);
        GM_LAUNCH_CHECK();
    }
    return GM_OK;
}
static int32_t upload_gamma(const std::vector<Fr>& gp, DevBuf* d, hipStream_t s) {
    int32_t rc = d->alloc(gp.size() * sizeof(Fr) + 32);
    if (rc) return rc;
    return upload_small(gp.data(), gp.size(), d->fr(), s);
}
// the same without a launch of its own: the eq-table launch that follows stores the powers (launch_eq_pair / launch_eq_sequence
// take them as `extra`); one launch less at the head of every layer, where the host's launch rate is what the device waits for
static int32_t alloc_gamma(const std::vector<Fr>& gp, DevBuf* d) { return d->alloc(gp.size() * sizeof(Fr) + 32); }

static std::vector<Fr> make_gamma_pows(const Fr& gamma, int count) {
    std::vector<Fr> g = {fr_one(), gamma};
    for (int i = 2; i < count; i++) g.push_back(fr_mul(g[i - 1], gamma));
    return g;
}

// ---- DenseSumcheckObjectSO (sumcheck.rs:237-347) ------------------------------------------------
// Sharded dense objects: once a rank's slice is down to 2^SHARD_GATHER_LOG elements the ranks exchange their slices
// (k columns x 2^T elements x 32 B per rank: 40 KB at T = 8, one host all-gather) and every rank finishes the remaining T + lg rounds
// on the whole 2^(T + lg) elements, unsharded: no exchange per round any more, and the persistent tail launch serves them (a sharded
// object runs ordinary pre-enqueued rounds: ~30 us each where the tail launch needs ~11).  T = 0 is the old behaviour (gather when
// one element per rank is left).  GM_SC_SHARD_GATHER_LOG overrides.
static uint32_t shard_gather_log() {
    static const uint32_t v = [] { const char* e = getenv("GM_SC_SHARD_GATHER_LOG"); return (uint32_t)(e ? atoi(e) : 8); }();
    return v > 12 ? 12u : v;
}
// all ranks' slices of n_loc elements per column, in rank (= global) order, as new device columns
static int32_t shard_gather_columns(const Shard& sh, const Fr* const* cur, int k, uint64_t n_loc, hipStream_t stream,
                                    std::vector<std::unique_ptr<DevBuf>>* owned, std::vector<const Fr*>* ptrs) {
    std::vector<Fr> mine((size_t)k * n_loc);
    for (int i = 0; i < k; i++) GM_HIP(hipMemcpyAsync(mine.data() + (size_t)i * n_loc, cur[i], n_loc * sizeof(Fr), hipMemcpyDeviceToHost, stream));
    GM_HIP(hipStreamSynchronize(stream));
    std::vector<char> all;
    int32_t rc = shard_all_gather(sh, mine.data(), mine.size() * sizeof(Fr), &all);
    if (rc) return rc;
    const Fr* a = reinterpret_cast<const Fr*>(all.data());
    std::vector<Fr> col((size_t)sh.world * n_loc);
    for (int i = 0; i < k; i++) {
        for (uint32_t r = 0; r < sh.world; r++) memcpy(col.data() + (size_t)r * n_loc, a + ((size_t)r * k + i) * n_loc, n_loc * sizeof(Fr));
        owned->emplace_back(new DevBuf());
        rc = owned->back()->alloc(col.size() * sizeof(Fr));
        if (rc) return rc;
        if (col.size() <= 48) {
            rc = upload_small(col.data(), col.size(), owned->back()->fr(), stream);
            if (rc) return rc;
        } else {
            GM_HIP(hipMemcpyAsync(owned->back()->p, col.data(), col.size() * sizeof(Fr), hipMemcpyHostToDevice, stream));
            GM_HIP(hipStreamSynchronize(stream));   // col is reused
        }
        ptrs->push_back(owned->back()->fr());
    }
    return GM_OK;
}

struct ScDensePipe {
    static bool enabled() {
        static const bool v = [] { const char* e = getenv("GM_SC_NO_PIPELINE"); return !(e && e[0] == '1'); }();
        return v;
    }
};

struct ScDense : gm_sc {
    int kind = 0;  // 0: EqWrapper(GammaWrapper(f, gamma)) with the eq column last; 1: Prod3
    SegPlan sp{};
    int D = 3;
    uint32_t num_vars = 0, round_idx = 0;
    FoldCols cols;
    DevBuf d_gamma;
    RoundScratch rs;
    Fr claim_;
    std::vector<Fr> cached;
    bool has_cached = false;
    std::vector<std::unique_ptr<DevBuf>> owned;  // columns built by a handover
    // sharded: the columns are this rank's contiguous slice of 2^loc_vars elements; rounds add up the ranks'
    // partial sums; when the slice is down to one element the ranks exchange it and finish the last lg rounds replicated
    Shard sh;
    uint32_t loc_vars = 0;

    Fr claim() const override { return claim_; }

    int32_t gather_cols() {
        const int k = cols.k;
        std::vector<const Fr*> ptrs;
        int32_t rc = shard_gather_columns(sh, cols.cur.data(), k, 1ull << loc_vars, stream, &owned, &ptrs);
        if (rc) return rc;
        const uint64_t n_all = (uint64_t)sh.world << loc_vars;
        cols = FoldCols();
        rc = cols.init(k, ptrs.data(), n_all);
        if (rc) return rc;
        loc_vars += sh.lg;
        sh = Shard();
        return GM_OK;
    }

    // ---- pre-enqueued small rounds, as in ScDenseDeg2
    uint32_t k_enq = 0;
    uint32_t k_seq[64] = {};
    bool fold_pending = false;
    uint32_t fold_ticket = 0;
    std::vector<Fr*> fold_dst;
    int32_t launch_round(const Fr* const* cur_cols, uint64_t npairs, uint32_t round) {
        ColPtrs cp;
        for (int i = 0; i < cols.k; i++) cp.p[i] = cur_cols[i];
        const bool split = npairs <= SC_SPLIT_MAX_PAIRS;
        const int ny = split ? D * (kind == 1 ? 1 : sp.nseg) : 1;
        const dim3 grid = round_grid(npairs, ny);
        FinishCtx fc;
        if (RoundScratch::dev_exchange(sh)) {
            int32_t rc = rs.ctx_dev(sh, &fc);
            if (rc) return rc;
        } else {
            fc = rs.ctx();
        }
        const int lean = (kind == 0 && D == 3 && !split && cols.k <= 7) ? lean_prim_of(sp) : 0;
        if (kind == 2) {
            FoldedCols fcols;
            for (int i = 0; i < cols.k; i++) fcols.p[i] = cur_cols[i];
            hipLaunchKernelGGL(k_round_folded_prod, round_grid(npairs, 1), dim3(SC_THREADS), 0, stream, fcols, cols.k / 2,
                               d_gamma.fr(), npairs, fc);
        } else if (kind == 1 && D == 3 && !split) {
            LeanCols lc;
            for (int i = 0; i < 3; i++) lc.p[i] = cur_cols[i];
            const int pi = prof_begin(stream, 3, 3, npairs, nullptr, 3 * 2);
            hipLaunchKernelGGL(k_round_prod3_lean, grid, dim3(SC_THREADS), 0, stream, lc, npairs, fc);
            prof_end(stream, pi);
        } else if (lean && lean != LEAN_AFF_L1_BC) {
            LeanCols lc;
            for (int i = 0; i < cols.k; i++) lc.p[i] = cur_cols[i];
            const int pi = prof_begin(stream, lean * 4 + 2, cols.k, npairs, nullptr, 3 * (lean_eval_muls(lean) + 1));
            int32_t rc = launch_generic3_lean(lean, grid, stream, lc, d_gamma.fr(), npairs, fc);
            prof_end(stream, pi);
            if (rc) return rc;
        } else if (D == 3 && split)
            hipLaunchKernelGGL((k_round_generic<3, true>), grid, dim3(SC_THREADS), 0, stream, kind, sp, cp, cols.k,
                               d_gamma.fr(), npairs, fc);
        else if (D == 3)
            hipLaunchKernelGGL((k_round_generic<3, false>), grid, dim3(SC_THREADS), 0, stream, kind, sp, cp, cols.k,
                               d_gamma.fr(), npairs, fc);
        else if (D == 2 && split)
            hipLaunchKernelGGL((k_round_generic<2, true>), grid, dim3(SC_THREADS), 0, stream, kind, sp, cp, cols.k,
                               d_gamma.fr(), npairs, fc);
        else if (D == 2)
            hipLaunchKernelGGL((k_round_generic<2, false>), grid, dim3(SC_THREADS), 0, stream, kind, sp, cp, cols.k,
                               d_gamma.fr(), npairs, fc);
        else
            return set_err(GM_ERR_INVALID, "unsupported degree %d", D);
        GM_LAUNCH_CHECK();
        k_seq[round & 63] = fc.seq;
        if (split || !(lean || (kind == 1 && D == 3))) prof_small_round(64.0 * cols.k * (double)npairs);
        return GM_OK;
    }
    ~ScDense() override {
        if (fold_pending) {  // never leave a waiting gate behind
            rs.publish(round_idx, fr_zero(), fold_ticket);
            (void)hipStreamSynchronize(stream);
        }
    }

    int32_t unipoly(std::vector<Fr>* coeffs) override {
        if (round_idx >= num_vars) return set_err(GM_ERR_STATE, "the protocol has already ended (sumcheck.rs:279)");
        if (!has_cached) {
            if (sh.comm && loc_vars <= shard_gather_log()) {
                int32_t rc = gather_cols();
                if (rc) return rc;
            }
            const uint64_t npairs = 1ull << (loc_vars - 1);
            const bool split = npairs <= SC_SPLIT_MAX_PAIRS;
            const bool piped = split && !sh.comm && ScDensePipe::enabled() && (rs.own_pinned || pinned_exclusive() || k_enq > round_idx);
            if (k_enq <= round_idx) {
                int32_t rc = launch_round(cols.cur.data(), npairs, round_idx);
                if (rc) return rc;
                k_enq = round_idx + 1;
            }
            if (piped && !fold_pending && round_idx + 1 < num_vars && k_enq == round_idx + 1) {
                // small rounds: enqueue this round's fold behind a gate (k_fold_gate) and the next round's kernel now
                cols.next(&fold_dst);
                ColPtrs ci;
                ColPtrsMut co;
                std::vector<const Fr*> cn(cols.k);
                for (int i = 0; i < cols.k; i++) { ci.p[i] = cols.cur[i]; co.p[i] = fold_dst[i]; cn[i] = fold_dst[i]; }
                fold_ticket = ++RoundScratch::ticket_counter();
                if (fold_ticket == 0) fold_ticket = ++RoundScratch::ticket_counter();
                Fr* d_t = reinterpret_cast<Fr*>(static_cast<char*>(rs.counter.p) + 64);
                const GateArgs ga = rs.gate_in_fold(round_idx, fold_ticket, (uint64_t)ceil_div(npairs, 256) * cols.k);
                if (ga.bar_slot) {
                    hipLaunchKernelGGL(k_dense_fold_gated, dim3(ceil_div(npairs, 256), cols.k), dim3(256), 0, stream, ci, co, npairs, ga);
                } else {
                    hipLaunchKernelGGL(k_fold_gate, dim3(1), dim3(64), 0, stream, rs.t_slot(round_idx), rs.ticket_word(), fold_ticket,
                                       rs.ticket_word() + 1, d_t, wait_timeout_ticks());
                    hipLaunchKernelGGL(k_dense_fold_dev, dim3(ceil_div(npairs, 256), cols.k), dim3(256), 0, stream, ci, co, npairs, d_t);
                }
                prof_fold(96.0 * cols.k * (double)npairs);
                GM_LAUNCH_CHECK();
                fold_pending = true;
                int32_t rc = launch_round(cn.data(), npairs >> 1, round_idx + 1);
                if (rc) return rc;
                k_enq = round_idx + 2;
            }
            Fr acc[4];
            const bool devx = RoundScratch::dev_exchange(sh);
            if (devx) {
                int32_t rc = rs.exchange(sh, D, stream);
                if (rc) return rc;
            }
            int32_t rc = rs.finish_seq(k_seq[round_idx & 63], D, stream, acc, !fold_pending);
            if (rc) return rc;
            if (rs.ticket_word()[1]) {
                rs.ticket_word()[1] = 0;   // the staging may be shared with later objects: report once
                return set_err(GM_ERR_STATE, "a pre-enqueued fold timed out waiting for its challenge (gm_set_wait_timeout_ms)");
            }
            if (sh.comm && !devx) {
                rc = shard_sum_fr(sh, acc, D);
                if (rc) return rc;
            }
            std::vector<Fr> total(D + 1);
            for (int s = 0; s < D; s++) total[s + 1] = acc[s];
            total[0] = fr_sub(claim_, total[1]);  // sumcheck.rs:325
            cached = unipoly_from_evals(total);
            has_cached = true;
        }
        *coeffs = cached;
        return GM_OK;
    }

    int32_t bind(const Fr& t) override {
        if (round_idx >= num_vars) return set_err(GM_ERR_STATE, "the protocol has already ended (sumcheck.rs:264)");
        if (!has_cached) return set_err(GM_ERR_STATE, "should evaluate unipoly before binding (sumcheck.rs:271)");
        if (fold_pending) {
            rs.publish(round_idx, t, fold_ticket);
            fold_pending = false;
            cols.commit(fold_dst);
        } else {
            std::vector<Fr*> dst;
            cols.next(&dst);
            const uint64_t n_out = 1ull << (loc_vars - 1);
            int32_t rc = launch_dense_fold(cols.cur.data(), dst.data(), cols.k, n_out, t, stream);
            if (rc) return rc;
            prof_fold(96.0 * cols.k * (double)n_out);
            cols.commit(dst);
        }
        round_idx++;
        loc_vars--;
        claim_ = evaluate_univar(cached, t);
        has_cached = false;
        return GM_OK;
    }

    int32_t final_evals(std::vector<Fr>* out) override {
        if (round_idx != num_vars) return set_err(GM_ERR_STATE, "can only call final evals after the last round (sumcheck.rs:338)");
        if (sh.comm) return set_err(GM_ERR_STATE, "sharded columns were never exchanged");
        return gather_finals(cols.cur.data(), cols.k, stream, out);
    }
};

// ---- DenseDeg2SumcheckObjectSO (dense_eq.rs:61-173) ---------------------------------------------
struct ScDenseDeg2 : gm_sc {
    SegPlan sp{};
    uint32_t num_vars = 0, round_idx = 0;
    FoldCols cols;
    std::vector<Fr> gamma_pows, point;
    DevBuf d_gamma, d_eq;           // eq levels 0..num_vars-1 packed: level i at offset 2^i - 1
    RoundScratch rs;
    Fr claim_, multiplier;
    std::vector<Fr> cached, inv_eq0;
    bool has_cached = false;
    Shard sh;                 // see ScDense
    uint32_t loc_vars = 0;
    uint64_t glob_off = 0;    // global index of the slice's first element at the current round
    std::vector<std::unique_ptr<DevBuf>> owned;

    Fr claim() const override { return claim_; }
    const Fr* eq_ext = nullptr;   // levels built by someone else (the VecVec object's row_eq_coefs scratch), packed like d_eq
    // Level i of eq_poly_sequence(point) at global index `off`.  A sharded object built by gm_sc_dense_deg2_create keeps only its own
    // slice of the levels its local rounds read (level i >= lg restricted to the rank = eq(point[0..lg), rank) x eq(point[lg..i], .):
    // 2^loc_vars entries instead of 2^num_vars) plus the lg small top levels of the replicated last rounds.
    // The whole levels kept are 0 .. eq_top - 1 with eq_top = lg + the early-gather threshold (shard_gather_log): the rounds every rank
    // finishes on the gathered columns read them.
    bool eq_sliced = false;
    uint32_t eq_lg = 0, eq_rank = 0, eq_top = 0;
    const Fr* eq_at(uint32_t i, uint64_t off) const {
        if (!eq_sliced) return (eq_ext ? eq_ext : d_eq.fr()) + ((1ull << i) - 1) + off;
        if (i >= eq_top) return d_eq.fr() + ((1ull << (i - eq_lg)) - 1) + (off - ((uint64_t)eq_rank << (i - eq_lg)));
        return d_eq.fr() + ((size_t)1 << (num_vars - eq_lg)) + ((1ull << i) - 1) + off;
    }

    int32_t gather_cols() {
        const int k = cols.k;
        std::vector<const Fr*> ptrs;
        int32_t rc = shard_gather_columns(sh, cols.cur.data(), k, 1ull << loc_vars, stream, &owned, &ptrs);
        if (rc) return rc;
        const uint64_t n_all = (uint64_t)sh.world << loc_vars;
        cols = FoldCols();
        rc = cols.init(k, ptrs.data(), n_all);
        if (rc) return rc;
        loc_vars += sh.lg;   // (the eq levels these rounds read, 0 .. loc_vars - 1 < eq_top, are whole: eq_at)
        glob_off = 0;
        sh = Shard();
        return GM_OK;
    }

    int32_t unipoly(std::vector<Fr>* coeffs) override {
        if (has_cached) return set_err(GM_ERR_STATE, "unipoly called twice without bind (dense_eq.rs:109-111)");
        if (round_idx >= num_vars) return set_err(GM_ERR_STATE, "the protocol has already ended");
        // rounds inside a running tail launch (its first round goes through unipoly_pipelined, which handles a launch that left at the
        // residency barrier; a launch adopted from a VecVec object has been running for rounds)
        if (tail_active && round_idx >= tail_r0 && (round_idx > tail_r0 || stage_adopted)) return unipoly_tail(coeffs);
        if (sh.comm && loc_vars <= shard_gather_log() && !fold_pending) {
            int32_t rc = gather_cols();
            if (rc) return rc;
        }
        const uint64_t npairs = 1ull << (loc_vars - 1);
        const Fr* eq_cur = eq_at(num_vars - 1 - round_idx, glob_off >> 1);
        ColPtrs cp;
        for (int i = 0; i < cols.k; i++) cp.p[i] = cols.cur[i];
        const bool split = npairs <= SC_SPLIT_MAX_PAIRS;
        const dim3 grid = round_grid(npairs, split ? 2 * sp.nseg : 1);
        const VVArgs none{nullptr, 0, nullptr, nullptr, nullptr};
        const int lean = (!split && cols.k <= 6) ? lean_prim_of(sp) : 0;
        // results of pre-enqueued kernels land in the pinned staging: it must be this object's alone for the duration
        static const bool pipe_large = [] { const char* e = getenv("GM_SC_PIPE_LARGE_DENSE"); return !(e && e[0] == '0'); }();   // A/B switch
        const bool devx = RoundScratch::dev_exchange(sh);
        // sharded with the round sums meeting on the host: the device side of a round is the unsharded one
        if ((split || pipe_large) && !devx && pipeline_enabled() && (rs.own_pinned || pinned_exclusive() || k_enq > round_idx))
            return unipoly_pipelined(coeffs, npairs, eq_cur, cp);
        FinishCtx fc0;
        if (devx) {
            int32_t rc = rs.ctx_dev(sh, &fc0);
            if (rc) return rc;
        } else {
            fc0 = rs.ctx();
        }
        if (lean) {
            LeanCols lc;
            for (int i = 0; i < cols.k; i++) lc.p[i] = cols.cur[i];
            const int pi = prof_begin(stream, lean * 4 + 0, cols.k, npairs, nullptr, 2 * lean_eval_muls(lean) + 2);
            int32_t rc = launch_deg2_lean<false>(lean, grid, stream, lc, eq_cur, d_gamma.fr(), npairs, none, fc0);
            prof_end(stream, pi);
            if (rc) return rc;
        } else if (split)
            hipLaunchKernelGGL((k_round_deg2<false, true>), grid, dim3(SC_THREADS), 0, stream, sp, cp,
                               eq_cur, d_gamma.fr(), npairs, none, fc0);
        else
            hipLaunchKernelGGL((k_round_deg2<false, false>), grid, dim3(SC_THREADS), 0, stream, sp, cp,
                               eq_cur, d_gamma.fr(), npairs, none, fc0);
        GM_LAUNCH_CHECK();
        if (!lean) prof_small_round(64.0 * cols.k * (double)npairs);
        if (devx) {
            int32_t rc = rs.exchange(sh, 2, stream);
            if (rc) return rc;
        }
        Fr acc[4];
        int32_t rc = rs.finish(2, stream, acc);
        if (rc) return rc;
        if (sh.comm && !devx) {
            rc = shard_sum_fr(sh, acc, 2);
            if (rc) return rc;
        }
        // full-length dense columns: sum of eq = 1, the trailing pad term (dense_eq.rs:141-146) vanishes
        const Fr total1 = fr_mul(acc[0], multiplier), total2 = fr_mul(acc[1], multiplier);
        if (inv_eq0.empty()) inv_eq0 = batch_inv_one_minus(point);  // first round: point is still complete
        cached = from12_inv(total1, total2, point.back(), inv_eq0[point.size() - 1], claim_);
        has_cached = true;
        *coeffs = cached;
        return GM_OK;
    }

    // ---- pre-enqueued small rounds (see k_fold_gate)
    uint32_t k_enq = 0;            // round kernels enqueued so far: rounds [0, k_enq)
    uint32_t k_seq[64] = {};       // result sequence number of the enqueued round kernels
    bool fold_pending = false;     // the fold of round `round_idx` is enqueued and waits for its challenge
    uint32_t fold_ticket = 0;
    std::vector<Fr*> fold_dst;
    static bool pipeline_enabled() {
        static const bool v = [] { const char* e = getenv("GM_SC_NO_PIPELINE"); return !(e && e[0] == '1'); }();
        return v;
    }
    // the round kernel of a pre-enqueued round, by size: split mode for small rounds, the lean kernel of a single-primitive layer or
    // the generic kernel for large ones (large rounds are pre-enqueued too: the fold and the next round kernel are then already
    // in the stream when the challenge arrives, ~10 us of launch latency per round)
    int32_t launch_small_round(const ColPtrs& cp, const Fr* eq, uint64_t npairs, uint32_t round) {
        const bool split = npairs <= SC_SPLIT_MAX_PAIRS;
        const dim3 grid = round_grid(npairs, split ? 2 * sp.nseg : 1);
        const VVArgs none{nullptr, 0, nullptr, nullptr, nullptr};
        const int lean = (!split && cols.k <= 6) ? lean_prim_of(sp) : 0;
        const FinishCtx fc = rs.ctx();
        if (lean) {
            LeanCols lc;
            for (int i = 0; i < cols.k; i++) lc.p[i] = cp.p[i];
            const int pi = prof_begin(stream, lean * 4 + 0, cols.k, npairs, nullptr, 2 * lean_eval_muls(lean) + 2);
            int32_t rc = launch_deg2_lean<false>(lean, grid, stream, lc, eq, d_gamma.fr(), npairs, none, fc);
            prof_end(stream, pi);
            if (rc) return rc;
        } else if (split) {
            hipLaunchKernelGGL((k_round_deg2<false, true>), grid, dim3(SC_THREADS), 0, stream, sp, cp, eq, d_gamma.fr(), npairs, none, fc);
        } else {
            hipLaunchKernelGGL((k_round_deg2<false, false>), grid, dim3(SC_THREADS), 0, stream, sp, cp, eq, d_gamma.fr(), npairs, none, fc);
        }
        GM_LAUNCH_CHECK();
        k_seq[round & 63] = fc.seq;
        if (!lean) prof_small_round(64.0 * cols.k * (double)npairs);
        return GM_OK;
    }
    // ---- persistent stage (see k_stage): rounds [tail_r0, num_vars) run inside one launch
    bool tail_active = false, tail_denied = false;   // denied: a launch was abandoned at its residency barrier, do not try again
    uint32_t tail_r0 = 0;
    std::shared_ptr<StageRun> stage;
    int stage_round() const { return stage->n_thin + (int)(round_idx - tail_r0); }
    bool stage_fits(uint32_t r0, uint64_t npairs0) const {
        return StageRun::fits(sp.nseg, 2 * npairs0, 0, (int)(num_vars - r0));
    }
    // cp: the columns as they are at round r0 (npairs0 pairs); everything before it in the stream has been enqueued
    int32_t launch_tail(const ColPtrs& cp, uint32_t r0, uint64_t npairs0, bool may_wait = true) {
        StageArgs a;
        memset(&a, 0, sizeof(a));
        const int nr = (int)(num_vars - r0);
        if (nr < 1 || nr > STAGE_MAX_ROUNDS || (1ull << (nr - 1)) != npairs0) return set_err(GM_ERR_STATE, "stage rounds: inconsistent shape");
        const uint64_t g0 = glob_off >> (r0 - round_idx);   // glob_off at round r0
        for (int q = 0; q < nr; q++) a.eq[q] = eq_at(num_vars - 1 - (r0 + q), g0 >> (q + 1));
        a.n_elems = (uint32_t)(2 * npairs0);
        const int hr = stage_host_rounds(sp, nr);
        a.n_dense = nr - hr;
        stage.reset(new StageRun());
        int32_t rc = stage->launch(sp, cp, d_gamma.fr(), a, stream, may_wait);
        if (rc) { stage.reset(); return rc; }
        host_r0 = r0 + (uint32_t)(nr - hr);
        for (int q = 0; q < nr; q++) {   // rounds and folds that happen inside the launch
            prof_small_round(64.0 * cols.k * (double)(npairs0 >> q));
            prof_fold(96.0 * cols.k * (double)(npairs0 >> q));
        }
        tail_active = true;
        tail_r0 = r0;
        return GM_OK;
    }
    // the dense stage of a VecVec object whose k_stage launch is already running (bind_into_dense inside the kernel)
    bool stage_adopted = false;
    void adopt_stage(const std::shared_ptr<StageRun>& run) {
        stage = run;
        stage_adopted = true;
        tail_active = true;
        tail_r0 = 0;
        host_r0 = (uint32_t)run->n_dense;
        k_enq = num_vars;
    }
    // ---- the last rounds on the host (see stage_host_rounds): rounds [host_r0, num_vars) over the elements the launch left
    uint32_t host_r0 = 0xffffffffu;
    std::vector<std::vector<Fr>> hcols;
    bool host_round() const { return tail_active && round_idx >= host_r0; }
    int32_t host_collect() {
        if (!hcols.empty()) return GM_OK;
        int32_t rc = stage->collect(cols.k, &hcols);
        if (rc || !sh.comm) return rc;
        // sharded: the launch left this rank's slice of every column; the rounds the host finishes run on all of them, replicated
        const size_t n = hcols[0].size(), kk = (size_t)cols.k;
        std::vector<Fr> mine(kk * n);
        for (size_t c = 0; c < kk; c++) memcpy(mine.data() + c * n, hcols[c].data(), n * sizeof(Fr));
        std::vector<char> all;
        rc = shard_all_gather(sh, mine.data(), kk * n * sizeof(Fr), &all);
        if (rc) return rc;
        const Fr* a = reinterpret_cast<const Fr*>(all.data());
        for (size_t c = 0; c < kk; c++) {
            hcols[c].resize((size_t)sh.world * n);
            for (uint32_t r = 0; r < sh.world; r++) memcpy(hcols[c].data() + (size_t)r * n, a + ((size_t)r * kk + c) * n, n * sizeof(Fr));
        }
        sh = Shard();
        return GM_OK;
    }
    int32_t host_round_sums(Fr* s1, Fr* s2) {
        int32_t rc = host_collect();
        if (rc) return rc;
        const size_t np = hcols[0].size() / 2;
        if (np < 1 || (np << 1) != hcols[0].size() || (size_t)1 << (point.size() - 1) != np)
            return set_err(GM_ERR_STATE, "host rounds: %zu elements left for %zu variables", hcols[0].size(), point.size());
        // eq(point[0 .. n-1), .): point[0] is the most significant variable (utils.rs:222-250)
        std::vector<Fr> eq(1, fr_one()), nx;
        for (size_t v = 0; v + 1 < point.size(); v++) {
            nx.resize(2 * eq.size());
            for (size_t j = 0; j < eq.size(); j++) {
                const Fr m = fr_mul(point[v], eq[j]);
                nx[2 * j] = fr_sub(eq[j], m);
                nx[2 * j + 1] = m;
            }
            eq.swap(nx);
        }
        Fr in1[GM_MAX_COLS], in2[GM_MAX_COLS], o1[GM_MAX_COLS], o2[GM_MAX_COLS];
        *s1 = fr_zero(); *s2 = fr_zero();
        for (size_t i = 0; i < np; i++) {
            for (int c = 0; c < cols.k; c++) {
                in1[c] = hcols[c][2 * i + 1];
                in2[c] = fr_sub(fr_dbl(in1[c]), hcols[c][2 * i]);
            }
            seg_plan_exec_host(sp, in1, o1);
            seg_plan_exec_host(sp, in2, o2);
            Fr a1 = o1[0], a2 = o2[0];
            for (int o = 1; o < sp.n_outs; o++) {
                a1 = fr_add(a1, fr_mul(gamma_pows[o], o1[o]));
                a2 = fr_add(a2, fr_mul(gamma_pows[o], o2[o]));
            }
            *s1 = fr_add(*s1, fr_mul(eq[i], a1));
            *s2 = fr_add(*s2, fr_mul(eq[i], a2));
        }
        return GM_OK;
    }
    void host_fold(const Fr& t) {
        for (auto& col : hcols) {
            const size_t n = col.size() / 2;
            for (size_t i = 0; i < n; i++) col[i] = fr_add(col[2 * i], fr_mul(t, fr_sub(col[2 * i + 1], col[2 * i])));
            col.resize(n);
        }
    }
    int32_t tail_round_sums(Fr* s1, Fr* s2) {
        if (host_round()) return host_round_sums(s1, s2);
        int32_t rc = stage->sums(stage_round(), s1, s2, nullptr);
        if (rc || !sh.comm) return rc;
        Fr v[2] = {*s1, *s2};   // sharded: the launch summed this rank's slice
        rc = shard_sum_fr(sh, v, 2);
        *s1 = v[0]; *s2 = v[1];
        return rc;
    }
    void tail_publish(const Fr& t) {
        if (host_round()) host_fold(t);
        else stage->publish(stage_round(), t);
    }

    // a round inside the tail launch (or one of the last rounds, on the host)
    int32_t unipoly_tail(std::vector<Fr>* coeffs) {
        Fr a1, a2;
        int32_t rc = tail_round_sums(&a1, &a2);
        if (rc) return rc;
        const Fr total1 = fr_mul(a1, multiplier), total2 = fr_mul(a2, multiplier);
        if (inv_eq0.empty()) inv_eq0 = batch_inv_one_minus(point);
        cached = from12_inv(total1, total2, point.back(), inv_eq0[point.size() - 1], claim_);
        has_cached = true;
        *coeffs = cached;
        return GM_OK;
    }
    int32_t unipoly_pipelined(std::vector<Fr>* coeffs, uint64_t npairs, const Fr* eq_cur, const ColPtrs& cp) {
        const uint32_t r = round_idx;
        // (a sharded object keeps to pre-enqueued rounds: whether a tail launch runs would have to be agreed between the ranks, as the
        // VecVec object does for its stage; sharded dense objects of the image part are the dense stages of VecVec layers, which
        // are inside that launch already)
        const bool tail_ok = !tail_denied && !sh.comm && stage_enabled() && sp.nseg <= 32 && (rs.own_pinned || pinned_exclusive());
        if (!tail_active && tail_ok && stage_fits(r, npairs) && k_enq <= r) {
            int32_t rc = launch_tail(cp, r, npairs);   // the object starts small: everything runs in the tail
            if (rc) return rc;
            k_enq = num_vars;
        }
        if (tail_active && r >= tail_r0) {
            Fr a1, a2;
            int32_t rc = tail_round_sums(&a1, &a2);
            if (rc == GM_STAGE_NOT_RESIDENT && r == tail_r0) {
                // the launch left before its first round (residency barrier): this and the following rounds as ordinary kernels
                stage.reset();
                tail_active = false;
                tail_denied = true;
                host_r0 = 0xffffffffu;
                k_enq = r;
                return unipoly_pipelined(coeffs, npairs, eq_cur, cp);
            }
            if (rc) return rc;
            const Fr total1 = fr_mul(a1, multiplier), total2 = fr_mul(a2, multiplier);
            if (inv_eq0.empty()) inv_eq0 = batch_inv_one_minus(point);
            cached = from12_inv(total1, total2, point.back(), inv_eq0[point.size() - 1], claim_);
            has_cached = true;
            *coeffs = cached;
            return GM_OK;
        }
        if (k_enq <= r) {  // the first small round of this object: nothing was enqueued ahead
            int32_t rc = launch_small_round(cp, eq_cur, npairs, r);
            if (rc) return rc;
            k_enq = r + 1;
        }
        // (sharded: the round at which the ranks gather their slices is not enqueued ahead -- its kernel runs on the gathered columns)
        if (!fold_pending && (sh.comm ? loc_vars > shard_gather_log() + 1 : loc_vars >= 2) && k_enq == r + 1) {
            // enqueue fold r (waiting for t_r) and round kernel r + 1 while round r is still running
            cols.next(&fold_dst);
            ColPtrs ci;
            ColPtrsMut co;
            ColPtrs cn;
            for (int i = 0; i < cols.k; i++) { ci.p[i] = cols.cur[i]; co.p[i] = fold_dst[i]; cn.p[i] = fold_dst[i]; }
            fold_ticket = ++RoundScratch::ticket_counter();
            if (fold_ticket == 0) fold_ticket = ++RoundScratch::ticket_counter();
            const uint64_t n_out = npairs;
            Fr* d_t = reinterpret_cast<Fr*>(static_cast<char*>(rs.counter.p) + 64);
            const GateArgs ga = rs.gate_in_fold(r, fold_ticket, (uint64_t)ceil_div(n_out, 256) * cols.k, sh.comm != nullptr);
            if (ga.bar_slot) {
                hipLaunchKernelGGL(k_dense_fold_gated, dim3(ceil_div(n_out, 256), cols.k), dim3(256), 0, stream, ci, co, n_out, ga);
            } else {
                hipLaunchKernelGGL(k_fold_gate, dim3(1), dim3(64), 0, stream, rs.t_slot(r), rs.ticket_word(), fold_ticket, rs.ticket_word() + 1, d_t,
                                   wait_timeout_ticks());
                hipLaunchKernelGGL(k_dense_fold_dev, dim3(ceil_div(n_out, 256), cols.k), dim3(256), 0, stream, ci, co, n_out, d_t);
            }
            prof_fold(96.0 * cols.k * (double)n_out);
            GM_LAUNCH_CHECK();
            fold_pending = true;
            bool staged = false;
            if (tail_ok && stage_fits(r + 1, npairs >> 1)) {   // everything after this fold runs in one launch
                // this round's gate is in the stream, waiting for this thread: try only (StageSlots)
                int32_t rc = launch_tail(cn, r + 1, npairs >> 1, false);
                if (rc && rc != GM_STAGE_BUSY) return rc;
                staged = rc == GM_OK;
            }
            if (staged) {
                k_enq = num_vars;
            } else {
                const Fr* eq_next = eq_at(num_vars - 2 - r, glob_off >> 2);
                int32_t rc = launch_small_round(cn, eq_next, npairs >> 1, r + 1);
                if (rc) return rc;
                k_enq = r + 2;
            }
        }
        Fr acc[4];
        int32_t rc = rs.finish_seq(k_seq[r & 63], 2, stream, acc, !fold_pending);
        if (rc) return rc;
        if (rs.ticket_word()[1]) {
                rs.ticket_word()[1] = 0;   // the staging may be shared with later objects: report once
                return set_err(GM_ERR_STATE, "a pre-enqueued fold timed out waiting for its challenge (gm_set_wait_timeout_ms)");
            }
        if (sh.comm) {
            rc = shard_sum_fr(sh, acc, 2);
            if (rc) return rc;
        }
        const Fr total1 = fr_mul(acc[0], multiplier), total2 = fr_mul(acc[1], multiplier);
        if (inv_eq0.empty()) inv_eq0 = batch_inv_one_minus(point);
        cached = from12_inv(total1, total2, point.back(), inv_eq0[point.size() - 1], claim_);
        has_cached = true;
        *coeffs = cached;
        return GM_OK;
    }
    ~ScDenseDeg2() override {
        if (fold_pending) {   // never leave a waiting kernel behind (a gate here; the stage launch releases its own waiters)
            rs.publish(round_idx, fr_zero(), fold_ticket);
            (void)hipStreamSynchronize(stream);
        }
    }

    int32_t bind(const Fr& t) override {
        if (!has_cached) return set_err(GM_ERR_STATE, "bind before unipoly (dense_eq.rs:105 unwrap)");
        multiplier = fr_mul(multiplier, eq_bind_factor(point.back(), t));
        if (tail_active && round_idx >= tail_r0) {   // the fold happens inside the tail kernel
            tail_publish(t);
            point.pop_back();
            round_idx++;
            loc_vars--;
            glob_off >>= 1;
            claim_ = evaluate_univar(cached, t);
            has_cached = false;
            return GM_OK;
        }
        if (fold_pending) {
            rs.publish(round_idx, t, fold_ticket);   // the waiting fold and the next round kernel take it from here
            fold_pending = false;
            cols.commit(fold_dst);
            point.pop_back();
            round_idx++;
            loc_vars--;
            glob_off >>= 1;
            claim_ = evaluate_univar(cached, t);
            has_cached = false;
            return GM_OK;
        }
        std::vector<Fr*> dst;
        cols.next(&dst);
        const uint64_t n_out = 1ull << (loc_vars - 1);
        int32_t rc = launch_dense_fold(cols.cur.data(), dst.data(), cols.k, n_out, t, stream);
        if (rc) return rc;
        prof_fold(96.0 * cols.k * (double)n_out);
        cols.commit(dst);
        point.pop_back();
        round_idx++;
        loc_vars--;
        glob_off >>= 1;
        claim_ = evaluate_univar(cached, t);
        has_cached = false;
        return GM_OK;
    }

    int32_t final_evals(std::vector<Fr>* out) override {
        if (tail_active) {
            if (round_idx != num_vars) return set_err(GM_ERR_STATE, "final_evals before the last round");
            int32_t rc = host_collect();   // one element per column is left (after the device's or the host's last fold)
            if (rc) return rc;
            out->resize(cols.k);
            for (int c = 0; c < cols.k; c++) {
                if (hcols[c].size() != 1) return set_err(GM_ERR_STATE, "final_evals: %zu elements left", hcols[c].size());
                (*out)[c] = hcols[c][0];
            }
            return GM_OK;
        }
        return gather_finals(cols.cur.data(), cols.k, stream, out);
    }
};

// ---- VecVecDeg2SumcheckObjectSO (vecvec_eq.rs:72-398) -------------------------------------------
struct ScVecVecDeg2 : gm_sc {
    SegPlan sp{};
    GmFn fn{};
    uint32_t nrows = 0, col_logsize = 0, row_logsize = 0;  // row_logsize shrinks with every sparse bind
    uint32_t n_row_vars0 = 0;                               // row variables at creation
    uint32_t already_bound = 0;
    int k = 0;
    std::vector<const Fr*> cur;
    std::vector<std::unique_ptr<DevBuf>> bufA, bufB;
    bool cur_is_a = false, started = false;
    const uint32_t* off_cur = nullptr;
    DevBuf off_all;  // row layouts of all sparse rounds: table l at off_tab + l * (nrows + 1), table 0 = the input layout
    const uint32_t* off_tab = nullptr;          // = off_all, or the shape's own table of layouts (gm_vv::off_levels)
    std::shared_ptr<DevBuf> off_keep;
    // coarse row tables of the shape's layouts (gm_vv::coarse): table of layout l at coarse_base + coarse_off[l0 + l]
    std::shared_ptr<DevBuf> coarse_keep;
    std::shared_ptr<std::vector<uint64_t>> coarse_off;
    uint32_t coarse_l0 = 0;
    const uint32_t* coarse_for(const uint32_t* off) const {
        if (!coarse_keep || !off_tab || off < off_tab) return nullptr;
        const uint64_t l = (uint64_t)(off - off_tab) / (nrows + 1);
        if ((uint64_t)(off - off_tab) % (nrows + 1) != 0 || coarse_l0 + l >= coarse_off->size()) return nullptr;
        return reinterpret_cast<const uint32_t*>(coarse_keep->p) + (*coarse_off)[coarse_l0 + l];
    }
    uint32_t cap_a = 0, cap_b = 0;
    std::vector<Fr> row_pad, col_pad, gamma_pows, point;
    int binding_var_idx = 0;
    uint32_t padded_vars = 0;  // leading row variables every row is shorter than (EQPolyPointParts)
    DevBuf d_gamma, d_row_coef, d_eq_seq, d_prefix;
    std::vector<uint64_t> eq_level_off;  // offset of level i of the padded eq sequence inside d_eq_seq
    std::vector<uint32_t> eq_level_len;
    Fr row_coef_tail_nrows = fr_zero();  // row_eq_coefs_tail_sums[nrows] (host): the only entry the rounds read
    RoundScratch rs;
    Fr claim_, multiplier;
    std::vector<Fr> cached, inv_eq0;
    bool has_cached = false;
    std::unique_ptr<ScDense> dense;
    std::unique_ptr<gm_sc> dense2;  // the dense stage through the eq-factored object (see bind_into_dense)
    Fr dense2_eq_final() const;
    Shard sh;                // sharded: this object holds the rows row_base .. row_base + nrows of the global polynomials
    uint32_t row_base = 0;

    Fr claim() const override { return dense2 ? dense2->claim() : dense ? dense->claim() : claim_; }

    int32_t unipoly(std::vector<Fr>* coeffs) override {
        if (dense2) return dense2->unipoly(coeffs);
        if (dense) return dense->unipoly(coeffs);
        if (has_cached) return set_err(GM_ERR_STATE, "unipoly called twice without bind (vecvec_eq.rs:305-307)");
        Fr acc[4];
        const bool devx = RoundScratch::dev_exchange(sh);
        const bool shard_host = sh.comm && !devx;   // sharded, the round sums meet on the host: rounds run exactly as unsharded ones
        // Sharded: whether the stage kernel takes over must be the SAME decision on every rank (the exchanges of a staged layer and of
        // an ordinary one differ).  It is taken once, at the first thin round -- the same round everywhere: the longest row of the WHOLE
        // polynomial decides it -- by exchanging one word: a rank whose launch could not be made or did not become resident says so
        // and everybody runs the layer's remaining rounds as ordinary kernels.
        const bool deciding = shard_host && !stage_decided && cur_max_len == 2 && stage_shape_ok();
        if (!stage_active && k_enq <= already_bound && cur_max_len == 2 && stage_ok() && (!shard_host || deciding)) {
            // every row is down to one pair: this round, the rest of the sparse stage and the whole dense stage run in one launch.
            // A sharded launch only TRIES for its share of the device's co-residency budget: ranks that are threads of one process
            // share that budget, and the rank holding it waits for this one's round sums -- waiting here would close the cycle.
            // "Busy" is one more way of "could not" for the agreement below.
            int32_t rc = launch_stage(cur.data(), off_cur, already_bound, !shard_host);
            if (rc && !(shard_host && rc == GM_STAGE_BUSY)) return rc;
        }
        bool healthy = false;
        if (stage_active) {
            int32_t rc = stage->sums((int)(already_bound - stage_r0), &acc[0], &acc[1], &acc[2]);
            if (rc == GM_STAGE_NOT_RESIDENT && already_bound == stage_r0) {
                // the launch left before its first round (residency barrier): this and the following rounds as ordinary kernels
                stage.reset();
                stage_active = false;
                stage_denied = true;
                k_enq = already_bound;
            } else if (rc) return rc;
            else healthy = true;
        }
        if (deciding) {
            stage_decided = true;
            std::vector<char> all;
            const uint32_t mine = healthy ? 1u : 0u;
            int32_t rc = shard_all_gather(sh, &mine, sizeof(uint32_t), &all);
            if (rc) return rc;
            bool everybody = true;
            for (uint32_t r = 0; r < sh.world; r++) {
                uint32_t v;
                memcpy(&v, all.data() + 4 * (size_t)r, 4);
                everybody = everybody && v == 1u;
            }
            if (!everybody) {
                if (stage_active) {   // another rank could not: leave the launch (its blocks are released; the columns were only read)
                    g_stage_left++;
                    stage.reset();
                    stage_active = false;
                    k_enq = already_bound;
                }
                stage_denied = true;
            }
        }
        if (stage_active && shard_host) {
            int32_t rc = shard_sum_fr(sh, acc, 3);
            if (rc) return rc;
        }
        if (!stage_active) {
        // every sparse round (large ones too: the fold and the next round kernel are then already in the stream when the
        // challenge arrives, ~8 us of launch latency per round) enqueues its fold behind a gate and the next round's kernel
        const bool piped = !devx && ScDenseDeg2::pipeline_enabled() &&
                           (rs.own_pinned || pinned_exclusive() || k_enq > already_bound);
        if (k_enq <= already_bound) {
            int32_t rc = launch_sparse_round(cur.data(), off_cur, cells_bound, already_bound);
            if (rc) return rc;
            k_enq = already_bound + 1;
        }
        if (piped && !fold_pending && k_enq == already_bound + 1 && (uint32_t)binding_var_idx > col_logsize &&
            already_bound + 1 < n_off_tables && k <= 16) {
            // enqueue the fold of this round (behind a gate that waits for t) and the next round's kernel now
            nx_to_a = !started || !cur_is_a;
            nx_off = off_tab + (uint64_t)(already_bound + 1) * (nrows + 1);
            nx_bound = cells_bound / 2 + nrows;
            ColPtrs ci;
            ColPtrsMut co;
            PadCols pd;
            nx_cur.resize(k);
            for (int i = 0; i < k; i++) {
                ci.p[i] = cur[i];
                co.p[i] = nx_to_a ? bufA[i]->fr() : bufB[i]->fr();
                pd.v[i] = row_pad[i];
                nx_cur[i] = co.p[i];
            }
            fold_ticket = ++RoundScratch::ticket_counter();
            if (fold_ticket == 0) fold_ticket = ++RoundScratch::ticket_counter();
            Fr* d_t = reinterpret_cast<Fr*>(static_cast<char*>(rs.counter.p) + 64);
            const GateArgs ga = rs.gate_in_fold(already_bound, fold_ticket, (uint64_t)ceil_div(nx_bound, SC_THREADS) * ((k + 1) / 2), sh.comm != nullptr);
            if (!ga.bar_slot)
                hipLaunchKernelGGL(k_fold_gate, dim3(1), dim3(64), 0, stream, rs.t_slot(already_bound), rs.ticket_word(), fold_ticket,
                                   rs.ticket_word() + 1, d_t, wait_timeout_ticks());
            hipLaunchKernelGGL(k_vv_fold, dim3(ceil_div(nx_bound, SC_THREADS), (k + 1) / 2), dim3(SC_THREADS), 0, stream, ci, co, off_cur,
                               nx_off, nrows, fr_zero(), pd, (const Fr*)d_t, k, coarse_for(nx_off), ga);
            GM_LAUNCH_CHECK();
            prof_fold(96.0 * k * (double)(cells_bound / 2));
            fold_pending = true;
            const uint32_t half = cur_max_len / 2, next_max = half + (half & 1);
            bool staged = false;
            if (next_max == 2 && stage_ok()) {   // the fold leaves one pair per row: everything after it runs in one launch
                // this round's gate is in the stream, waiting for this thread: try only (StageSlots)
                int32_t rc = launch_stage(nx_cur.data(), nx_off, already_bound + 1, false);
                if (rc && rc != GM_STAGE_BUSY) return rc;
                staged = rc == GM_OK;
            }
            if (staged) {
                stage_active = false;   // this round still reports through its own kernel; bind() switches over
                stage_armed = true;
                k_enq = 0x7fffffffu;
            } else {
                int32_t rc = launch_sparse_round(nx_cur.data(), nx_off, nx_bound, already_bound + 1);
                if (rc) return rc;
                k_enq = already_bound + 2;
            }
        }
        if (devx) {
            int32_t rc = rs.exchange(sh, 3, stream);
            if (rc) return rc;
        }
        int32_t rc = rs.finish_seq(k_seq[already_bound & 63], 3, stream, acc, !fold_pending);
        if (rc) return rc;
        if (rs.ticket_word()[1]) {
                rs.ticket_word()[1] = 0;   // the staging may be shared with later objects: report once
                return set_err(GM_ERR_STATE, "a pre-enqueued fold timed out waiting for its challenge (gm_set_wait_timeout_ms)");
            }
        if (sh.comm && !devx) {
            rc = shard_sum_fr(sh, acc, 3);
            if (rc) return rc;
        }
        }
        const Fr* w = acc + 2;
        // pads: f(row_pad..) weighted by W, f(col_pad..) by the coefficient tail (vecvec_eq.rs:309-315, 345-371)
        Fr in[GM_MAX_COLS], pr[GM_MAX_COLS], pc[GM_MAX_COLS];
        for (int i = 0; i < k; i++) in[i] = row_pad[i];
        seg_plan_exec_host(sp, in, pr);
        for (int i = 0; i < k; i++) in[i] = col_pad[i];
        seg_plan_exec_host(sp, in, pc);
        Fr padsum = fr_zero(), colsum = fr_zero();
        for (int o = 0; o < sp.n_outs; o++) {
            padsum = fr_add(padsum, o == 0 ? pr[o] : fr_mul(pr[o], gamma_pows[o]));
            colsum = fr_add(colsum, o == 0 ? pc[o] : fr_mul(pc[o], gamma_pows[o]));
        }
        Fr extra = fr_mul(padsum, w[0]);
        if (!sh.comm && nrows < (1u << col_logsize)) extra = fr_add(extra, fr_mul(colsum, row_coef_tail_nrows));
        const Fr total1 = fr_mul(fr_add(acc[0], extra), multiplier);
        const Fr total2 = fr_mul(fr_add(acc[1], extra), multiplier);
        if (inv_eq0.empty()) inv_eq0 = batch_inv_one_minus(point);
        cached = from12_inv(total1, total2, point[binding_var_idx], inv_eq0[binding_var_idx], claim_);
        has_cached = true;
        *coeffs = cached;
        return GM_OK;
    }

    uint64_t cells_bound = 0;  // upper bound of off_cur[nrows]
    uint32_t cur_max_len = 0;  // longest stored row now (halves, re-padded to even, with every sparse bind)

    // ---- persistent stage (k_stage): the thin sparse rounds + bind_into_dense + the whole dense stage in one launch
    std::shared_ptr<StageRun> stage;
    bool stage_active = false, stage_armed = false, stage_denied = false;
    uint32_t stage_r0 = 0;     // already_bound of the launch's first round
    bool stage_decided = false;   // sharded: the ranks have agreed whether this layer's thin rounds run in the stage kernel
    // dense rounds the launch runs on the device: all but the last few, which the host finishes (stage_host_rounds); sharded, the
    // host takes at least the log2(world) rounds that need every rank's elements
    int stage_dev_rounds() const {
        int hr = stage_host_rounds(sp, (int)col_logsize);
        if (sh.comm && hr < (int)sh.lg) hr = (int)sh.lg;
        return (int)col_logsize - hr;
    }
    // what every rank sees alike: the shape fits one launch (the geometry is this rank's slice of the rows, the same on every rank)
    bool stage_shape_ok() const {
        if (!stage_enabled() || RoundScratch::dev_exchange(sh) || k > 16 || col_logsize < 1 || nrows > (1u << col_logsize)) return false;
        const uint32_t nd = sh.comm ? nrows : (1u << col_logsize);
        if (nd < 2 || (nd & (nd - 1)) != 0 || stage_dev_rounds() < 1) return false;
        for (uint32_t i = 0; i < col_logsize; i++)
            if (fr_eq(point[i], fr_one())) return false;   // the dense stage would need the generic object (from12 divides by 1 - q)
        // thin rounds left once the longest row is one pair: the row variables above the first
        const int n_thin_max = (int)n_row_vars0 - (int)already_bound - (cur_max_len > 2 ? 1 : 0);
        return n_thin_max >= 1 && StageRun::fits(sp.nseg, nd, n_thin_max, stage_dev_rounds());
    }
    bool stage_ok() const { return !stage_denied && pinned_exclusive() && stage_shape_ok(); }
    int32_t launch_stage(const Fr* const* cols_now, const uint32_t* off, uint32_t ab0, bool may_wait = true) {
        const int n_thin = (int)(n_row_vars0 - ab0);
        const uint32_t nd = sh.comm ? nrows : (1u << col_logsize);
        if (n_thin < 1 || !StageRun::fits(sp.nseg, nd, n_thin, stage_dev_rounds()))
            return set_err(GM_ERR_STATE, "stage launch: shape does not fit (%d thin rounds, %u dense)", n_thin, col_logsize);
        StageArgs a;
        memset(&a, 0, sizeof(a));
        a.nrows = nrows;
        a.n_elems = nd;
        a.n_thin = n_thin;
        a.n_dense = stage_dev_rounds();   // the host finishes the last few rounds
        a.off = off;
        a.row_coef = d_row_coef.fr() + row_base;
        for (int tr = 0; tr < n_thin; tr++) a.thin_eq[tr] = d_eq_seq.fr() + eq_level_off[eq_level_len.size() - 1 - (ab0 + tr)];
        // the dense stage's eq tables are the lower levels of row_eq_coefs = eq(point[0..col_logsize]), kept in the scratch half; the
        // kernel indexes them by the pair's index inside this rank's slice
        const Fr* levels = d_row_coef.fr() + ((size_t)1 << col_logsize);
        for (int dr = 0; dr < a.n_dense; dr++) a.eq[dr] = levels + ((1ull << (col_logsize - 1 - dr)) - 1) + (row_base >> (dr + 1));
        for (int i = 0; i < k; i++) { a.row_pad.v[i] = row_pad[i]; a.col_pad.v[i] = col_pad[i]; }
        ColPtrs cp;
        for (int i = 0; i < k; i++) cp.p[i] = cols_now[i];
        stage.reset(new StageRun());
        {   // what the host applies to the thin rounds' sums (see k_stage): entry 0 of round tr's eq level = prod_{j < level} (1 - pt[j]) over
            // the row variables (padded_eq_poly_sequence, utils.rs:189-220), and the sum of row_eq_coefs over this object's rows
            const Fr* pt = point.data() + col_logsize;
            stage->thin_e0.resize(n_thin);
            for (int tr = 0; tr < n_thin; tr++) {
                const uint32_t level = (uint32_t)eq_level_len.size() - 1 - (ab0 + (uint32_t)tr);
                Fr e0 = fr_one();
                for (uint32_t j = 0; j < level; j++) e0 = fr_mul(e0, fr_sub(fr_one(), pt[j]));
                stage->thin_e0[tr] = e0;
            }
            stage->c_all = fr_sub(eq_sum_host(point.data(), col_logsize, (uint64_t)row_base + nrows),
                                  eq_sum_host(point.data(), col_logsize, row_base));
        }
        int32_t rc = stage->launch(sp, cp, d_gamma.fr(), a, stream, may_wait);
        if (rc) { stage.reset(); return rc; }
        for (int tr = 0; tr < n_thin; tr++) { prof_small_round(64.0 * k * (double)nrows); prof_fold(96.0 * k * (double)nrows); }
        for (uint32_t dr = 0; (nd >> (dr + 1)) >= 1; dr++) {
            prof_small_round(64.0 * k * (double)(nd >> (dr + 1)));
            prof_fold(96.0 * k * (double)(nd >> (dr + 1)));
        }
        stage_active = true;
        stage_r0 = ab0;
        k_enq = 0x7fffffffu;
        return GM_OK;
    }

    // ---- pre-enqueued small sparse rounds (same scheme as ScDenseDeg2's, see k_fold_gate); rounds are indexed by already_bound
    uint32_t k_enq = 0;
    uint32_t k_seq[64] = {};
    bool fold_pending = false, nx_to_a = false;
    uint32_t fold_ticket = 0;
    const uint32_t* nx_off = nullptr;
    uint64_t nx_bound = 0;
    std::vector<const Fr*> nx_cur;
    int32_t launch_sparse_round(const Fr* const* cols_now, const uint32_t* off, uint64_t cb, uint32_t ab) {
        // eq level: row_eq_poly_seq[len - 1 - already_bound]  (vecvec.rs:129-135) and its prefix sums
        const size_t lvl = eq_level_len.size() - 1 - ab;
        const Fr* eq_row = d_eq_seq.fr() + eq_level_off[lvl];
        const Fr* eq_pre = d_prefix.fr() + eq_level_off[lvl] + lvl;  // level l has len+1 prefix entries
        ColPtrs cp;
        for (int i = 0; i < k; i++) cp.p[i] = cols_now[i];
        // grid from the capacity bound: the exact cell count lives on the device (off[nrows])
        const uint64_t bound_pairs = cb / 2 + 1;
        const bool split = bound_pairs <= SC_SPLIT_MAX_PAIRS;
        const uint64_t gx = bound_pairs > nrows ? bound_pairs : nrows;  // the tail-weight loop runs over rows
        const dim3 grid = round_grid(gx, split ? 2 * sp.nseg : 1);
        const VVArgs va{off, nrows, d_row_coef.fr() + row_base, eq_pre, coarse_for(off)};
        const int lean = (!split && k <= 6) ? lean_prim_of(sp) : 0;
        FinishCtx fc;
        if (RoundScratch::dev_exchange(sh)) {
            int32_t rc = rs.ctx_dev(sh, &fc);
            if (rc) return rc;
        } else {
            fc = rs.ctx();
        }
        if (lean) {
            LeanCols lc;
            for (int i = 0; i < k; i++) lc.p[i] = cols_now[i];
            const int pi = prof_begin(stream, lean * 4 + 1, k, cb / 2, off + nrows, 2 * lean_eval_muls(lean) + 3);
            int32_t rc = launch_deg2_lean<true>(lean, grid, stream, lc, eq_row, d_gamma.fr(), (uint64_t)0, va, fc);
            prof_end(stream, pi);
            if (rc) return rc;
        } else if (split) {
            // single-primitive layers: the lean split kernel (two workgroup rows, one per evaluation point)
            const int lean_s = k <= 6 ? lean_prim_of(sp) : 0;
            LeanCols lc;
            for (int i = 0; i < k && i < 7; i++) lc.p[i] = cols_now[i];
            if (!(lean_s && launch_deg2_lean_split(lean_s, dim3(grid.x, 2), stream, lc, eq_row, d_gamma.fr(), va, fc)))
                hipLaunchKernelGGL((k_round_deg2<true, true>), grid, dim3(SC_THREADS), 0, stream, sp, cp, eq_row, d_gamma.fr(),
                                   (uint64_t)0, va, fc);
        }
        else
            hipLaunchKernelGGL((k_round_deg2<true, false>), grid, dim3(SC_THREADS), 0, stream, sp, cp, eq_row, d_gamma.fr(),
                               (uint64_t)0, va, fc);
        GM_LAUNCH_CHECK();
        k_seq[ab & 63] = fc.seq;
        if (!lean) prof_small_round(64.0 * k * (double)(cb / 2));   // capacity bound of the cells (exact count lives on the device)
        return GM_OK;
    }
    ~ScVecVecDeg2() override {
        if (fold_pending) {  // never leave a waiting gate behind
            rs.publish(already_bound, fr_zero(), fold_ticket);
            (void)hipStreamSynchronize(stream);
        }
    }
    uint32_t n_off_tables = 0;

    int32_t bind(const Fr& t) override {
        if (dense2) return dense2->bind(t);
        if (dense) return dense->bind(t);
        if (!has_cached) return set_err(GM_ERR_STATE, "bind before unipoly (vecvec_eq.rs:299 unwrap)");
        if (stage_active) {   // the fold happens inside the stage kernel
            stage->publish((int)(already_bound - stage_r0), t);
            const Fr mult_next = fr_mul(multiplier, eq_bind_factor(point[binding_var_idx], t));
            const Fr claim_next = evaluate_univar(cached, t);
            has_cached = false;
            if ((uint32_t)binding_var_idx > col_logsize) {   // sparse bind (vecvec_eq.rs:295-300)
                multiplier = mult_next;
                claim_ = claim_next;
                row_logsize--;
                binding_var_idx--;
                already_bound++;
                return GM_OK;
            }
            // bind_into_dense (vecvec_eq.rs:157-190): the launch goes on with the dense stage; the host side is the eq-factored object
            std::unique_ptr<ScDenseDeg2> d(new ScDenseDeg2());
            d->stream = stream;
            d->sp = sp;
            d->num_vars = col_logsize;
            d->sh = sh;                              // sharded: the launch holds this rank's slice; host_collect gathers what it leaves
            d->loc_vars = col_logsize - sh.lg;
            d->glob_off = row_base;
            d->gamma_pows = gamma_pows;
            d->point.assign(point.begin(), point.begin() + col_logsize);
            d->multiplier = mult_next;
            d->claim_ = claim_next;
            d->cols.k = k;
            d->cols.cur.assign(k, nullptr);
            // 1 / (1 - point_j) of the vertical coordinates: already inverted with the row coordinates (a field inversion costs the
            // host ~15 us, and this is the moment every block of the launch is waiting for the next challenge)
            if (inv_eq0.size() >= col_logsize) d->inv_eq0.assign(inv_eq0.begin(), inv_eq0.begin() + col_logsize);
            d->adopt_stage(stage);
            dense2 = std::move(d);
            return GM_OK;
        }
        if ((uint32_t)binding_var_idx > col_logsize) {
            // sparse bind (vecvec_eq.rs:295-300)
            if (fold_pending) {  // the fold is already in the stream: hand it the challenge
                rs.publish(already_bound, t, fold_ticket);
                fold_pending = false;
                for (int i = 0; i < k; i++) cur[i] = nx_cur[i];
                off_cur = nx_off;
                cur_is_a = nx_to_a;
                started = true;
                cells_bound = nx_bound;
                { const uint32_t half = cur_max_len / 2; cur_max_len = half + (half & 1); }
                if (stage_armed) { stage_armed = false; stage_active = true; }   // the stage kernel is behind this fold in the stream
                row_logsize--;
                multiplier = fr_mul(multiplier, eq_bind_factor(point[binding_var_idx], t));
                binding_var_idx--;
                already_bound++;
                claim_ = evaluate_univar(cached, t);
                has_cached = false;
                return GM_OK;
            }
            const bool to_a = !started || !cur_is_a;
            if (already_bound + 1 >= n_off_tables) return set_err(GM_ERR_STATE, "more sparse binds than row variables");
            const uint32_t* off_next = off_tab + (uint64_t)(already_bound + 1) * (nrows + 1);
            const uint64_t new_bound = cells_bound / 2 + nrows;
            ColPtrs ci;
            ColPtrsMut co;
            PadCols pd;
            if (k > 16) return set_err(GM_ERR_INVALID, "VecVec sumcheck supports at most 16 polynomials");
            for (int i = 0; i < k; i++) {
                ci.p[i] = cur[i];
                co.p[i] = to_a ? bufA[i]->fr() : bufB[i]->fr();
                pd.v[i] = row_pad[i];
            }
            hipLaunchKernelGGL(k_vv_fold, dim3(ceil_div(new_bound, SC_THREADS), (k + 1) / 2), dim3(SC_THREADS), 0, stream, ci, co,
                               off_cur, off_next, nrows, t, pd, (const Fr*)nullptr, k, coarse_for(off_next), GateArgs{nullptr, 0u, nullptr, 0ull});
            GM_LAUNCH_CHECK();
            prof_fold(96.0 * k * (double)(cells_bound / 2));
            for (int i = 0; i < k; i++) cur[i] = co.p[i];
            off_cur = off_next;
            cur_is_a = to_a;
            started = true;
            cells_bound = new_bound;
            { const uint32_t half = cur_max_len / 2; cur_max_len = half + (half & 1); }
            row_logsize--;
            multiplier = fr_mul(multiplier, eq_bind_factor(point[binding_var_idx], t));
            binding_var_idx--;
            already_bound++;
            claim_ = evaluate_univar(cached, t);
            has_cached = false;
            return GM_OK;
        }
        return bind_into_dense(t);
    }

    // vecvec_eq.rs:157-190
    // The reference hands over to a generic DenseSumcheckObjectSO whose function is eq * GammaWrapper(f) with the eq table as
    // one more column (vecvec_eq.rs:177-189).  The round polynomials of that object are those of the eq-factored one
    // (DenseDeg2: two evaluations of f per pair instead of three, no eq column to fold), so the latter is used; its running
    // multiplier is the eq column's final evaluation.  from12 divides by 1 - point_j: with a coordinate equal to 1 the
    // generic object is kept.
    int32_t bind_into_dense(const Fr& t) {
        bool coord_is_one = col_logsize == 0;
        for (uint32_t i = 0; i < col_logsize; i++) coord_is_one = coord_is_one || fr_eq(point[i], fr_one());
        if (!coord_is_one) return bind_into_dense_deg2(t);
        std::unique_ptr<ScDense> d(new ScDense());
        d->stream = stream;
        d->kind = 0;
        d->sp = sp;
        d->D = 3;
        d->num_vars = col_logsize;
        d->sh = sh;
        d->loc_vars = col_logsize - sh.lg;
        const uint32_t nd_glob = 1u << col_logsize;
        const uint32_t nd = sh.comm ? nrows : nd_glob;  // sharded: the dense columns are this rank's slice of the rows
        std::vector<const Fr*> cptr;
        ColPtrs ci;
        ColPtrsMut co;
        PadCols rp, cpad;
        for (int i = 0; i < k; i++) {
            d->owned.emplace_back(new DevBuf());
            int32_t rc = d->owned.back()->alloc((size_t)nd * sizeof(Fr));
            if (rc) return rc;
            ci.p[i] = cur[i];
            co.p[i] = d->owned.back()->fr();
            rp.v[i] = row_pad[i];
            cpad.v[i] = col_pad[i];
            cptr.push_back(d->owned.back()->fr());
        }
        hipLaunchKernelGGL(k_vv_fold_to_dense, dim3(ceil_div(nd, SC_THREADS), k), dim3(SC_THREADS), 0, stream, ci, co,
                           off_cur, nrows, nd, t, rp, cpad);
        GM_LAUNCH_CHECK();
        prof_fold(96.0 * k * (double)nd);
        // eq over the vertical variables, scaled by the multiplier after this bind (vecvec_eq.rs:177-180)
        const Fr mult = fr_mul(multiplier, eq_bind_factor(point[binding_var_idx], t));
        d->owned.emplace_back(new DevBuf());
        int32_t rc = d->owned.back()->alloc((size_t)2 * nd_glob * sizeof(Fr));
        if (rc) return rc;
        Fr* base = d->owned.back()->fr();
        std::vector<Fr*> lv(col_logsize + 1);
        for (uint32_t i = 0; i <= col_logsize; i++) lv[i] = base + ((1ull << i) - 1);
        rc = launch_eq_sequence(mult, point.data(), col_logsize, lv.data(), stream);
        if (rc) return rc;
        cptr.push_back(lv[col_logsize] + row_base);
        rc = d->cols.init(k + 1, cptr.data(), nd);
        if (rc) return rc;
        // GammaWrapper::new(func, gamma_pows[1])  (vecvec_eq.rs:185-187): same powers gamma^o
        rc = upload_gamma(gamma_pows, &d->d_gamma, stream);
        if (rc) return rc;
        rc = d->rs.init(stream);
        if (rc) return rc;
        d->claim_ = evaluate_univar(cached, t);
        has_cached = false;
        dense = std::move(d);
        return GM_OK;
    }

    int32_t bind_into_dense_deg2(const Fr& t);

    int32_t final_evals(std::vector<Fr>* out) override {
        if (dense2) {
            int32_t rc = dense2->final_evals(out);
            if (rc) return rc;
            out->push_back(dense2_eq_final());  // the eq column of the reference's dense stage (vecvec_eq.rs:451 drops it again)
            return GM_OK;
        }
        if (!dense) return set_err(GM_ERR_STATE, "final_evals in the sparse stage (vecvec_eq.rs:391 unreachable!)");
        return dense->final_evals(out);
    }
};

Fr ScVecVecDeg2::dense2_eq_final() const { return static_cast<const ScDenseDeg2*>(dense2.get())->multiplier; }

int32_t ScVecVecDeg2::bind_into_dense_deg2(const Fr& t) {
    std::unique_ptr<ScDenseDeg2> d(new ScDenseDeg2());
    d->stream = stream;
    d->sp = sp;
    d->num_vars = col_logsize;
    d->sh = sh;
    d->loc_vars = col_logsize - sh.lg;
    d->glob_off = row_base;
    d->gamma_pows = gamma_pows;
    d->point.assign(point.begin(), point.begin() + col_logsize);
    d->multiplier = fr_mul(multiplier, eq_bind_factor(point[binding_var_idx], t));  // vecvec_eq.rs:177-180
    d->claim_ = evaluate_univar(cached, t);
    if (inv_eq0.size() >= col_logsize) d->inv_eq0.assign(inv_eq0.begin(), inv_eq0.begin() + col_logsize);
    const uint32_t nd = sh.comm ? nrows : (1u << col_logsize);  // sharded: this rank's slice of the rows
    std::vector<const Fr*> cptr;
    ColPtrs ci;
    ColPtrsMut co;
    PadCols rp, cpad;
    for (int i = 0; i < k; i++) {
        d->owned.emplace_back(new DevBuf());
        int32_t rc = d->owned.back()->alloc((size_t)nd * sizeof(Fr));
        if (rc) return rc;
        ci.p[i] = cur[i];
        co.p[i] = d->owned.back()->fr();
        rp.v[i] = row_pad[i];
        cpad.v[i] = col_pad[i];
        cptr.push_back(d->owned.back()->fr());
    }
    hipLaunchKernelGGL(k_vv_fold_to_dense, dim3(ceil_div(nd, SC_THREADS), k), dim3(SC_THREADS), 0, stream, ci, co, off_cur, nrows,
                       nd, t, rp, cpad);
    GM_LAUNCH_CHECK();
    prof_fold(96.0 * k * (double)nd);
    int32_t rc = d->cols.init(k, cptr.data(), nd);
    if (rc) return rc;
    rc = upload_gamma(gamma_pows, &d->d_gamma, stream);
    if (rc) return rc;
    // eq_poly_sequence(point[0..col_logsize - 1]) = the lower levels of row_eq_coefs, already in the scratch half of d_row_coef
    d->eq_ext = d_row_coef.fr() + ((size_t)1 << col_logsize);
    rc = d->rs.init(stream);
    if (rc) return rc;
    has_cached = false;
    dense2 = std::move(d);
    return GM_OK;
}

}  // namespace

// ============================================================================================ C ABI
static int32_t parse_fn(const gm_fn* f, GmFn* g, SegPlan* sp) {
    int32_t rc = to_gmfn(f, g);
    if (rc) return rc;
    if (!seg_plan_build(*g, sp)) return set_err(GM_ERR_INVALID, "function too wide");
    return GM_OK;
}

static Fr rlc_claims(const std::vector<Fr>& gp, const uint64_t* h_claims, int n) {
    // claim = claims[0] + sum_{i>=1} gamma^i claims[i]   (dense_eq.rs:45-49, vecvec_eq.rs:57-61)
    Fr c;
    memcpy(&c, h_claims, 32);
    for (int i = 1; i < n; i++) {
        Fr ci;
        memcpy(&ci, h_claims + 4 * i, 32);
        c = fr_add(c, fr_mul(gp[i], ci));
    }
    return c;
}

extern "C" int32_t gm_sc_dense_deg2_create(const gm_fn* f, uint32_t num_vars, const uint64_t* const* d_cols,
                                           const uint64_t* h_point, const uint64_t* h_gamma, const uint64_t* h_claims,
                                           gm_sc** out, void* stream) {
    GM_REQUIRE(out && d_cols && h_point && h_gamma && h_claims && num_vars >= 1 && num_vars <= 30, "bad argument");
    std::unique_ptr<ScDenseDeg2> so(new ScDenseDeg2());
    GmFn g;
    int32_t rc = parse_fn(f, &g, &so->sp);
    if (rc) return rc;
    GM_REQUIRE(so->sp.deg == 2, "DenseDeg2Sumcheck needs a degree-2 function (dense_eq.rs:200)");
    so->stream = as_stream(stream);
    so->num_vars = num_vars;
    so->sh = current_shard();
    GM_REQUIRE(so->sh.lg <= num_vars, "more ranks than elements");
    so->loc_vars = num_vars - so->sh.lg;
    so->glob_off = (uint64_t)so->sh.rank << so->loc_vars;
    Fr gamma;
    memcpy(&gamma, h_gamma, 32);
    so->gamma_pows = make_gamma_pows(gamma, so->sp.n_outs);
    so->claim_ = rlc_claims(so->gamma_pows, h_claims, so->sp.n_outs);
    so->point.resize(num_vars);
    memcpy(so->point.data(), h_point, 32 * (size_t)num_vars);
    so->multiplier = fr_one();
    rc = so->cols.init(so->sp.n_ins, reinterpret_cast<const Fr* const*>(d_cols), 1ull << so->loc_vars);
    if (rc) return rc;
    rc = alloc_gamma(so->gamma_pows, &so->d_gamma);
    if (rc) return rc;
    // eq_poly_sequence(point[0 .. n-1])  (dense_eq.rs:85): levels 0..n-1, level i has 2^i entries
    if (so->sh.comm && so->sh.lg >= 1 && so->loc_vars >= 1) {
        // sharded: this rank's slice of the levels lg .. n-1 (local levels 0 .. loc_vars-1, scaled by eq(point[0..lg), rank)) and the
        // whole levels 0 .. lg-1 behind them
        const uint32_t lg = so->sh.lg, lv_n = so->loc_vars;
        const uint32_t top_n = (lg + shard_gather_log() < num_vars) ? lg + shard_gather_log() : num_vars;   // whole levels 0 .. top_n - 1
        so->eq_sliced = true; so->eq_lg = lg; so->eq_rank = so->sh.rank; so->eq_top = top_n;
        rc = so->d_eq.alloc((((size_t)1 << lv_n) + ((size_t)1 << top_n)) * sizeof(Fr));
        if (rc) return rc;
        Fr factor = fr_one();   // eq(point[0..lg), rank): point[0] is the most significant variable
        for (uint32_t j = 0; j < lg; j++) {
            const Fr& q = so->point[j];
            factor = fr_mul(factor, ((so->sh.rank >> (lg - 1 - j)) & 1u) ? q : fr_sub(fr_one(), q));
        }
        std::vector<Fr*> lv(lv_n);
        for (uint32_t i = 0; i < lv_n; i++) lv[i] = so->d_eq.fr() + ((1ull << i) - 1);
        rc = launch_eq_sequence(factor, so->point.data() + lg, lv_n - 1, lv.data(), so->stream, so->gamma_pows.data(),
                                (uint32_t)so->gamma_pows.size(), so->d_gamma.fr());
        if (rc) return rc;
        std::vector<Fr*> top(top_n);
        for (uint32_t i = 0; i < top_n; i++) top[i] = so->d_eq.fr() + ((size_t)1 << lv_n) + ((1ull << i) - 1);
        rc = launch_eq_sequence(fr_one(), so->point.data(), top_n - 1, top.data(), so->stream);
        if (rc) return rc;
    } else {
    rc = so->d_eq.alloc(((size_t)1 << num_vars) * sizeof(Fr));
    if (rc) return rc;
    std::vector<Fr*> lv(num_vars);
    for (uint32_t i = 0; i < num_vars; i++) lv[i] = so->d_eq.fr() + ((1ull << i) - 1);
    rc = launch_eq_sequence(fr_one(), so->point.data(), num_vars - 1, lv.data(), so->stream, so->gamma_pows.data(),
                            (uint32_t)so->gamma_pows.size(), so->d_gamma.fr());
    if (rc) return rc;
    }
    rc = so->rs.init(so->stream);
    if (rc) return rc;
    so->inv_eq0 = batch_inv_one_minus(so->point);   // now, while the device builds the eq tables: not when the first round's sums are in
    *out = so.release();
    return GM_OK;
}

extern "C" int32_t gm_sc_vecvec_deg2_create(const gm_fn* f, const gm_vv* polys, const uint64_t* h_point,
                                            const uint64_t* h_gamma, const uint64_t* h_claims, gm_sc** out,
                                            void* stream) {
    GM_REQUIRE(out && polys && h_point && h_gamma && h_claims, "bad argument");
    std::unique_ptr<ScVecVecDeg2> so(new ScVecVecDeg2());
    int32_t rc = parse_fn(f, &so->fn, &so->sp);
    if (rc) return rc;
    GM_REQUIRE(so->sp.deg == 2, "VecVecDeg2Sumcheck needs a degree-2 function (vecvec_eq.rs:426)");
    GM_REQUIRE((int)polys->k == so->sp.n_ins, "%u polys for a %d-input function", polys->k, so->sp.n_ins);
    GM_REQUIRE(polys->k <= 16, "VecVec sumcheck supports at most 16 polynomials");
    GM_REQUIRE(polys->row_logsize >= 1, "row_logsize must be >= 1");
    GM_REQUIRE(polys->max_row_len >= 2, "all rows empty (log_2(0), vecvec.rs:86)");
    hipStream_t s = as_stream(stream);
    so->stream = s;
    so->k = polys->k;
    so->nrows = polys->nrows;
    so->col_logsize = polys->col_logsize;
    so->row_logsize = polys->row_logsize;
    so->n_row_vars0 = polys->row_logsize;
    so->row_pad = polys->row_pad;
    so->col_pad = polys->col_pad;
    so->cells_bound = polys->total;
    so->cur_max_len = polys->max_row_len + (polys->max_row_len & 1);
    so->sh = current_shard();
    so->row_base = polys->row_base;
    if (so->sh.comm) {
        GM_REQUIRE(((uint64_t)polys->nrows << so->sh.lg) == (1ull << polys->col_logsize) &&
                       polys->row_base == so->sh.rank * polys->nrows,
                   "sharded VecVec sumcheck: every rank must hold 2^col_logsize / world rows, in rank order");
    } else {
        GM_REQUIRE(polys->row_base == 0, "a partial VecVec needs a sharding context");
    }
    const uint32_t nvars = polys->row_logsize + polys->col_logsize;
    Fr gamma;
    memcpy(&gamma, h_gamma, 32);
    so->gamma_pows = make_gamma_pows(gamma, so->sp.n_outs);
    so->claim_ = rlc_claims(so->gamma_pows, h_claims, so->sp.n_outs);
    so->point.resize(nvars);
    memcpy(so->point.data(), h_point, 32 * (size_t)nvars);
    so->multiplier = fr_one();
    so->binding_var_idx = (int)nvars - 1;
    for (uint32_t i = 0; i < polys->k; i++) so->cur.push_back(polys->cols[i]->fr());
    so->off_cur = reinterpret_cast<const uint32_t*>(polys->off->p);
    so->n_off_tables = polys->row_logsize + 1;
    if (polys->off_levels && polys->off_level + polys->row_logsize <= polys->n_off_levels) {
        // the layouts of the row_logsize sparse rounds (tables 0 .. row_logsize - 1) are in the shape's table already
        so->off_tab = reinterpret_cast<const uint32_t*>(polys->off_levels->p) + (size_t)polys->off_level * (so->nrows + 1);
        so->off_keep = polys->off_levels;
        so->n_off_tables = polys->row_logsize;
        so->coarse_keep = polys->coarse;
        so->coarse_off = polys->coarse_off;
        so->coarse_l0 = polys->off_level;
    } else {
        rc = so->off_all.alloc((size_t)so->n_off_tables * (so->nrows + 1) * 4);
        if (rc) return rc;
        rc = launch_offsets_all_from_off(so->off_cur, reinterpret_cast<uint32_t*>(so->off_all.p), so->nrows, so->n_off_tables, s);
        if (rc) return rc;
        so->off_tab = reinterpret_cast<const uint32_t*>(so->off_all.p);
    }
    for (uint32_t i = 0; i < polys->k; i++) {
        so->bufA.emplace_back(new DevBuf());
        so->bufB.emplace_back(new DevBuf());
        rc = so->bufA.back()->alloc((size_t)(polys->total / 2 + so->nrows + 2) * sizeof(Fr));
        if (rc) return rc;
        rc = so->bufB.back()->alloc((size_t)(polys->total / 4 + 2 * so->nrows + 2) * sizeof(Fr));
        if (rc) return rc;
    }
    rc = alloc_gamma(so->gamma_pows, &so->d_gamma);   // stored by the eq-table launch below
    if (rc) return rc;
    // EQPolyData::new (vecvec.rs:85-119)
    uint32_t max_seg_log = 0;
    {   // liblasso log_2: exact for powers of two, else bit length
        uint32_t m = polys->max_row_len, bl = 0;
        while ((1u << bl) < m) bl++;
        max_seg_log = bl;
    }
    GM_REQUIRE(max_seg_log <= polys->row_logsize, "row longer than 2^row_logsize");
    const uint32_t segment_vars_idx = nvars - max_seg_log;
    const uint32_t padded = segment_vars_idx - polys->col_logsize;       // padded_vars_range().len()
    const uint32_t n_seq_vars = (nvars - 1) - polys->col_logsize;        // row_vars_range().len()
    so->padded_vars = padded;
    // row_eq_coefs = eq(point[0..col_logsize]) and its tail sums
    std::vector<Fr*> row_lv;
    {
        rc = so->d_row_coef.alloc(((size_t)2 << polys->col_logsize) * sizeof(Fr));
        if (rc) return rc;
        std::vector<Fr*> lv(polys->col_logsize + 1);
        Fr* scratch = so->d_row_coef.fr() + ((size_t)1 << polys->col_logsize);
        for (uint32_t i = 0; i < polys->col_logsize; i++) lv[i] = scratch + ((1ull << i) - 1);
        lv[polys->col_logsize] = so->d_row_coef.fr();
        row_lv = lv;   // launched below, together with the padded row sequence
        // row_eq_coefs_tail_sums[nrows] = sum_{j >= nrows} eq(point[0..col], j) = 1 - eq_sum(point[0..col], nrows)
        so->row_coef_tail_nrows = fr_sub(fr_one(), eq_sum_host(so->point.data(), polys->col_logsize, so->nrows));   // (a 2^col-entry host vector was zero-filled here for this one entry: 262 KB per layer at config B)
    }
    // padded_eq_poly_sequence(padded, point[row vars])  (utils.rs:189-220): levels 0..n_seq_vars
    {
        so->eq_level_off.resize(n_seq_vars + 1);
        so->eq_level_len.resize(n_seq_vars + 1);
        uint64_t tot = 0;
        for (uint32_t i = 0; i <= n_seq_vars; i++) {
            so->eq_level_len[i] = (i <= padded) ? 1u : (1u << (i - padded));
            so->eq_level_off[i] = tot;
            tot += so->eq_level_len[i];
        }
        rc = so->d_eq_seq.alloc((size_t)tot * sizeof(Fr));
        if (rc) return rc;
        // the first `padded` variables only scale: level i (i <= padded) = prod_{j<i} (1 - pt[j])
        const Fr* pt = so->point.data() + polys->col_logsize;
        Fr m = fr_one();
        for (uint32_t i = 1; i <= padded; i++) m = fr_mul(m, fr_sub(fr_one(), pt[i - 1]));
        // upload scalar levels 0..padded
        Fr acc = fr_one();
        std::vector<Fr> scal(padded + 1);
        scal[0] = fr_one();
        for (uint32_t i = 1; i <= padded; i++) { acc = fr_mul(acc, fr_sub(fr_one(), pt[i - 1])); scal[i] = acc; }
        std::vector<Fr*> lv(n_seq_vars - padded + 1);
        for (uint32_t i = padded; i <= n_seq_vars; i++) lv[i - padded] = so->d_eq_seq.fr() + so->eq_level_off[i];
        // levels padded..n_seq_vars are the ordinary doubling levels started from the scalar m; both sequences and the scalar levels
        // in one launch when they fit (they do for every shape with col_logsize, row variables <= 14)
        if (!launch_eq_pair(fr_one(), so->point.data(), polys->col_logsize, row_lv.data(), m, pt + padded, n_seq_vars - padded, lv.data(),
                            scal.data(), (uint32_t)scal.size(), so->d_eq_seq.fr(), s, so->gamma_pows.data(), (uint32_t)so->gamma_pows.size(),
                            so->d_gamma.fr())) {
            rc = upload_small(so->gamma_pows.data(), so->gamma_pows.size(), so->d_gamma.fr(), s);
            if (rc) return rc;
            rc = launch_eq_sequence(fr_one(), so->point.data(), polys->col_logsize, row_lv.data(), s);
            if (rc) return rc;
            rc = upload_small(scal.data(), scal.size(), so->d_eq_seq.fr(), s);
            if (rc) return rc;
            rc = launch_eq_sequence(m, pt + padded, n_seq_vars - padded, lv.data(), s);
            if (rc) return rc;
        }
        // row_eq_poly_prefix_seq (vecvec.rs:101-109): level l -> len_l + 1 prefix sums, packed at offset off_l + l
        rc = so->d_prefix.alloc((size_t)(tot + n_seq_vars + 2) * sizeof(Fr));
        if (rc) return rc;
        const uint32_t top_len = 1u << (n_seq_vars - padded);
        hipLaunchKernelGGL(k_prefix_sums_levels, dim3(ceil_div((uint64_t)top_len + 1, SC_THREADS), n_seq_vars + 1), dim3(SC_THREADS),
                           0, s, so->d_eq_seq.fr(), so->d_prefix.fr(), padded, n_seq_vars);
        GM_LAUNCH_CHECK();
    }
    rc = so->rs.init(so->stream);
    if (rc) return rc;
    so->inv_eq0 = batch_inv_one_minus(so->point);   // see gm_sc_dense_deg2_create
    *out = so.release();
    return GM_OK;
}

// DenseSumcheckObjectSO with F = EqWrapper(GammaWrapper(f, gamma)) (kind 0: d_cols = f.n_ins columns + the eq column)
// or Prod3Fn (kind 1: 3 columns, f ignored); claim_hint as in sumcheck.rs:250.
extern "C" int32_t gm_sc_dense_create(int32_t kind, const gm_fn* f, uint32_t num_vars, const uint64_t* const* d_cols,
                                      const uint64_t* h_gamma, const uint64_t* h_claim, gm_sc** out, void* stream) {
    GM_REQUIRE(out && d_cols && h_claim && num_vars >= 1 && num_vars <= 30 && kind >= 0 && kind <= 2, "bad argument");
    std::unique_ptr<ScDense> so(new ScDense());
    so->stream = as_stream(stream);
    so->kind = kind;
    so->num_vars = num_vars;
    int ncols = 3;
    std::vector<Fr> gp = {fr_one()};
    if (kind == 0) {
        GM_REQUIRE(h_gamma, "gamma required");
        GmFn g;
        int32_t rc = parse_fn(f, &g, &so->sp);
        if (rc) return rc;
        GM_REQUIRE(so->sp.n_outs > 1, "GammaWrapper needs n_outs > 1 (sumcheck.rs:714)");
        so->D = so->sp.deg + 1;
        ncols = so->sp.n_ins + 1;
        Fr gamma;
        memcpy(&gamma, h_gamma, 32);
        gp = make_gamma_pows(gamma, so->sp.n_outs);
    } else if (kind == 2) {
        // FoldedProdAlgFn(gamma, nargs) (multiopen_reduction.rs:13-42): f = IdAlgFn(nargs) only carries nargs
        GM_REQUIRE(h_gamma && f && f->nseg == 1 && f->count[0] >= 1 && f->count[0] <= 8, "kind 2 needs gamma and f = {GM_FN_ID x nargs}, nargs <= 8");
        const int nargs = f->count[0];
        so->D = 2;
        ncols = 2 * nargs;
        Fr gamma;
        memcpy(&gamma, h_gamma, 32);
        gp = make_gamma_pows(gamma, nargs > 1 ? nargs : 2);
    } else {
        so->D = 3;
    }
    GM_REQUIRE(so->D == 2 || so->D == 3, "unsupported degree %d", so->D);
    memcpy(&so->claim_, h_claim, 32);
    so->sh = current_shard();
    GM_REQUIRE(so->sh.lg <= num_vars, "more ranks than elements");
    so->loc_vars = num_vars - so->sh.lg;
    int32_t rc = so->cols.init(ncols, reinterpret_cast<const Fr* const*>(d_cols), 1ull << so->loc_vars);
    if (rc) return rc;
    rc = upload_gamma(gp, &so->d_gamma, so->stream);
    if (rc) return rc;
    rc = so->rs.init(so->stream);
    if (rc) return rc;
    *out = so.release();
    return GM_OK;
}

extern "C" int32_t gm_sc_unipoly(gm_sc* so, uint64_t* h_coeffs, uint32_t* n_coeffs) {
    GM_REQUIRE(so && h_coeffs, "null argument");
    std::vector<Fr> c;
    int32_t rc = so->unipoly(&c);
    if (rc) return rc;
    memcpy(h_coeffs, c.data(), c.size() * sizeof(Fr));
    if (n_coeffs) *n_coeffs = (uint32_t)c.size();
    return GM_OK;
}

// diagnostics: blocks of the persistent round kernel the calling process has in flight on the current device, and the budget
extern "C" int32_t gm_stage_slots(uint32_t* in_flight, uint32_t* capacity) {
    const int dev = StageSlots::device();
    const uint32_t c = StageSlots::get().cap(dev);
    std::lock_guard<std::mutex> g(StageSlots::get().mu);
    if (in_flight) *in_flight = StageSlots::get().in_flight[dev];
    if (capacity) *capacity = c;
    return GM_OK;
}

extern "C" int32_t gm_sc_bind(gm_sc* so, const uint64_t* h_t) {
    GM_REQUIRE(so && h_t, "null argument");
    Fr t;
    memcpy(&t, h_t, 32);
    return so->bind(t);
}

extern "C" int32_t gm_sc_final_evals(gm_sc* so, uint64_t* h_evals, uint32_t* n_evals) {
    GM_REQUIRE(so && h_evals, "null argument");
    std::vector<Fr> e;
    int32_t rc = so->final_evals(&e);
    if (rc) return rc;
    memcpy(h_evals, e.data(), e.size() * sizeof(Fr));
    if (n_evals) *n_evals = (uint32_t)e.size();
    return GM_OK;
}

extern "C" int32_t gm_sc_claim(const gm_sc* so, uint64_t* h_claim) {
    GM_REQUIRE(so && h_claim, "null argument");
    Fr c = so->claim();
    memcpy(h_claim, &c, 32);
    return GM_OK;
}

// ---- profiler ABI (see ScProf)
// k_stage launches made by this process so far, and how many of them were left before their first round (the grid did not become
// resident, or -- sharded -- another rank's did not)
extern "C" int32_t gm_sc_stage_counts(uint64_t* launched, uint64_t* left_early) {
    if (launched) *launched = g_stage_launched.load();
    if (left_early) *left_early = g_stage_left.load();
    return GM_OK;
}

extern "C" int32_t gm_sc_profile(int32_t mode) {
    GM_REQUIRE(mode >= 0 && mode <= 2, "mode 0 (off), 1 (time the large round kernels) or 2 (+ account every round and fold)");
    ScProf& p = sc_prof();
    for (auto& r : p.recs) { p.free_events.push_back(r.e0); p.free_events.push_back(r.e1); }
    p.recs.clear();
    p.n_slots = 0;
    p.small_round_bytes = p.fold_bytes = p.small_rounds = p.folds = 0;
    p.mode = mode;
    return GM_OK;
}

extern "C" int32_t gm_sc_profile_read(gm_sc_profile_row* rows, uint32_t cap, uint32_t* n_rows, double* other_round_bytes,
                                      double* fold_bytes, void* stream) {
    GM_REQUIRE(n_rows, "null argument");
    ScProf& p = sc_prof();
    GM_HIP(hipStreamSynchronize(as_stream(stream)));
    std::vector<gm_sc_profile_row> acc;
    std::vector<int> cls_of;
    for (auto& r : p.recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
        const uint64_t pairs = r.h_cells ? (uint64_t)(*r.h_cells >> 1) : r.pairs;
        size_t j = 0;
        for (; j < cls_of.size(); j++) if (cls_of[j] == r.cls) break;
        if (j == cls_of.size()) {
            cls_of.push_back(r.cls);
            gm_sc_profile_row row;
            memset(&row, 0, sizeof(row));
            snprintf(row.kernel, sizeof(row.kernel), "%s", sc_class_name(r.cls));
            row.k_cols = (uint32_t)r.k;
            acc.push_back(row);
        }
        gm_sc_profile_row& row = acc[j];
        row.launches++;
        row.total_ms += ms;
        row.pairs += (double)pairs;
        row.alg_bytes += (64.0 * r.k + ((r.cls & 3) == 2 ? 0.0 : ((r.cls & 3) == 3 ? 0.0 : 32.0))) * (double)pairs;
        row.fr_mul += (double)r.fr_mul_per_pair * (double)pairs;
        if (ms > row.max_ms) { row.max_ms = ms; row.max_ms_pairs = (double)pairs; }
    }
    *n_rows = (uint32_t)acc.size();
    if (rows) {
        GM_REQUIRE(acc.size() <= cap, "row buffer too small (%zu classes)", acc.size());
        for (size_t j = 0; j < acc.size(); j++) rows[j] = acc[j];
    }
    if (other_round_bytes) *other_round_bytes = p.small_round_bytes;
    if (fold_bytes) *fold_bytes = p.fold_bytes;
    for (auto& r : p.recs) { p.free_events.push_back(r.e0); p.free_events.push_back(r.e1); }
    p.recs.clear();
    p.n_slots = 0;
    p.small_round_bytes = p.fold_bytes = p.small_rounds = p.folds = 0;
    return GM_OK;
}

extern "C" int32_t gm_set_wait_timeout_ms(uint32_t ms) {
    wait_timeout_ms().store(ms ? ms : 20000u);
    return GM_OK;
}

extern "C" int32_t gm_sc_destroy(gm_sc* so) {
    delete so;
    return GM_OK;
}
