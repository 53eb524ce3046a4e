// Knuckles opening: the last step of Pippenger::prove's "open" span, on the device.
//   KnucklesProvingKey::new (the `inverses` table), compute_t        /root/reference/src/commitments/knuckles.rs:64-82, 111-154
//   KzgProvingKey::{commit, open}, div_by_linear, ev                 /root/reference/src/commitments/kzg.rs:73-81, 123-132, 142-150
//   KzgVerifyingKey::verify_reduce_to_pair                           /root/reference/src/commitments/kzg.rs:46-59
//   KnucklesOpeningProtocol::prove                                   /root/reference/src/cleanup/protocols/opening.rs:39-98
// Work: num_vars streaming passes over <= 2N - 1 field elements (compute_t), three G1 MSMs of 2N - 1 / 2N - 2 points
// (gm_g1_msm, the dominant cost), two polynomial evaluations and two divisions by a linear factor.  The reference's
// div_by_linear / ev are serial Horner loops; here a polynomial is cut into 64-coefficient chunks, every chunk is reduced
// with Horner by one thread, the <= 2^15 chunk values are chained on the host, and a second pass writes the quotient.
// All field results are the same canonical elements as the serial loops produce (exact arithmetic); G1 results are the
// same group elements.
#include <vector>

#include "g1.hip.h"
#include "internal.hpp"

using namespace gm;

extern "C" {
int32_t gm_g1_msm(const uint64_t* d_bases_aff, const uint64_t* d_scalars, uint64_t n, int32_t scalars_mont, uint32_t nbits,
                  uint64_t* h_out_aff, void* stream);
int32_t gm_g1_combine_parts(const gm_comm* comm, const uint64_t* h_parts_jac, uint32_t n, uint64_t* h_out_aff);
}

namespace gm {

static constexpr uint32_t KN_CHUNK = 64;

__device__ __forceinline__ Fr fr_pow_u64(Fr b, uint64_t e) {
    Fr acc = fr_one();
    while (e) {
        if (e & 1) acc = fr_mul(acc, b);
        b = fr_sqr(b);
        e >>= 1;
    }
    return acc;
}

// inverses[s] = 1 / (k^s - k^(N-1)), with 1 at s = N - 1 (knuckles.rs:64-82)
__global__ void __launch_bounds__(128) k_kn_inverses(Fr k, Fr k_n1, uint64_t n, uint64_t total, Fr* __restrict__ out) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total) return;
    Fr v = fr_sub(fr_pow_u64(k, s), k_n1);
    if (s == n - 1) v = fr_add(v, fr_one());
    fr_store(out + s, fr_inv(v));
}

// one pass of compute_t (knuckles.rs:131-146): with s(j) = t[j] * c for j < curr (0 beyond),
//   out[idx] = t[idx] - s(idx) + (idx >= offset ? s(idx - offset) : 0)   for idx < curr + offset
__global__ void __launch_bounds__(256) k_kn_pass(const Fr* __restrict__ t, Fr c, uint64_t curr, uint64_t offset, uint64_t total,
                                                  Fr* __restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    Fr v = idx < curr + offset ? fr_load(t + idx) : fr_zero();
    if (idx < curr + offset) {
        if (idx < curr) v = fr_sub(v, fr_mul(v, c));
        if (idx >= offset && idx - offset < curr) v = fr_add(v, fr_mul(fr_load(t + idx - offset), c));
    }
    fr_store(out + idx, v);
}

__global__ void __launch_bounds__(256) k_kn_scale(Fr* __restrict__ t, const Fr* __restrict__ inv, uint64_t zero_at, uint64_t total) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    fr_store(t + idx, idx == zero_at ? fr_zero() : fr_mul(fr_load(t + idx), fr_load(inv + idx)));
}

// out[i] = lambda * t[i] + (i < plen ? poly[i] : 0)   (opening.rs:67-77)
__global__ void __launch_bounds__(256) k_kn_plt(const Fr* __restrict__ t, const Fr* __restrict__ poly, uint64_t plen, Fr lambda,
                                                 uint64_t total, Fr* __restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    Fr v = fr_mul(lambda, fr_load(t + idx));
    if (idx < plen) v = fr_add(v, fr_load(poly + idx));
    fr_store(out + idx, v);
}

// H[c] = sum_{i < 64} p[64 c + i] x^i
__global__ void __launch_bounds__(128) k_kn_chunk_eval(const Fr* __restrict__ p, uint64_t len, Fr x, uint64_t nchunks, Fr* __restrict__ H) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t lo = c * KN_CHUNK, hi = (lo + KN_CHUNK < len) ? lo + KN_CHUNK : len;
    Fr acc = fr_zero();
    for (uint64_t i = hi; i-- > lo;) acc = fr_add(fr_mul(acc, x), fr_load(p + i));
    fr_store(H + c, acc);
}

// quotient of p by (X - x) inside chunk c given the remainder entering it from above (div_by_linear, kzg.rs:73-81):
//   rem = R[c]; for i = hi-1 .. lo: q[i - 1] = rem (i >= 1); rem = p[i] + rem * x
__global__ void __launch_bounds__(128) k_kn_chunk_div(const Fr* __restrict__ p, uint64_t len, Fr x, uint64_t nchunks,
                                                       const Fr* __restrict__ R, Fr* __restrict__ q) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t lo = c * KN_CHUNK, hi = (lo + KN_CHUNK < len) ? lo + KN_CHUNK : len;
    Fr rem = fr_load(R + c);
    for (uint64_t i = hi; i-- > lo;) {
        // q[i] = (value of the running remainder before absorbing p[i]) for i < len - 1; the top coefficient only seeds it
        if (i + 1 < len) fr_store(q + i, rem);
        rem = fr_add(fr_load(p + i), fr_mul(rem, x));
        if (i + 1 == len) rem = fr_load(p + i);
    }
}

// ---- the same passes on one rank's slice of the arrays (sharded opening).  The 2N-element array t lives as contiguous
// slices of S = 2N / world elements; `base` = rank * S.  shifted(i) = element base + i - offset of the array BEFORE the pass:
// for offset < S the slice's own element i - offset, or for i < offset the lower neighbour's top `offset` elements (halo[i]);
// for offset >= S the slice of rank r - offset / S (halo[i]); zeros where no such rank exists (the halo is zero-filled).
__global__ void __launch_bounds__(256) k_kn_pass_sh(const Fr* __restrict__ t, const Fr* __restrict__ halo, Fr c, uint64_t curr,
                                                     uint64_t offset, uint64_t base, uint64_t S, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const uint64_t idx = base + i;
    Fr v = fr_zero();
    if (idx < curr + offset) {
        v = fr_load(t + i);
        if (idx < curr) v = fr_sub(v, fr_mul(v, c));
        if (idx >= offset && idx - offset < curr) {
            const Fr sh = (offset < S && i >= offset) ? fr_load(t + i - offset) : fr_load(halo + i);
            v = fr_add(v, fr_mul(sh, c));
        }
    }
    fr_store(out + i, v);
}

// chunked Horner with a chunk length that may be below 64 (small slices): H[c] = sum_{i < chunk} p[chunk c + i] x^i
__global__ void __launch_bounds__(128) k_kn_chunk_eval_n(const Fr* __restrict__ p, uint64_t len, Fr x, uint64_t nchunks, uint32_t chunk,
                                                          Fr* __restrict__ H) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t lo = c * chunk, hi = (lo + chunk < len) ? lo + chunk : len;
    Fr acc = fr_zero();
    for (uint64_t i = hi; i-- > lo;) acc = fr_add(fr_mul(acc, x), fr_load(p + i));
    fr_store(H + c, acc);
}

// quotient coefficients of this slice: rem enters chunk c from above (R[c], which includes what the higher ranks carry down);
// coefficient base + i of the quotient exists for base + i + 1 < total_len
__global__ void __launch_bounds__(128) k_kn_chunk_div_sh(const Fr* __restrict__ p, uint64_t len, Fr x, uint64_t nchunks, uint32_t chunk,
                                                          const Fr* __restrict__ R, uint64_t base, uint64_t total_len, Fr* __restrict__ q) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t lo = c * chunk, hi = (lo + chunk < len) ? lo + chunk : len;
    Fr rem = fr_load(R + c);
    for (uint64_t i = hi; i-- > lo;) {
        if (base + i + 1 < total_len) fr_store(q + i, rem);
        rem = fr_add(fr_load(p + i), fr_mul(rem, x));
    }
}

// t[i] *= inv[i], zero at local index zero_at (~0: none)
__global__ void __launch_bounds__(256) k_kn_scale_sh(Fr* __restrict__ t, const Fr* __restrict__ inv, uint64_t zero_at, uint64_t n_inv,
                                                      uint64_t total) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    fr_store(t + idx, (idx == zero_at || idx >= n_inv) ? fr_zero() : fr_mul(fr_load(t + idx), fr_load(inv + idx)));
}

// inverses[first + i] for i < count (gm_knuckles_setup_range)
__global__ void __launch_bounds__(128) k_kn_inverses_range(Fr k, Fr k_n1, uint64_t n, uint64_t first, uint64_t count, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t s = first + i;
    Fr v = fr_sub(fr_pow_u64(k, s), k_n1);
    if (s == n - 1) v = fr_add(v, fr_one());
    fr_store(out + i, fr_inv(v));
}

}  // namespace gm

namespace {

#define TRY(x)                      \
    do {                            \
        int32_t rc__ = (x);         \
        if (rc__) return rc__;      \
    } while (0)

Fr host_pow(Fr b, uint64_t e) {
    Fr acc = fr_one();
    while (e) {
        if (e & 1) acc = fr_mul(acc, b);
        b = fr_sqr(b);
        e >>= 1;
    }
    return acc;
}

// p(x) and, optionally, the quotient p / (X - x) (len - 1 coefficients) of a device polynomial
int32_t eval_and_divide(const Fr* d_p, uint64_t len, const Fr& x, Fr* out_ev, Fr* d_q, hipStream_t s) {
    const uint64_t nchunks = (len + KN_CHUNK - 1) / KN_CHUNK;
    DevBuf dH;
    TRY(dH.alloc(nchunks * sizeof(Fr)));
    hipLaunchKernelGGL(k_kn_chunk_eval, dim3(ceil_div(nchunks, 128)), dim3(128), 0, s, d_p, len, x, nchunks, dH.fr());
    GM_LAUNCH_CHECK();
    std::vector<Fr> H(nchunks), R(nchunks);
    GM_HIP(hipMemcpyAsync(H.data(), dH.p, nchunks * sizeof(Fr), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    // R[c] = sum_{c' > c} H[c'] x^(64 (c' - c - 1)): the remainder the serial loop carries into chunk c from above
    const Fr x64 = host_pow(x, KN_CHUNK);
    Fr run = fr_zero();
    for (uint64_t c = nchunks; c-- > 0;) {
        R[c] = run;
        run = fr_add(H[c], fr_mul(run, x64));
    }
    *out_ev = run;
    if (d_q) {
        // the top chunk is seeded by the top coefficient itself (rem = poly[len - 1]); the kernel handles that case
        GM_HIP(hipMemcpyAsync(dH.p, R.data(), nchunks * sizeof(Fr), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_kn_chunk_div, dim3(ceil_div(nchunks, 128)), dim3(128), 0, s, d_p, len, x, nchunks, dH.fr(), d_q);
        GM_LAUNCH_CHECK();
        GM_HIP(hipStreamSynchronize(s));
    }
    return GM_OK;
}

G1Jac host_mul(const G1Jac& p, const Fr& k_mont) {
    const Fr k = fr_from_mont(k_mont);
    G1Jac acc = g1_inf();
    for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            acc = g1_dbl(acc);
            if ((k.l[i] >> b) & 1) acc = g1_add(acc, p);
        }
    return acc;
}

G1Jac aff_in(const uint64_t* h) {
    G1Aff a;
    memcpy(&a, h, sizeof(G1Aff));
    return g1_from_aff(a);
}
void aff_out(uint64_t* h, const G1Jac& p) {
    const G1Aff a = g1_to_aff(p);
    memcpy(h, &a, sizeof(G1Aff));
}

// verify_reduce_to_pair (kzg.rs:46-59): ([Q] * at - g0 * opening + [P], [Q])
void reduce_to_pair(const G1Jac& g0, const G1Jac& P, const G1Jac& Q, const Fr& at, const Fr& opening, G1Jac* a, G1Jac* b) {
    *a = g1_add(g1_add(host_mul(Q, at), g1_neg(host_mul(g0, opening))), P);
    *b = Q;
}

struct KnTape {
    const uint64_t* tape;
    uint64_t n, pos;
    const gm_transcript* cb;
    int32_t challenge(Fr* out) {
        Fr c;
        if (cb) {
            const int32_t rc = cb->challenge(cb->ctx, 1, 128, reinterpret_cast<uint64_t*>(&c));
            if (rc) return set_err(GM_ERR_STATE, "transcript challenge callback failed with %d", rc);
        } else {
            if (pos >= n) return set_err(GM_ERR_INVALID, "challenge tape exhausted after %llu challenges", (unsigned long long)pos);
            memcpy(&c, tape + 4 * pos, 32);
        }
        pos++;
        *out = fr_to_mont(c);
        return GM_OK;
    }
    int32_t scalars(const Fr* v, uint64_t cnt) {
        if (cb && cb->write_scalars) {
            const int32_t rc = cb->write_scalars(cb->ctx, reinterpret_cast<const uint64_t*>(v), cnt);
            if (rc) return set_err(GM_ERR_STATE, "transcript write_scalars callback failed with %d", rc);
        }
        return GM_OK;
    }
    int32_t point(const uint64_t* aff) {
        if (cb && cb->write_points) {
            const int32_t rc = cb->write_points(cb->ctx, aff, 1);
            if (rc) return set_err(GM_ERR_STATE, "transcript write_points callback failed with %d", rc);
        }
        return GM_OK;
    }
};

int32_t knuckles_open(const uint64_t* d_basis_aff, const uint64_t* d_inverses, const uint64_t* h_k, uint32_t num_vars,
                      const uint64_t* d_poly, uint64_t poly_len, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                      const uint64_t* h_commitment_aff, KnTape* tr, uint64_t* h_proof, uint64_t* h_pair, void* stream) {
    GM_REQUIRE(d_basis_aff && d_inverses && h_k && d_poly && h_point && h_claimed_ev && h_commitment_aff && h_proof && h_pair,
               "null argument");
    GM_REQUIRE(num_vars >= 1 && num_vars <= 26, "bad num_vars");
    const uint64_t N = 1ull << num_vars, total = 2 * N - 1;
    GM_REQUIRE(poly_len >= 1 && poly_len <= N, "poly.len() must be in 1..=2^num_vars (knuckles.rs:118)");
    hipStream_t s = as_stream(stream);
    Fr k, claimed;
    memcpy(&k, h_k, 32);
    memcpy(&claimed, h_claimed_ev, 32);
    std::vector<Fr> pt(num_vars);
    memcpy(pt.data(), h_point, num_vars * sizeof(Fr));

    // ---- compute_t (knuckles.rs:111-154): pt reversed, multiply by 1 - pt
    DevBuf ta, tb, plt, quo;
    TRY(ta.alloc(total * sizeof(Fr)));
    TRY(tb.alloc(total * sizeof(Fr)));
    GM_HIP(hipMemsetAsync(ta.p, 0, total * sizeof(Fr), s));
    GM_HIP(hipMemcpyAsync(ta.p, d_poly, poly_len * sizeof(Fr), hipMemcpyDeviceToDevice, s));
    Fr *cur = ta.fr(), *nxt = tb.fr();
    uint64_t curr = N;
    for (uint32_t i = 0; i < num_vars; i++) {
        const Fr c = fr_sub(fr_one(), pt[num_vars - 1 - i]);
        const uint64_t offset = 1ull << i;
        hipLaunchKernelGGL(k_kn_pass, dim3(ceil_div(total, 256)), dim3(256), 0, s, cur, c, curr, offset, total, nxt);
        GM_LAUNCH_CHECK();
        curr += offset;
        Fr* sw = cur; cur = nxt; nxt = sw;
    }
    Fr opening;
    GM_HIP(hipMemcpyAsync(&opening, cur + (N - 1), sizeof(Fr), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    GM_REQUIRE(fr_eq(opening, claimed), "Incorrect opening claim (opening.rs:49 / knuckles.rs:176)");
    hipLaunchKernelGGL(k_kn_scale, dim3(ceil_div(total, 256)), dim3(256), 0, s, cur, reinterpret_cast<const Fr*>(d_inverses), N - 1, total);
    GM_LAUNCH_CHECK();
    const Fr* t = cur;
    Fr* scratch = nxt;  // free again

    uint64_t* t_comm = h_proof;            // 12
    Fr* t_x = reinterpret_cast<Fr*>(h_proof + 12);
    Fr* p_x = reinterpret_cast<Fr*>(h_proof + 16);
    uint64_t* plt_proof = h_proof + 20;    // 12
    Fr* t_kx = reinterpret_cast<Fr*>(h_proof + 32);
    uint64_t* tkx_proof = h_proof + 36;    // 12
    // t_comm = commit(t)
    TRY(gm_g1_msm(d_basis_aff, reinterpret_cast<const uint64_t*>(t), total, 1, 255, t_comm, stream));
    TRY(tr->point(t_comm));
    Fr x;
    TRY(tr->challenge(&x));
    const Fr kx = fr_mul(x, k);
    TRY(eval_and_divide(t, total, x, t_x, nullptr, s));
    TRY(eval_and_divide(reinterpret_cast<const Fr*>(d_poly), poly_len, x, p_x, nullptr, s));
    {
        Fr two[2] = {*t_x, *p_x};
        TRY(tr->scalars(two, 2));
    }
    Fr lambda;
    TRY(tr->challenge(&lambda));
    // p_lt = lambda * t + poly (zero-padded), opened at x
    hipLaunchKernelGGL(k_kn_plt, dim3(ceil_div(total, 256)), dim3(256), 0, s, t, reinterpret_cast<const Fr*>(d_poly), poly_len, lambda,
                       total, scratch);
    GM_LAUNCH_CHECK();
    TRY(quo.alloc(total * sizeof(Fr)));
    Fr plt_at_x;
    TRY(eval_and_divide(scratch, total, x, &plt_at_x, quo.fr(), s));
    TRY(gm_g1_msm(d_basis_aff, reinterpret_cast<const uint64_t*>(quo.p), total - 1, 1, 255, plt_proof, stream));
    TRY(tr->point(plt_proof));
    // open t at kx
    TRY(eval_and_divide(t, total, kx, t_kx, quo.fr(), s));
    TRY(tr->scalars(t_kx, 1));
    TRY(gm_g1_msm(d_basis_aff, reinterpret_cast<const uint64_t*>(quo.p), total - 1, 1, 255, tkx_proof, stream));
    TRY(tr->point(tkx_proof));
    Fr fin;
    TRY(tr->challenge(&fin));
    // deferred pairing pair (opening.rs:90-96)
    G1Aff g0a;
    GM_HIP(hipMemcpyAsync(&g0a, d_basis_aff, sizeof(G1Aff), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    const G1Jac g0 = g1_from_aff(g0a);
    const G1Jac T = aff_in(t_comm), C = aff_in(h_commitment_aff);
    const G1Jac p_lt_comm = g1_add(host_mul(T, lambda), C);
    const Fr p_lt_open = fr_add(fr_mul(*t_x, lambda), *p_x);
    G1Jac a0, b0, a1, b1;
    reduce_to_pair(g0, p_lt_comm, aff_in(plt_proof), x, p_lt_open, &a0, &b0);
    reduce_to_pair(g0, T, aff_in(tkx_proof), kx, *t_kx, &a1, &b1);
    aff_out(h_pair, g1_add(a0, host_mul(a1, fin)));
    aff_out(h_pair + 12, g1_add(b0, host_mul(b1, fin)));
    return GM_OK;
}

// ================================================================================================================ sharded opening
// KnucklesOpeningProtocol::prove with the polynomial, the array t, the inverses table and the KEY distributed over the ranks of a
// gm_comm (config E: N = 2^28, the key 51 GB, t 16 GiB).  Rank r holds
//   poly   [r N / G, (r + 1) N / G)              (the slice form the sharded multi-open reduction leaves its columns in)
//   t, inverses, quotients, key   [r S, (r + 1) S) with S = 2N / G   (clipped to the 2N - 1 / 2N - 2 entries that exist)
// Every step is linear in the polynomial: compute_t's passes are local but for a shifted read (a halo from the lower neighbour
// while offset < S, a whole slice from rank r - offset / S above), evaluations and quotients chain the ranks' slice values the
// way the chunks chain inside a slice, and each commitment is the rank's MSM over its key range, combined by one all-gather of
// a group element (gm_g1_combine_parts).  Same transcript, same proof and pairing pair on every rank as the unsharded opening.
int32_t eval_and_divide_sh(const Shard& sh, const Fr* d_p, uint64_t S, uint64_t total_len, const Fr& x, Fr* out_ev, Fr* d_q, hipStream_t s) {
    const uint64_t base = (uint64_t)sh.rank * S;
    const uint64_t len = base >= total_len ? 0 : (total_len - base < S ? total_len - base : S);
    const uint32_t chunk = (uint32_t)(S < KN_CHUNK ? S : KN_CHUNK);
    const uint64_t nchunks = (len + chunk - 1) / chunk;
    DevBuf dH;
    TRY(dH.alloc((nchunks ? nchunks : 1) * sizeof(Fr)));
    std::vector<Fr> H(nchunks), R(nchunks);
    if (nchunks) {
        hipLaunchKernelGGL(k_kn_chunk_eval_n, dim3(ceil_div(nchunks, 128)), dim3(128), 0, s, d_p, len, x, nchunks, chunk, dH.fr());
        GM_LAUNCH_CHECK();
        GM_HIP(hipMemcpyAsync(H.data(), dH.p, nchunks * sizeof(Fr), hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
    }
    const Fr xc = host_pow(x, chunk), xS = host_pow(x, S);
    Fr mine = fr_zero();   // E_r = sum_j p[base + j] x^j
    for (uint64_t c = nchunks; c-- > 0;) mine = fr_add(H[c], fr_mul(mine, xc));
    std::vector<char> all;
    TRY(shard_all_gather(sh, &mine, sizeof(Fr), &all));
    const Fr* E = reinterpret_cast<const Fr*>(all.data());
    Fr ev = fr_zero(), carry = fr_zero();   // carry = sum_{r' > r} E_r' x^(S (r' - r - 1)): the remainder entering this slice from above
    for (uint32_t r = sh.world; r-- > 0;) {
        if (r == sh.rank) carry = ev;
        ev = fr_add(E[r], fr_mul(ev, xS));
    }
    *out_ev = ev;
    if (d_q && nchunks) {
        if (!fr_is_zero(carry) && len != S) return set_err(GM_ERR_STATE, "sharded division: a carry enters a clipped slice");
        Fr run = carry;
        for (uint64_t c = nchunks; c-- > 0;) {
            R[c] = run;
            run = fr_add(H[c], fr_mul(run, xc));
        }
        GM_HIP(hipMemcpyAsync(dH.p, R.data(), nchunks * sizeof(Fr), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_kn_chunk_div_sh, dim3(ceil_div(nchunks, 128)), dim3(128), 0, s, d_p, len, x, nchunks, chunk, dH.fr(), base,
                           total_len, d_q);
        GM_LAUNCH_CHECK();
        GM_HIP(hipStreamSynchronize(s));
    }
    return GM_OK;
}

// commit of a distributed coefficient vector: this rank's `count` coefficients (global indices first .. first + count) against its
// range of the key; one group element per rank crosses the communicator
int32_t commit_slices(const Shard& sh, const gm_key_view* key, uint64_t first, const Fr* d_coefs, uint64_t count, uint64_t* h_out_aff,
                      const char* what, void* stream) {
    G1Jac part = g1_inf();
    if (count) {
        GM_KEY_RANGE(kp, key, first, count, what);
        uint64_t a12[12];
        TRY(gm_g1_msm(kp, reinterpret_cast<const uint64_t*>(d_coefs), count, 1, 255, a12, stream));
        part = aff_in(a12);
    }
    return gm_g1_combine_parts(sh.comm, reinterpret_cast<const uint64_t*>(&part), 1, h_out_aff);
}

int32_t knuckles_open_sharded(const Shard& sh, const gm_key_view* key, const uint64_t* d_inverses_slice, const uint64_t* h_k,
                              uint32_t num_vars, const uint64_t* d_poly_slice, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                              const uint64_t* h_commitment_aff, KnTape* tr, uint64_t* h_proof, uint64_t* h_pair, void* stream) {
    GM_REQUIRE(sh.comm && key && d_inverses_slice && h_k && d_poly_slice && h_point && h_claimed_ev && h_commitment_aff && h_proof && h_pair,
               "null argument");
    const uint32_t G = sh.world;
    GM_REQUIRE(G >= 2 && (G & (G - 1)) == 0 && sh.rank < G, "bad sharding context");
    GM_REQUIRE(num_vars >= 1 && num_vars <= 30 && (1ull << num_vars) >= G, "bad num_vars / more ranks than coefficients");
    const uint64_t N = 1ull << num_vars, total = 2 * N - 1, S = 2 * N / G, SL = N / G, base = (uint64_t)sh.rank * S;
    const uint64_t mine = base >= total ? 0 : (total - base < S ? total - base : S);        // entries of t this rank holds
    const uint64_t mine_q = base >= total - 1 ? 0 : (total - 1 - base < S ? total - 1 - base : S);   // ... of a quotient (2N - 2)
    hipStream_t s = as_stream(stream);
    Fr k, claimed;
    memcpy(&k, h_k, 32);
    memcpy(&claimed, h_claimed_ev, 32);
    std::vector<Fr> pt(num_vars);
    memcpy(pt.data(), h_point, num_vars * sizeof(Fr));
    bool host_staged = false;

    // the polynomial, re-spread from slices of N / G to this rank's S = 2 (N / G) entries of the 2N array (zeros from N on)
    DevBuf ta, tb, halo, pol, quo;
    {
        ExportableScope exported;   // the passes read halos of each other's ta / tb (dist_read)
        TRY(ta.alloc(S * sizeof(Fr))); TRY(tb.alloc(S * sizeof(Fr)));
    }
    TRY(halo.alloc(S * sizeof(Fr))); TRY(pol.alloc(S * sizeof(Fr)));
    TRY(dist_read(sh, reinterpret_cast<const Fr*>(d_poly_slice), SL, (int64_t)base, S, pol.fr(), &host_staged, s));
    GM_HIP(hipMemcpyAsync(ta.p, pol.p, S * sizeof(Fr), hipMemcpyDeviceToDevice, s));
    // ---- compute_t (knuckles.rs:111-154)
    Fr *cur = ta.fr(), *nxt = tb.fr();
    uint64_t curr = N;
    for (uint32_t i = 0; i < num_vars; i++) {
        const Fr c = fr_sub(fr_one(), pt[num_vars - 1 - i]);
        const uint64_t offset = 1ull << i;
        // the shifted read: [base - offset, base - offset + min(offset, S)) of the array as it stands
        TRY(dist_read(sh, cur, S, (int64_t)base - (int64_t)offset, offset < S ? offset : S, halo.fr(), &host_staged, s));
        hipLaunchKernelGGL(k_kn_pass_sh, dim3(ceil_div(S, 256)), dim3(256), 0, s, cur, halo.fr(), c, curr, offset, base, S, nxt);
        GM_LAUNCH_CHECK();
        curr += offset;
        Fr* sw = cur; cur = nxt; nxt = sw;
    }
    // opening = t[N - 1]: its owner announces it
    Fr opening = fr_zero();
    const uint32_t owner = (uint32_t)((N - 1) / S);
    if (sh.rank == owner) {
        GM_HIP(hipMemcpyAsync(&opening, cur + ((N - 1) - base), sizeof(Fr), hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
    }
    {
        std::vector<char> all;
        TRY(shard_all_gather(sh, &opening, sizeof(Fr), &all));
        memcpy(&opening, all.data() + (size_t)owner * sizeof(Fr), sizeof(Fr));
    }
    GM_REQUIRE(fr_eq(opening, claimed), "Incorrect opening claim (opening.rs:49 / knuckles.rs:176)");
    hipLaunchKernelGGL(k_kn_scale_sh, dim3(ceil_div(S, 256)), dim3(256), 0, s, cur, reinterpret_cast<const Fr*>(d_inverses_slice),
                       sh.rank == owner ? (N - 1) - base : ~0ull, mine, S);
    GM_LAUNCH_CHECK();
    const Fr* t = cur;
    Fr* scratch = nxt;

    uint64_t* t_comm = h_proof;            // 12
    Fr* t_x = reinterpret_cast<Fr*>(h_proof + 12);
    Fr* p_x = reinterpret_cast<Fr*>(h_proof + 16);
    uint64_t* plt_proof = h_proof + 20;    // 12
    Fr* t_kx = reinterpret_cast<Fr*>(h_proof + 32);
    uint64_t* tkx_proof = h_proof + 36;    // 12
    TRY(commit_slices(sh, key, base, t, mine, t_comm, "t_comm", stream));
    TRY(tr->point(t_comm));
    Fr x;
    TRY(tr->challenge(&x));
    const Fr kx = fr_mul(x, k);
    TRY(eval_and_divide_sh(sh, t, S, total, x, t_x, nullptr, s));
    TRY(eval_and_divide_sh(sh, pol.fr(), S, N, x, p_x, nullptr, s));
    {
        Fr two[2] = {*t_x, *p_x};
        TRY(tr->scalars(two, 2));
    }
    Fr lambda;
    TRY(tr->challenge(&lambda));
    // p_lt = lambda t + poly, opened at x
    hipLaunchKernelGGL(k_kn_plt, dim3(ceil_div(S, 256)), dim3(256), 0, s, t, pol.fr(), S, lambda, S, scratch);
    GM_LAUNCH_CHECK();
    TRY(quo.alloc(S * sizeof(Fr)));
    Fr plt_at_x;
    TRY(eval_and_divide_sh(sh, scratch, S, total, x, &plt_at_x, quo.fr(), s));
    TRY(commit_slices(sh, key, base, quo.fr(), mine_q, plt_proof, "p_lt opening", stream));
    TRY(tr->point(plt_proof));
    // open t at kx
    TRY(eval_and_divide_sh(sh, t, S, total, kx, t_kx, quo.fr(), s));
    TRY(tr->scalars(t_kx, 1));
    TRY(commit_slices(sh, key, base, quo.fr(), mine_q, tkx_proof, "t opening at kx", stream));
    TRY(tr->point(tkx_proof));
    Fr fin;
    TRY(tr->challenge(&fin));
    // deferred pairing pair (opening.rs:90-96): g0 = key[0] from the rank that holds it
    G1Aff g0a;
    memset(&g0a, 0, sizeof(g0a));
    if (sh.rank == 0) {
        GM_KEY_RANGE(k0, key, 0, 1, "g0");
        GM_HIP(hipMemcpyAsync(&g0a, k0, sizeof(G1Aff), hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
    }
    {
        std::vector<char> all;
        TRY(shard_all_gather(sh, &g0a, sizeof(G1Aff), &all));
        memcpy(&g0a, all.data(), sizeof(G1Aff));
    }
    const G1Jac g0 = g1_from_aff(g0a);
    const G1Jac T = aff_in(t_comm), C = aff_in(h_commitment_aff);
    const G1Jac p_lt_comm = g1_add(host_mul(T, lambda), C);
    const Fr p_lt_open = fr_add(fr_mul(*t_x, lambda), *p_x);
    G1Jac a0, b0, a1, b1;
    reduce_to_pair(g0, p_lt_comm, aff_in(plt_proof), x, p_lt_open, &a0, &b0);
    reduce_to_pair(g0, T, aff_in(tkx_proof), kx, *t_kx, &a1, &b1);
    aff_out(h_pair, g1_add(a0, host_mul(a1, fin)));
    aff_out(h_pair + 12, g1_add(b0, host_mul(b1, fin)));
    return GM_OK;
}

}  // namespace

// div_by_linear (kzg.rs:73-81) and ev (kzg.rs:142-150) on a device polynomial of `len` coefficients (lowest first):
// h_ev <- poly(pt) (== the remainder), d_quotient (len - 1 coefficients, may be NULL) <- poly / (X - pt)
extern "C" int32_t gm_kzg_div_by_linear(const uint64_t* d_poly, uint64_t len, const uint64_t* h_pt, uint64_t* d_quotient,
                                        uint64_t* h_ev, void* stream) {
    GM_REQUIRE(d_poly && h_pt && h_ev && len >= 1, "bad argument");
    Fr x, e;
    memcpy(&x, h_pt, sizeof(Fr));
    TRY(eval_and_divide(reinterpret_cast<const Fr*>(d_poly), len, x, &e, reinterpret_cast<Fr*>(d_quotient), as_stream(stream)));
    memcpy(h_ev, &e, sizeof(Fr));
    return GM_OK;
}

extern "C" int32_t gm_knuckles_setup(const uint64_t* h_k, uint32_t num_vars, uint64_t* d_inverses, void* stream) {
    GM_REQUIRE(h_k && d_inverses && num_vars >= 1 && num_vars <= 26, "bad argument");
    Fr k;
    memcpy(&k, h_k, 32);
    const uint64_t n = 1ull << num_vars, total = 2 * n - 1;
    hipLaunchKernelGGL(k_kn_inverses, dim3(ceil_div(total, 128)), dim3(128), 0, as_stream(stream), k, host_pow(k, n - 1), n, total,
                       reinterpret_cast<Fr*>(d_inverses));
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int32_t gm_knuckles_open(const uint64_t* d_basis_aff, const uint64_t* d_inverses, const uint64_t* h_k, uint32_t num_vars,
                                    const uint64_t* d_poly, uint64_t poly_len, const uint64_t* h_point, const uint64_t* h_claimed_ev,
                                    const uint64_t* h_commitment_aff, const uint64_t* h_tape, uint64_t n_tape, uint64_t* h_proof,
                                    uint64_t* h_pair, void* stream) {
    GM_REQUIRE(h_tape, "null tape");
    KnTape tr{h_tape, n_tape, 0, nullptr};
    return knuckles_open(d_basis_aff, d_inverses, h_k, num_vars, d_poly, poly_len, h_point, h_claimed_ev, h_commitment_aff, &tr, h_proof,
                         h_pair, stream);
}

extern "C" int32_t gm_knuckles_open_tr(const uint64_t* d_basis_aff, const uint64_t* d_inverses, const uint64_t* h_k, uint32_t num_vars,
                                       const uint64_t* d_poly, uint64_t poly_len, const uint64_t* h_point,
                                       const uint64_t* h_claimed_ev, const uint64_t* h_commitment_aff, const gm_transcript* tr,
                                       uint64_t* h_proof, uint64_t* h_pair, void* stream) {
    GM_REQUIRE(tr && tr->challenge, "null transcript");
    KnTape t{nullptr, 0, 0, tr};
    return knuckles_open(d_basis_aff, d_inverses, h_k, num_vars, d_poly, poly_len, h_point, h_claimed_ev, h_commitment_aff, &t, h_proof,
                         h_pair, stream);
}

// One rank's range [first, first + count) of the `inverses` table (a rank of a sharded opening holds 2N / world entries of it)
extern "C" int32_t gm_knuckles_setup_range(const uint64_t* h_k, uint32_t num_vars, uint64_t first, uint64_t count, uint64_t* d_inverses,
                                           void* stream) {
    GM_REQUIRE(h_k && d_inverses && num_vars >= 1 && num_vars <= 30, "bad argument");
    Fr k;
    memcpy(&k, h_k, 32);
    const uint64_t n = 1ull << num_vars, total = 2 * n - 1;
    GM_REQUIRE(first <= total && count <= total - first, "range outside the table");
    if (!count) return GM_OK;
    hipLaunchKernelGGL(k_kn_inverses_range, dim3(ceil_div(count, 128)), dim3(128), 0, as_stream(stream), k, host_pow(k, n - 1), n, first,
                       count, reinterpret_cast<Fr*>(d_inverses));
    GM_LAUNCH_CHECK();
    return GM_OK;
}

static int32_t shard_of(const gm_comm* comm, Shard* sh) {
    GM_REQUIRE(comm && comm->all_gather && comm->rank < comm->world && (comm->world & (comm->world - 1)) == 0, "bad gm_comm");
    sh->comm = comm; sh->rank = comm->rank; sh->world = comm->world; sh->lg = 0;
    while ((1u << sh->lg) < comm->world) sh->lg++;
    return GM_OK;
}

// The opening with polynomial, key and inverses distributed over the ranks of `comm` (collective; see knuckles_open_sharded):
//   d_poly_slice       this rank's 2^num_vars / world coefficients (zero-padded polynomial)
//   d_inverses_slice   gm_knuckles_setup_range(first = rank * S, count = min(S, 2N - 1 - first)), S = 2^(num_vars + 1) / world
//   key                must hold points [rank * S, rank * S + that count) (and point 0 on rank 0)
// Same proof and pair on every rank as gm_knuckles_open over the whole polynomial and key.
extern "C" int32_t gm_knuckles_open_sharded(const gm_comm* comm, const gm_key_view* key, const uint64_t* d_inverses_slice,
                                            const uint64_t* h_k, uint32_t num_vars, const uint64_t* d_poly_slice, const uint64_t* h_point,
                                            const uint64_t* h_claimed_ev, const uint64_t* h_commitment_aff, const uint64_t* h_tape,
                                            uint64_t n_tape, uint64_t* h_proof, uint64_t* h_pair, void* stream) {
    GM_REQUIRE(h_tape, "null tape");
    Shard sh;
    TRY(shard_of(comm, &sh));
    KnTape tr{h_tape, n_tape, 0, nullptr};
    return knuckles_open_sharded(sh, key, d_inverses_slice, h_k, num_vars, d_poly_slice, h_point, h_claimed_ev, h_commitment_aff, &tr,
                                 h_proof, h_pair, stream);
}

extern "C" int32_t gm_knuckles_open_sharded_tr(const gm_comm* comm, const gm_key_view* key, const uint64_t* d_inverses_slice,
                                               const uint64_t* h_k, uint32_t num_vars, const uint64_t* d_poly_slice,
                                               const uint64_t* h_point, const uint64_t* h_claimed_ev, const uint64_t* h_commitment_aff,
                                               const gm_transcript* tr, uint64_t* h_proof, uint64_t* h_pair, void* stream) {
    GM_REQUIRE(tr && tr->challenge, "null transcript");
    Shard sh;
    TRY(shard_of(comm, &sh));
    KnTape t{nullptr, 0, 0, tr};
    return knuckles_open_sharded(sh, key, d_inverses_slice, h_k, num_vars, d_poly_slice, h_point, h_claimed_ev, h_commitment_aff, &t,
                                 h_proof, h_pair, stream);
}
