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
