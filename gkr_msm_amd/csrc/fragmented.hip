// gen-1 fragmented multilinear polynomials on the device: FragmentedPoly{data, consts, shape} for ANY shape
// (/root/reference/src/polynomial/fragmented.rs) and the shape-aware eq tables of EqPoly (/root/reference/src/copoly.rs).
//
// `gkr_msm_prove` itself only ever builds Shape::full (gkr_msm_simple.rs:150-186), which the dense kernels of poly.hip
// serve; this file is the general case behind the same seams (FragmentedPoly::{split, bind, into_vec},
// Shape::full_split, compute_segment_split, EqPoly::materialize_eq_with_shape), so that a caller holding a ragged shape --
// constants standing for padded tails -- keeps them compressed on the device too.
//
// Design: the shape (a handful of fragments) is host data and is recomputed by the same rules as the reference
// (fragments are part of the contract: they fix which cells are data and which constant index is which); the cells move
// by position: data cell j of the target lies at polynomial index pos = frag.start + (j - frag.mem_idx), and its two
// sources are polynomial indices 2 pos and 2 pos + 1 of the source, found by binary search over the source fragments.
// One thread per target cell, coalesced 32-byte stores; nothing is ever expanded to the dense 2^n vector.
#include <algorithm>
#include <vector>

#include "fr.hip.h"
#include "internal.hpp"

namespace gm {
namespace {

struct Frag {  // device copy of a fragment
    uint64_t mem_idx, len, start;
    uint32_t is_const, pad;
};

constexpr uint64_t MERGE_THRESH = 2;  // fragmented.rs:65

struct HostShape {
    std::vector<gm_fragment> fr;
    uint64_t data_len = 0, num_consts = 0, dedup_consts_len = 0;
    uint64_t len() const { return fr.empty() ? 0 : fr.back().start + fr.back().len; }
};

// Shape::new -> finalize (fragmented.rs:94-99, 168-183): fragments taken as given, counters recomputed, invariants checked
int32_t shape_from(const gm_fragment* f, uint32_t n, uint64_t num_consts, HostShape* s) {
    s->fr.assign(f, f + n);
    s->num_consts = num_consts;
    s->data_len = s->dedup_consts_len = 0;
    uint64_t pos = 0;
    for (uint32_t i = 0; i < n; i++) {
        GM_REQUIRE(f[i].content <= GM_FRAG_CONSTS, "fragment %u: unknown content %u", i, f[i].content);
        GM_REQUIRE(f[i].start == pos, "fragment %u starts at %llu, expected %llu (fragments must tile the index range)", i,
                   (unsigned long long)f[i].start, (unsigned long long)pos);
        pos += f[i].len;
        if (f[i].content == GM_FRAG_DATA) {
            GM_REQUIRE(f[i].mem_idx == s->data_len, "Shape data incorrect at fragment %u: mem_idx %llu, data_len %llu", i,
                       (unsigned long long)f[i].mem_idx, (unsigned long long)s->data_len);
            s->data_len += f[i].len;
        } else {
            s->dedup_consts_len++;
            GM_REQUIRE(f[i].mem_idx < num_consts, "fragment %u: constant index %llu >= num_consts %llu", i,
                       (unsigned long long)f[i].mem_idx, (unsigned long long)num_consts);
        }
    }
    return GM_OK;
}

bool should_merge(const gm_fragment& a, const gm_fragment& b) {  // fragmented.rs:67-78
    if (a.content == GM_FRAG_DATA) return b.content == GM_FRAG_DATA || b.len < MERGE_THRESH;
    if (b.content == GM_FRAG_DATA) return false;
    return a.mem_idx == b.mem_idx;
}

void shape_add(HostShape* s, gm_fragment f) {  // Shape::add + merge_in (fragmented.rs:121-166)
    if (!s->fr.empty() && should_merge(s->fr.back(), f)) {
        gm_fragment& prev = s->fr.back();
        prev.len += f.len;
        if (prev.content == GM_FRAG_DATA) s->data_len += f.len;
        return;
    }
    if (f.content == GM_FRAG_DATA) s->data_len += f.len;
    else s->dedup_consts_len++;
    s->fr.push_back(f);
}

// Shape::full_split + prune_consts (fragmented.rs:285-364)
void shape_full_split(const HostShape& src, HostShape* l, std::vector<uint64_t>* perm) {
    l->fr.clear();
    l->num_consts = src.num_consts;
    l->data_len = l->dedup_consts_len = 0;
    for (const gm_fragment& fg : src.fr) {
        uint64_t len = fg.len, start = fg.start;
        if (start & 1) {
            if (fg.content == GM_FRAG_DATA) {
                len += 1;
                start -= 1;
            } else {
                len -= 1;
                start += 1;
                shape_add(l, gm_fragment{l->data_len, 1, (start - 2) / 2, GM_FRAG_DATA, 0});
            }
        }
        if (len & 1) len -= 1;
        if (len == 0) continue;
        if (fg.content == GM_FRAG_DATA || len / 2 < MERGE_THRESH)
            shape_add(l, gm_fragment{l->data_len, len / 2, start / 2, GM_FRAG_DATA, 0});
        else
            shape_add(l, gm_fragment{fg.mem_idx, len / 2, start / 2, GM_FRAG_CONSTS, 0});
    }
    perm->clear();
    std::vector<int64_t> hit(src.num_consts, -1);
    for (gm_fragment& fg : l->fr)
        if (fg.content == GM_FRAG_CONSTS) {
            if (hit[fg.mem_idx] < 0) {
                perm->push_back(fg.mem_idx);
                hit[fg.mem_idx] = (int64_t)perm->size() - 1;
            }
            fg.mem_idx = (uint64_t)hit[fg.mem_idx];
        }
}

// fragment that holds polynomial index `pos` (fragments tile [0, len) in order)
__device__ __forceinline__ uint32_t frag_of_pos(const Frag* f, uint32_t n, uint64_t pos) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (f[mid].start <= pos) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ Fr frag_value(const Frag* f, uint32_t n, const Fr* data, const Fr* consts, uint64_t pos) {
    const Frag g = f[frag_of_pos(f, n, pos)];
    return g.is_const ? fr_load(consts + g.mem_idx) : fr_load(data + g.mem_idx + (pos - g.start));
}
// data fragment of the target that holds data cell j: data fragments in order own consecutive mem_idx ranges
__device__ __forceinline__ uint64_t pos_of_cell(const Frag* dfr, uint32_t nd, uint64_t j) {
    uint32_t lo = 0, hi = nd - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (dfr[mid].mem_idx <= j) lo = mid;
        else hi = mid - 1;
    }
    return dfr[lo].start + (j - dfr[lo].mem_idx);
}

// FragmentedPoly::split (fragmented.rs:676-732), optionally fused with bind_from (:736-741): BIND -> out_l = l + t (r - l)
template <bool BIND>
__global__ void __launch_bounds__(256) k_frag_split(const Frag* __restrict__ src, uint32_t n_src,
                                                    const Frag* __restrict__ tgt_data, uint32_t n_tgt_data,
                                                    const Fr* __restrict__ data, const Fr* __restrict__ consts,
                                                    uint64_t n_cells, Fr* __restrict__ out_l, Fr* __restrict__ out_r, Fr t) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cells) return;
    const uint64_t pos = pos_of_cell(tgt_data, n_tgt_data, j);
    const Fr a = frag_value(src, n_src, data, consts, 2 * pos);
    const Fr b = frag_value(src, n_src, data, consts, 2 * pos + 1);
    if (BIND) {
        fr_store(out_l + j, fr_add(a, fr_mul(t, fr_sub(b, a))));
    } else {
        fr_store(out_l + j, a);
        fr_store(out_r + j, b);
    }
}

// new constants: consts[perm[i]], bound with themselves when BIND (l += t (r - l) with l == r leaves the constant)
__global__ void k_frag_consts(const Fr* __restrict__ consts, const uint64_t* __restrict__ perm, uint32_t n,
                              Fr* __restrict__ out_l, Fr* __restrict__ out_r) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr v = fr_load(consts + perm[i]);
    fr_store(out_l + i, v);
    if (out_r) fr_store(out_r + i, v);
}

// FragmentedPoly::into_vec (fragmented.rs:831-846)
__global__ void __launch_bounds__(256) k_frag_to_dense(const Frag* __restrict__ src, uint32_t n_src, const Fr* __restrict__ data,
                                                       const Fr* __restrict__ consts, uint64_t len, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) fr_store(out + i, frag_value(src, n_src, data, consts, i));
}

// eq(point, pos) * multiplier for every data cell (materialize_eq_with_shape's `values`, copoly.rs:492-567): cell j at
// polynomial index pos gets multiplier * prod_b (bit_b(pos) ? pt_b : 1 - pt_b), pt in LSB-first order in `pt_lsb`
__global__ void __launch_bounds__(256) k_frag_eq_values(const Frag* __restrict__ dfr, uint32_t nd, uint64_t n_cells,
                                                        const Fr* __restrict__ pt_lsb, const Fr* __restrict__ one_minus_pt_lsb,
                                                        uint32_t nvars, Fr mult, Fr* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cells) return;
    const uint64_t pos = pos_of_cell(dfr, nd, j);
    Fr acc = mult;
    for (uint32_t b = 0; b < nvars; b++) acc = fr_mul(acc, fr_load(((pos >> b) & 1) ? pt_lsb + b : one_minus_pt_lsb + b));
    fr_store(out + j, acc);
}

std::vector<Frag> to_dev_frags(const std::vector<gm_fragment>& v, bool data_only) {
    std::vector<Frag> o;
    for (const gm_fragment& f : v)
        if (!data_only || f.content == GM_FRAG_DATA) o.push_back(Frag{f.mem_idx, f.len, f.start, f.content == GM_FRAG_CONSTS, 0});
    return o;
}

int32_t upload(const std::vector<Frag>& v, DevBuf* b, hipStream_t s) {
    int32_t rc = b->alloc(std::max<size_t>(v.size(), 1) * sizeof(Frag));
    if (rc) return rc;
    if (!v.empty()) GM_HIP(hipMemcpyAsync(b->p, v.data(), v.size() * sizeof(Frag), hipMemcpyHostToDevice, s));
    return GM_OK;
}

int32_t frag_split_impl(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const Fr* d_data, const Fr* d_consts,
                        Fr* l_data, Fr* r_data, Fr* l_consts, Fr* r_consts, const Fr* t, hipStream_t s) {
    HostShape src, tgt;
    std::vector<uint64_t> perm;
    int32_t rc = shape_from(frags, n_frags, num_consts, &src);
    if (rc) return rc;
    GM_REQUIRE(src.len() >= 2 && (src.len() & (src.len() - 1)) == 0, "split needs a power-of-two length >= 2 (got %llu)",
               (unsigned long long)src.len());
    shape_full_split(src, &tgt, &perm);
    const std::vector<Frag> hs = to_dev_frags(src.fr, false), ht = to_dev_frags(tgt.fr, true);
    DevBuf ds, dt, dperm;
    if ((rc = upload(hs, &ds, s)) || (rc = upload(ht, &dt, s))) return rc;
    if (tgt.data_len) {
        const unsigned grid = ceil_div(tgt.data_len, 256);
        if (t)
            k_frag_split<true><<<grid, 256, 0, s>>>((const Frag*)ds.p, (uint32_t)hs.size(), (const Frag*)dt.p, (uint32_t)ht.size(), d_data,
                                                    d_consts, tgt.data_len, l_data, nullptr, *t);
        else
            k_frag_split<false><<<grid, 256, 0, s>>>((const Frag*)ds.p, (uint32_t)hs.size(), (const Frag*)dt.p, (uint32_t)ht.size(), d_data,
                                                     d_consts, tgt.data_len, l_data, r_data, fr_zero());
        GM_LAUNCH_CHECK();
    }
    if (!perm.empty()) {
        if ((rc = dperm.alloc(perm.size() * 8))) return rc;
        GM_HIP(hipMemcpyAsync(dperm.p, perm.data(), perm.size() * 8, hipMemcpyHostToDevice, s));
        k_frag_consts<<<ceil_div(perm.size(), 64), 64, 0, s>>>(d_consts, (const uint64_t*)dperm.p, (uint32_t)perm.size(), l_consts, r_consts);
        GM_LAUNCH_CHECK();
    }
    GM_HIP(hipStreamSynchronize(s));  // the staging buffers above are released on return
    return GM_OK;
}

}  // namespace
}  // namespace gm

using namespace gm;

extern "C" {

int32_t gm_frag_shape_full_split(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, gm_fragment* out_frags,
                                 uint32_t out_cap, uint32_t* n_out, uint64_t* out_perm, uint32_t perm_cap, uint32_t* n_perm,
                                 uint64_t* out_data_len) {
    GM_REQUIRE(frags && n_out && n_perm, "null argument");
    HostShape src, tgt;
    std::vector<uint64_t> perm;
    int32_t rc = shape_from(frags, n_frags, num_consts, &src);
    if (rc) return rc;
    shape_full_split(src, &tgt, &perm);
    *n_out = (uint32_t)tgt.fr.size();
    *n_perm = (uint32_t)perm.size();
    if (out_data_len) *out_data_len = tgt.data_len;
    GM_REQUIRE(tgt.fr.size() <= out_cap && perm.size() <= perm_cap, "output capacity too small (%zu fragments, %zu constants)",
               tgt.fr.size(), perm.size());
    if (out_frags) std::copy(tgt.fr.begin(), tgt.fr.end(), out_frags);
    if (out_perm) std::copy(perm.begin(), perm.end(), out_perm);
    return GM_OK;
}

int32_t gm_frag_split(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const uint64_t* d_data,
                      const uint64_t* d_consts, uint64_t* d_l_data, uint64_t* d_r_data, uint64_t* d_l_consts,
                      uint64_t* d_r_consts, void* stream) {
    GM_REQUIRE(frags && d_l_data && d_r_data, "null argument");
    return frag_split_impl(frags, n_frags, num_consts, (const Fr*)d_data, (const Fr*)d_consts, (Fr*)d_l_data, (Fr*)d_r_data,
                           (Fr*)d_l_consts, (Fr*)d_r_consts, nullptr, as_stream(stream));
}

int32_t gm_frag_bind(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const uint64_t* d_data,
                     const uint64_t* d_consts, const uint64_t* h_t, uint64_t* d_out_data, uint64_t* d_out_consts, void* stream) {
    GM_REQUIRE(frags && d_out_data && h_t, "null argument");
    Fr t;
    memcpy(&t, h_t, sizeof(Fr));
    return frag_split_impl(frags, n_frags, num_consts, (const Fr*)d_data, (const Fr*)d_consts, (Fr*)d_out_data, nullptr,
                           (Fr*)d_out_consts, nullptr, &t, as_stream(stream));
}

int32_t gm_frag_to_dense(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const uint64_t* d_data,
                         const uint64_t* d_consts, uint64_t* d_out, void* stream) {
    GM_REQUIRE(frags && d_out, "null argument");
    HostShape src;
    int32_t rc = shape_from(frags, n_frags, num_consts, &src);
    if (rc) return rc;
    if (!src.len()) return GM_OK;
    hipStream_t s = as_stream(stream);
    const std::vector<Frag> hs = to_dev_frags(src.fr, false);
    DevBuf ds;
    if ((rc = upload(hs, &ds, s))) return rc;
    k_frag_to_dense<<<ceil_div(src.len(), 256), 256, 0, s>>>((const Frag*)ds.p, (uint32_t)hs.size(), (const Fr*)d_data, (const Fr*)d_consts,
                                                             src.len(), (Fr*)d_out);
    GM_LAUNCH_CHECK();
    GM_HIP(hipStreamSynchronize(s));
    return GM_OK;
}

int32_t gm_segment_split(uint64_t start, uint64_t end, uint64_t* out_starts, uint8_t* out_loglengths, uint32_t cap, uint32_t* n_out) {
    GM_REQUIRE(n_out && start <= end, "bad segment");
    uint32_t n = 0;
    while (start < end) {  // copoly.rs:139-148
        const uint32_t tz = start ? (uint32_t)__builtin_ctzll(start) : 64, lf = 63 - (uint32_t)__builtin_clzll(end - start);
        const uint32_t ll = tz < lf ? tz : lf;
        if (n < cap) {
            if (out_starts) out_starts[n] = start;
            if (out_loglengths) out_loglengths[n] = (uint8_t)ll;
        }
        n++;
        start += (uint64_t)1 << ll;
    }
    *n_out = n;
    GM_REQUIRE(n <= cap, "output capacity too small (%u standard subsets)", n);
    return GM_OK;
}

int32_t gm_frag_eq_materialize(const gm_fragment* frags, uint32_t n_frags, uint64_t num_consts, const uint64_t* h_multiplier,
                               const uint64_t* h_point, uint32_t nvars, uint64_t* d_values, uint64_t* h_sums, void* stream) {
    GM_REQUIRE(frags && h_multiplier && (h_point || !nvars), "null argument");
    HostShape src;
    int32_t rc = shape_from(frags, n_frags, num_consts, &src);
    if (rc) return rc;
    GM_REQUIRE(nvars < 64 && src.len() == ((uint64_t)1 << nvars), "shape length %llu != 2^%u", (unsigned long long)src.len(), nvars);
    hipStream_t s = as_stream(stream);
    const Fr* pt = reinterpret_cast<const Fr*>(h_point);
    Fr mult;
    memcpy(&mult, h_multiplier, sizeof(Fr));
    // sums: a constant stands for the sum of eq over its segments = sum over standard subsets of the prefix product
    if (h_sums) {
        std::vector<Fr> sums(num_consts, fr_zero());
        for (const gm_fragment& f : src.fr) {
            if (f.content != GM_FRAG_CONSTS) continue;
            uint64_t a = f.start, e = f.start + f.len;
            while (a < e) {
                const uint32_t tz = a ? (uint32_t)__builtin_ctzll(a) : 64, lf = 63 - (uint32_t)__builtin_clzll(e - a);
                const uint32_t ll = tz < lf ? tz : lf;
                Fr m = mult;
                uint64_t prefix = a >> ll;
                for (int i = (int)(nvars - ll) - 1; i >= 0; i--) {  // pt[i] <-> index bit nvars-1-i
                    m = fr_mul(m, (prefix & 1) ? pt[i] : fr_sub(fr_one(), pt[i]));
                    prefix >>= 1;
                }
                sums[f.mem_idx] = fr_add(sums[f.mem_idx], m);
                a += (uint64_t)1 << ll;
            }
        }
        if (num_consts) memcpy(h_sums, sums.data(), num_consts * sizeof(Fr));
    }
    if (d_values && src.data_len) {
        std::vector<Fr> lsb(2 * std::max(nvars, 1u));
        for (uint32_t b = 0; b < nvars; b++) {
            lsb[b] = pt[nvars - 1 - b];
            lsb[nvars + b] = fr_sub(fr_one(), pt[nvars - 1 - b]);
        }
        DevBuf dpt, dfr;
        if ((rc = dpt.alloc(lsb.size() * sizeof(Fr)))) return rc;
        GM_HIP(hipMemcpyAsync(dpt.p, lsb.data(), lsb.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
        const std::vector<Frag> hd = to_dev_frags(src.fr, true);
        if ((rc = upload(hd, &dfr, s))) return rc;
        k_frag_eq_values<<<ceil_div(src.data_len, 256), 256, 0, s>>>((const Frag*)dfr.p, (uint32_t)hd.size(), src.data_len, dpt.fr(),
                                                                     dpt.fr() + nvars, nvars, mult, (Fr*)d_values);
        GM_LAUNCH_CHECK();
        GM_HIP(hipStreamSynchronize(s));
    }
    return GM_OK;
}

}  // extern "C"
