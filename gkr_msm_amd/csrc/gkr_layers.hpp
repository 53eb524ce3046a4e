// Protocol shapes shared by the provers (prover.hip) and the verifier (verifier.hip): the layer lists of the two GKR circuits
// and the verifier-side polynomials.  Host only.
#pragma once
#include <vector>

#include "internal.hpp"
#include "segfn.hip.h"

namespace gm {

inline gm_fn mkfn(int p0, int c0, int p1 = 0, int c1 = 0) {
    gm_fn f;
    memset(&f, 0, sizeof(f));
    f.nseg = p1 ? 2 : 1;
    f.prim[0] = p0; f.count[0] = c0;
    f.prim[1] = p1; f.count[1] = c1;
    return f;
}

inline SegPlan plan_of(const gm_fn& f) {
    GmFn g;
    to_gmfn(&f, &g);
    SegPlan sp;
    seg_plan_build(g, &sp);
    return sp;
}

struct Layer {
    enum Kind { VECVEC, DENSE, SPLIT, ZEROCHECK } kind;
    gm_fn f;
    uint32_t num_vars = 0;
    bool split_hi = false;
    uint32_t split_idx = 0, bundle = 3;
};

// bintree_add::builder::protocol::build (bintree_add.rs:247-375)
inline std::vector<Layer> bintree_layers(uint32_t num_vars, uint32_t num_adds, uint32_t row_logsize, bool do_bitcheck) {
    std::vector<Layer> layers;
    for (uint32_t i = 0; i < num_adds; i++) {
        for (int step = 0; step < 3; step++) {
            Layer L;
            L.kind = (i == 0 || i + 1 < row_logsize) ? Layer::VECVEC : Layer::DENSE;
            L.num_vars = num_vars - i - 1;
            const int prim = (i == 0) ? (step == 0 ? GM_FN_AFF_L1 : step == 1 ? GM_FN_AFF_L2 : GM_FN_AFF_L3)
                                      : (step == 0 ? GM_FN_PROJ_L1 : step == 1 ? GM_FN_PROJ_L2 : GM_FN_PROJ_L3);
            L.f = (i == 0 && step == 0 && do_bitcheck) ? mkfn(GM_FN_AFF_L1, 1, GM_FN_BITCHECK, 2) : mkfn(prim, 1);
            layers.push_back(L);
            if (i == 0 && step == 0 && do_bitcheck) {
                Layer Z;
                Z.kind = Layer::ZEROCHECK;
                layers.push_back(Z);
            }
        }
        if (i != num_adds - 1) {
            Layer S;
            S.kind = Layer::SPLIT; S.split_hi = false; S.split_idx = 0; S.bundle = 3;
            layers.push_back(S);
        }
    }
    return layers;
}

// triangle_add::builder::protocol::build (triangle_add.rs:173-232)
inline std::vector<Layer> triangle_layers(uint32_t num_vars, uint32_t hi_idx) {
    std::vector<Layer> layers;
    const uint32_t num_layers = num_vars - hi_idx;
    for (uint32_t l = 0; l <= num_layers; l++) {
        Layer a, b, c;
        a.kind = b.kind = c.kind = Layer::DENSE;
        a.num_vars = b.num_vars = c.num_vars = num_vars - l;
        a.f = mkfn(GM_FN_TRI_L1, 1, GM_FN_PROJ_L1, (int)l);
        b.f = mkfn(GM_FN_PROJ_L2, (int)l + 3);
        c.f = mkfn(GM_FN_PROJ_L3, (int)l + 3);
        layers.push_back(a); layers.push_back(b); layers.push_back(c);
        if (l < num_layers) {
            Layer S;
            S.kind = Layer::SPLIT; S.split_hi = true; S.split_idx = hi_idx; S.bundle = 3;
            layers.push_back(S);
        }
    }
    return layers;
}

// EqTruncPoly::evaluate (verifier_polys.rs:108-147)
inline Fr eq_trunc_evaluate(uint32_t nv, uint64_t k, const Fr* r, const Fr* pt) {
    std::vector<Fr> partial(nv + 1);
    partial[0] = fr_one();
    for (uint32_t i = 0; i < nv; i++) {
        const uint32_t j = nv - i - 1;
        partial[i + 1] = fr_mul(partial[i], eq_bind_factor(r[j], pt[j]));
    }
    if (k >= (1ull << nv)) return partial[nv];
    Fr mult = fr_one(), acc = fr_zero();
    for (uint32_t i = 0; i < nv; i++) {
        const uint64_t left = k >> (nv - i - 1);
        const Fr prev = mult;
        if (left == 1) {
            mult = fr_mul(fr_mul(mult, pt[i]), r[i]);
            acc = fr_add(acc, fr_mul(fr_mul(fr_mul(prev, fr_sub(fr_one(), pt[i])), fr_sub(fr_one(), r[i])), partial[nv - i - 1]));
        } else {
            mult = fr_mul(fr_mul(mult, fr_sub(fr_one(), pt[i])), fr_sub(fr_one(), r[i]));
        }
        k -= left << (nv - i - 1);
    }
    return acc;
}


}  // namespace gm
