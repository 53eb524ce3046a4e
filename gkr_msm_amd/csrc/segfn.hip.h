// Device-side evaluation plan of an AlgFn: a list of "segments", each one primitive layer function
// with explicit input column indices and an output base index.
//
// The reference composes functions with Rust generics (StackedAlgFn / RepeatedAlgFn,
// /root/reference/src/cleanup/utils/algfn.rs:181-258; triangle_twisted_edwards_add_l1 is itself three
// twisted_edwards_add_l1 on (a,c),(b,d),(c,d), twisted_edwards_ops.rs:67-80).  All of them are
// block-diagonal: every output depends on at most 6 inputs.  Kernels therefore walk the segments one
// at a time with at most 6 input elements live in registers, whatever the width of the layer
// (the widest triangle layer at d_logsize = 10 has 60 input columns).
#pragma once
#include "algfn.hip.h"

namespace gm {

#define GM_MAX_COLS 64
#define GM_MAX_SEGS 32

struct Seg {
    int8_t prim;    // FN_* primitive (never FN_TRI_L1: expanded into three FN_PROJ_L1)
    int8_t n_in;
    int8_t n_out;
    int8_t out0;    // index of the first output
    int8_t in[6];   // input column indices
};

struct SegPlan {
    int nseg;
    int n_ins;
    int n_outs;
    int deg;
    Seg seg[GM_MAX_SEGS];
};

// returns false if the function does not fit the static limits
inline bool seg_plan_build(const GmFn& f, SegPlan* sp) {
    sp->nseg = 0;
    sp->n_ins = fn_n_ins(f);
    sp->n_outs = fn_n_outs(f);
    sp->deg = fn_deg(f);
    if (sp->n_ins > GM_MAX_COLS || sp->n_outs > GM_MAX_COLS) return false;
    int io = 0, oo = 0;
    for (int s = 0; s < f.nseg; s++) {
        for (int c = 0; c < f.count[s]; c++) {
            const int prim = f.prim[s];
            if (prim == FN_TRI_L1) {
                static const int pairs[3][2] = {{0, 2}, {1, 3}, {2, 3}};  // (a,c) (b,d) (c,d)
                for (int k = 0; k < 3; k++) {
                    if (sp->nseg >= GM_MAX_SEGS) return false;
                    Seg& g = sp->seg[sp->nseg++];
                    g.prim = FN_PROJ_L1; g.n_in = 6; g.n_out = 4; g.out0 = (int8_t)(oo + 4 * k);
                    for (int q = 0; q < 3; q++) {
                        g.in[q] = (int8_t)(io + 3 * pairs[k][0] + q);
                        g.in[3 + q] = (int8_t)(io + 3 * pairs[k][1] + q);
                    }
                }
            } else {
                if (sp->nseg >= GM_MAX_SEGS) return false;
                Seg& g = sp->seg[sp->nseg++];
                g.prim = (int8_t)prim; g.n_in = (int8_t)prim_n_ins(prim); g.n_out = (int8_t)prim_n_outs(prim);
                g.out0 = (int8_t)oo;
                for (int q = 0; q < g.n_in; q++) g.in[q] = (int8_t)(io + q);
                for (int q = g.n_in; q < 6; q++) g.in[q] = 0;
            }
            io += prim_n_ins(prim);
            oo += prim_n_outs(prim);
        }
    }
    return true;
}

// The primitive of a plan whose segments ALL apply the same one (the witness builders' maps: a layer function repeated over column
// bundles), 0 otherwise.  The map kernels (k_dense_map, k_dense_map_split, k_vv_map, k_vv_map_split) have an instance per primitive:
// the generic instance runs prim_exec's switch, whose widest arm sets the register count for every arm (196 VGPRs = two waves per SIMD
// for kernels that stream (k + m) x 32 B per element with 4-5 products); forcing three or four waves on it with launch bounds was
// measured (round 4) and loses to its spills: gen-1 witness 92 -> 109 / 238 ms.
inline int uniform_prim_of(const SegPlan& sp) {
    if (sp.nseg < 1) return 0;
    const int prim = sp.seg[0].prim;
    for (int s = 1; s < sp.nseg; s++)
        if (sp.seg[s].prim != prim) return 0;
    switch (prim) {
        case FN_AFF_L1: case FN_AFF_L2: case FN_AFF_L3: case FN_PROJ_L1: case FN_PROJ_L2: case FN_PROJ_L3: case FN_ID:
        case FN_PT_BIT_CHOICE: case FN_ADD_INVERSES: case FN_LOGUP_LAYER: return prim;
        default: return 0;
    }
}
// inputs / outputs of a map kernel's instance: the primitive's own counts, or the widest (6 / 4) for the generic instance (PRIM = 0)
template <int PRIM> struct PrimShape {
    static constexpr int n_in = PRIM == FN_AFF_L1 ? 4 : (PRIM == FN_AFF_L2 || PRIM == FN_AFF_L3 || PRIM == FN_PT_BIT_CHOICE) ? 3
                              : PRIM == FN_PROJ_L1 ? 6 : (PRIM == FN_PROJ_L2 || PRIM == FN_PROJ_L3 || PRIM == FN_LOGUP_LAYER) ? 4
                              : PRIM == FN_ID ? 1 : PRIM == FN_ADD_INVERSES ? 2 : 6;
    static constexpr int n_out = (PRIM == FN_AFF_L1 || PRIM == FN_AFF_L2 || PRIM == FN_AFF_L3 || PRIM == FN_PROJ_L3) ? 3
                               : (PRIM == FN_PROJ_L1 || PRIM == FN_PROJ_L2) ? 4 : PRIM == FN_ID ? 1
                               : (PRIM == FN_PT_BIT_CHOICE || PRIM == FN_ADD_INVERSES || PRIM == FN_LOGUP_LAYER) ? 2 : 4;
};
// launch KERNEL<prim> for the primitives uniform_prim_of returns, KERNEL<0> otherwise
#define GM_MAP_DISPATCH(PRIM_, LAUNCH_)                                                                          \
    switch (PRIM_) {                                                                                             \
        case FN_AFF_L1: { LAUNCH_(FN_AFF_L1); } break;                                                           \
        case FN_AFF_L2: { LAUNCH_(FN_AFF_L2); } break;                                                           \
        case FN_AFF_L3: { LAUNCH_(FN_AFF_L3); } break;                                                           \
        case FN_PROJ_L1: { LAUNCH_(FN_PROJ_L1); } break;                                                         \
        case FN_PROJ_L2: { LAUNCH_(FN_PROJ_L2); } break;                                                         \
        case FN_PROJ_L3: { LAUNCH_(FN_PROJ_L3); } break;                                                         \
        case FN_ID: { LAUNCH_(FN_ID); } break;                                                                   \
        case FN_PT_BIT_CHOICE: { LAUNCH_(FN_PT_BIT_CHOICE); } break;                                             \
        case FN_ADD_INVERSES: { LAUNCH_(FN_ADD_INVERSES); } break;                                               \
        case FN_LOGUP_LAYER: { LAUNCH_(FN_LOGUP_LAYER); } break;                                                 \
        default: { LAUNCH_(0); } break;                                                                          \
    }

// host evaluation through the plan (pads, final combinator checks)
inline void seg_plan_exec_host(const SegPlan& sp, const Fr* in, Fr* out) {
    for (int s = 0; s < sp.nseg; s++) {
        const Seg& g = sp.seg[s];
        Fr a[6], o[4];
        for (int q = 0; q < g.n_in; q++) a[q] = in[g.in[q]];
        prim_exec(g.prim, a, o);
        for (int q = 0; q < g.n_out; q++) out[g.out0 + q] = o[q];
    }
}

// ---- term split (the persistent stage kernel, sumcheck.hip).  The gamma-combined layer function sum_o gamma^o f_o(v) of a
// twisted-Edwards layer is a sum of a few products, and any partition of the products over workgroups gives the same round sums
// (the cross-block reduction adds them up, exactly).  A "part" is a pseudo-segment that evaluates one share of a segment's
// products from the inputs those products need: a lone wave needs ~0.6 us per dependent multiplication, so cutting a segment's
// chain of 9 multiplications (and 6 folds per round) into parts of 2-3 (and 2-3 folds) is what shortens a small round.
// Part segments: prim = FN_P_*, in[] = the inputs the part reads (indices into the plan's columns), out0 = the out0 of the
// segment they come from (gamma index base), n_out unused.
enum {
    FN_P_PROJ_L1_A = 64,  // v0 (G0 v4 + 5 G2 v3)            in: v0, v3, v4
    FN_P_PROJ_L1_B = 65,  // v1 (G1 v3 + G2 v4)              in: v1, v3, v4
    FN_P_PROJ_L1_C = 66,  // G3 v2 v5                        in: v2, v5
    FN_P_PROJ_L2_A = 67,  // G0 (v0 + v1) v3 + G3 v0 v1      in: v0, v1, v3
    FN_P_PROJ_L2_B = 68,  // v3 (G1 v2 + G2 v3)              in: v2, v3
    FN_P_PROJ_L3_A = 69,  // G0 m v0                         in: v0, v2, v3      (m = v2 - d v3, q = v2 + d v3)
    FN_P_PROJ_L3_B = 70,  // G1 q v1                         in: v1, v2, v3
    FN_P_PROJ_L3_C = 71,  // G2 m q                          in: v2, v3
    FN_P_AFF_L1_A = 72,   // v0 (G0 v3 + 5 G2 v2)            in: v0, v2, v3
    FN_P_AFF_L1_B = 73,   // v1 (G1 v2 + G2 v3)              in: v1, v2, v3
    FN_P_AFF_L3_A = 74,   // G0 m v0                         in: v0, v2          (m = 1 - d v2, q = 1 + d v2)
    FN_P_AFF_L3_B = 75,   // G1 q v1                         in: v1, v2
    FN_P_AFF_L3_C = 76,   // G2 m q                          in: v2
    FN_P_LOGUP_A = 77,    // v3 (G0 v0 + G1 v1)              in: v0, v1, v3      (a d + b c, b d)
    FN_P_LOGUP_B = 78,    // G0 v1 v2                        in: v1, v2
};
GM_HD bool prim_is_part(int prim) { return prim >= 64; }

// the plan with every splittable segment replaced by its parts; false (and *out untouched) when nothing splits or it does not fit
inline bool seg_plan_split_terms(const SegPlan& sp, SegPlan* out) {
    struct PartDef { int prim, n_in, in[3]; };
    static const PartDef P_L1[3] = {{FN_P_PROJ_L1_A, 3, {0, 3, 4}}, {FN_P_PROJ_L1_B, 3, {1, 3, 4}}, {FN_P_PROJ_L1_C, 2, {2, 5, 0}}};
    static const PartDef P_L2[2] = {{FN_P_PROJ_L2_A, 3, {0, 1, 3}}, {FN_P_PROJ_L2_B, 2, {2, 3, 0}}};
    static const PartDef P_L3[3] = {{FN_P_PROJ_L3_A, 3, {0, 2, 3}}, {FN_P_PROJ_L3_B, 3, {1, 2, 3}}, {FN_P_PROJ_L3_C, 2, {2, 3, 0}}};
    static const PartDef A_L1[2] = {{FN_P_AFF_L1_A, 3, {0, 2, 3}}, {FN_P_AFF_L1_B, 3, {1, 2, 3}}};
    static const PartDef A_L3[3] = {{FN_P_AFF_L3_A, 2, {0, 2, 0}}, {FN_P_AFF_L3_B, 2, {1, 2, 0}}, {FN_P_AFF_L3_C, 1, {2, 0, 0}}};
    static const PartDef LGU[2] = {{FN_P_LOGUP_A, 3, {0, 1, 3}}, {FN_P_LOGUP_B, 2, {1, 2, 0}}};
    SegPlan r = sp;
    r.nseg = 0;
    bool any = false;
    for (int s = 0; s < sp.nseg; s++) {
        const Seg& g = sp.seg[s];
        const PartDef* pd = nullptr;
        int np = 0;
        switch (g.prim) {
            case FN_PROJ_L1: pd = P_L1; np = 3; break;
            case FN_PROJ_L2: pd = P_L2; np = 2; break;
            case FN_PROJ_L3: pd = P_L3; np = 3; break;
            case FN_AFF_L1: pd = A_L1; np = 2; break;
            case FN_AFF_L3: pd = A_L3; np = 3; break;
            case FN_LOGUP_LAYER: pd = LGU; np = 2; break;
            default: break;
        }
        if (!pd) {
            if (r.nseg >= GM_MAX_SEGS) return false;
            r.seg[r.nseg++] = g;
            continue;
        }
        for (int k = 0; k < np; k++) {
            if (r.nseg >= GM_MAX_SEGS) return false;
            Seg& o = r.seg[r.nseg++];
            o.prim = (int8_t)pd[k].prim; o.n_in = (int8_t)pd[k].n_in; o.n_out = 0; o.out0 = g.out0;
            for (int q = 0; q < 6; q++) o.in[q] = q < pd[k].n_in ? g.in[pd[k].in[q]] : 0;
        }
        any = true;
    }
    if (!any) return false;
    *out = r;
    return true;
}

struct ColPtrs {
    const Fr* p[GM_MAX_COLS];
};
struct ColPtrsMut {
    Fr* p[GM_MAX_COLS];
};

}  // namespace gm
