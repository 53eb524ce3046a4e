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

struct ColPtrs {
    const Fr* p[GM_MAX_COLS];
};
struct ColPtrsMut {
    Fr* p[GM_MAX_COLS];
};

}  // namespace gm
