// Twisted-Edwards addition as three degree-2 layers + the AlgFn combinators, as device/host inlines.
//
// Mirrors (formulas and output order are the contract, they define the layer polynomials):
//   /root/reference/src/cleanup/utils/twisted_edwards_ops.rs:10-80
//   /root/reference/src/cleanup/utils/algfn.rs:129-292   (Id / Repeated / Stacked / BitCheck)
//   /root/reference/src/gkr_msm_simple.rs:82-84          (pt_bit_choice, gen-1)
// A function is selected by a small descriptor (GmFn) instead of a Rust generic.
#pragma once
#include "fr.hip.h"

namespace gm {

// Primitive function ids (values are part of the C ABI, see include/gkrmsm.h)
enum : int {
    FN_AFF_L1 = 1,   // affine_twisted_edwards_add_l1   4 -> 3
    FN_AFF_L2 = 2,   // affine_twisted_edwards_add_l2   3 -> 3
    FN_AFF_L3 = 3,   // affine_twisted_edwards_add_l3   3 -> 3
    FN_PROJ_L1 = 4,  // twisted_edwards_add_l1          6 -> 4
    FN_PROJ_L2 = 5,  // twisted_edwards_add_l2          4 -> 4
    FN_PROJ_L3 = 6,  // twisted_edwards_add_l3          4 -> 3
    FN_TRI_L1 = 7,   // triangle_twisted_edwards_add_l1 12 -> 12
    FN_ID = 8,       // IdAlgFn(1)                      1 -> 1
    FN_BITCHECK = 9, // BitCheckFn                      1 -> 1
    FN_PT_BIT_CHOICE = 10,  // gen-1 pt_bit_choice      3 -> 2
    FN_ADD_INVERSES = 11,   // AddInversesFn (pushforward.rs:255-281)      (a, b) -> (a + b, a b)
    FN_LOGUP_LAYER = 12,    // LogupLayerFn (logup_mainphase.rs:30-61)     (a, b, c, d) -> (a d + b c, b d)
};

GM_HD int prim_n_ins(int id) {
    switch (id) {
        case FN_AFF_L1: return 4; case FN_AFF_L2: return 3; case FN_AFF_L3: return 3;
        case FN_PROJ_L1: return 6; case FN_PROJ_L2: return 4; case FN_PROJ_L3: return 4;
        case FN_TRI_L1: return 12; case FN_ID: return 1; case FN_BITCHECK: return 1;
        case FN_PT_BIT_CHOICE: return 3; case FN_ADD_INVERSES: return 2; case FN_LOGUP_LAYER: return 4; default: return 0;
    }
}
GM_HD int prim_n_outs(int id) {
    switch (id) {
        case FN_AFF_L1: return 3; case FN_AFF_L2: return 3; case FN_AFF_L3: return 3;
        case FN_PROJ_L1: return 4; case FN_PROJ_L2: return 4; case FN_PROJ_L3: return 3;
        case FN_TRI_L1: return 12; case FN_ID: return 1; case FN_BITCHECK: return 1;
        case FN_PT_BIT_CHOICE: return 2; case FN_ADD_INVERSES: return 2; case FN_LOGUP_LAYER: return 2; default: return 0;
    }
}
GM_HD int prim_deg(int id) { return (id == FN_ID) ? 1 : 2; }

// ---- layer formulas -------------------------------------------------------------------------
GM_HD void aff_l1(const Fr* a, Fr* o) {  // (x1,y1,x2,y2) -> (x1y2, x2y1, y1y2 - a x1x2)
    o[0] = fr_mul(a[0], a[3]);
    o[1] = fr_mul(a[2], a[1]);
    o[2] = fr_sub(fr_mul(a[1], a[3]), fr_mul_by_a(fr_mul(a[0], a[2])));
}
GM_HD void aff_l2(const Fr* a, Fr* o) {  // (x1y2, x2y1, t) -> (sum, t, prod)
    Fr s = fr_add(a[0], a[1]);
    Fr p = fr_mul(a[0], a[1]);
    Fr t = a[1 + 1];
    o[0] = s; o[1] = t; o[2] = p;
}
GM_HD void aff_l3(const Fr* a, Fr* o) {  // (x, y, xy) -> ((1-dxy)x, (1+dxy)y, (1-dxy)(1+dxy))
    Fr dxy = fr_mul_by_d(a[2]);
    Fr m = fr_sub(fr_one(), dxy);
    Fr p = fr_add(fr_one(), dxy);
    Fr x = a[0], y = a[1];
    o[0] = fr_mul(m, x); o[1] = fr_mul(p, y); o[2] = fr_mul(m, p);
}
GM_HD void proj_l1(const Fr* a, Fr* o) {  // (x1,y1,z1,x2,y2,z2) -> (x1y2, x2y1, y1y2 - a x1x2, z1z2)
    Fr r0 = fr_mul(a[0], a[4]);
    Fr r1 = fr_mul(a[3], a[1]);
    Fr r2 = fr_sub(fr_mul(a[1], a[4]), fr_mul_by_a(fr_mul(a[0], a[3])));
    Fr r3 = fr_mul(a[2], a[5]);
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
GM_HD void proj_l2(const Fr* a, Fr* o) {  // (x1y2, x2y1, t, zz) -> ((x1y2+x2y1)zz, t zz, zz^2, x1y2 x2y1)
    Fr r0 = fr_mul(fr_add(a[0], a[1]), a[3]);
    Fr r1 = fr_mul(a[2], a[3]);
    Fr r2 = fr_sqr(a[3]);
    Fr r3 = fr_mul(a[0], a[1]);
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
GM_HD void proj_l3(const Fr* a, Fr* o) {  // (x, y, z2, xy) -> ((z2-dxy)x, (z2+dxy)y, (z2-dxy)(z2+dxy))
    Fr dxy = fr_mul_by_d(a[3]);
    Fr m = fr_sub(a[2], dxy);
    Fr p = fr_add(a[2], dxy);
    Fr x = a[0], y = a[1];
    o[0] = fr_mul(m, x); o[1] = fr_mul(p, y); o[2] = fr_mul(m, p);
}
GM_HD void tri_l1(const Fr* pts, Fr* o) {  // (a,b,c,d) -> l1(a,c) | l1(b,d) | l1(c,d)
    Fr in[6];
#pragma unroll
    for (int i = 0; i < 3; i++) { in[i] = pts[i]; in[3 + i] = pts[6 + i]; }
    proj_l1(in, o);
#pragma unroll
    for (int i = 0; i < 3; i++) { in[i] = pts[3 + i]; in[3 + i] = pts[9 + i]; }
    proj_l1(in, o + 4);
#pragma unroll
    for (int i = 0; i < 3; i++) { in[i] = pts[6 + i]; in[3 + i] = pts[9 + i]; }
    proj_l1(in, o + 8);
}

GM_HD void prim_exec(int id, const Fr* a, Fr* o) {
    switch (id) {
        case FN_AFF_L1: aff_l1(a, o); break;
        case FN_AFF_L2: aff_l2(a, o); break;
        case FN_AFF_L3: aff_l3(a, o); break;
        case FN_PROJ_L1: proj_l1(a, o); break;
        case FN_PROJ_L2: proj_l2(a, o); break;
        case FN_PROJ_L3: proj_l3(a, o); break;
        case FN_TRI_L1: tri_l1(a, o); break;
        case FN_ID: o[0] = a[0]; break;
        case FN_BITCHECK: o[0] = fr_sub(fr_sqr(a[0]), a[0]); break;
        case FN_PT_BIT_CHOICE: {  // (b,x,y) -> (b x, b (y-1) + 1)
            Fr bx = fr_mul(a[0], a[1]);
            Fr by = fr_add(fr_mul(a[0], fr_sub(a[2], fr_one())), fr_one());
            o[0] = bx; o[1] = by;
        } break;
        case FN_ADD_INVERSES: {
            Fr s0 = fr_add(a[0], a[1]);
            Fr p0 = fr_mul(a[0], a[1]);
            o[0] = s0; o[1] = p0;
        } break;
        case FN_LOGUP_LAYER: {
            Fr n0 = fr_add(fr_mul(a[0], a[3]), fr_mul(a[1], a[2]));
            Fr d0 = fr_mul(a[1], a[3]);
            o[0] = n0; o[1] = d0;
        } break;
        default: break;
    }
}

// A composite AlgFn = up to GM_FN_MAX_SEG segments (prim id, repeat count), laid out left to right:
//   StackedAlgFn(f1, RepeatedAlgFn(f2, n)) == segs {(f1,1),(f2,n)}
#define GM_FN_MAX_SEG 4
struct GmFn {
    int nseg;
    int prim[GM_FN_MAX_SEG];
    int count[GM_FN_MAX_SEG];
};

GM_HD int fn_n_ins(const GmFn& f) {
    int n = 0;
    for (int s = 0; s < f.nseg; s++) n += prim_n_ins(f.prim[s]) * f.count[s];
    return n;
}
GM_HD int fn_n_outs(const GmFn& f) {
    int n = 0;
    for (int s = 0; s < f.nseg; s++) n += prim_n_outs(f.prim[s]) * f.count[s];
    return n;
}
GM_HD int fn_deg(const GmFn& f) {
    int d = 0;
    for (int s = 0; s < f.nseg; s++) {
        if (f.count[s] > 0) { int pd = prim_deg(f.prim[s]); d = pd > d ? pd : d; }
    }
    return d;
}

}  // namespace gm
