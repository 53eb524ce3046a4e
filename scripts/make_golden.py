#!/usr/bin/env python3
"""Generates tests/golden/*.json from the Python big-int restatement (oracle/pyref).

The reference (Rust, /root/reference) cannot be built or run here (no cargo/rustc; un-vendored arkworks/liblasso
crates) and holds no byte-level vectors of its own, so these fixtures pin OUR canonical outputs: field elements are
canonical integers mod p as hex (non-Montgomery), the form `serialize_compressed` writes little-endian
(/root/reference/src/cleanup/proof_transcript.rs:52-57).  The one reference-sourced constant, COEFF_D
(/root/reference/src/utils.rs:35), is included as the Montgomery limbs the reference spells out.

Usage: python scripts/make_golden.py   (deterministic; rewrites tests/golden/)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from pyref import algfn as A  # noqa: E402
from pyref import field as F  # noqa: E402
from pyref import gkr as G  # noqa: E402
from pyref import polys as PL  # noqa: E402
from pyref import sumcheck as SC  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def hx(v):
    return "%064x" % v


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")


def field_fixture():
    rng = F.SplitMix64(0x474B524D534D)
    a = [0, 1, F.P - 1, 2, F.TE_D] + [rng.next_fr() for _ in range(11)]
    b = [0, F.P - 1, F.P - 1, F.P - 2, 5] + [rng.next_fr() for _ in range(11)]
    g = (18886178867200960497001835917649091219057080094937609519140440539760939937304,
         19188667384257783945677642223292697773471335439753913231509108946878080696678)
    return {
        "modulus": hx(F.P), "montgomery_R": hx(F.R), "te_a": hx(F.TE_A), "te_d": hx(F.TE_D),
        "coeff_d_montgomery_limbs_u64": [str(x) for x in F.COEFF_D_MONT_LIMBS],
        "bandersnatch_order": hx(F.BS_ORDER), "generator": [hx(g[0]), hx(g[1])],
        "a": [hx(x) for x in a], "b": [hx(x) for x in b],
        "add": [hx((x + y) % F.P) for x, y in zip(a, b)], "sub": [hx((x - y) % F.P) for x, y in zip(a, b)],
        "mul": [hx(x * y % F.P) for x, y in zip(a, b)], "neg": [hx((-x) % F.P) for x in a],
        "inv": [hx(F.inv(x)) if x else hx(0) for x in a],
        "mul_by_a": [hx(F.mul_by_a(x)) for x in a], "mul_by_d": [hx(F.mul_by_d(x)) for x in a],
        "to_montgomery": [hx(F.to_mont(x)) for x in a],
    }


def layer_fixture():
    rng = F.SplitMix64(77)
    fns = {"affine_twisted_edwards_add_l1": A.AFF_L1, "affine_twisted_edwards_add_l2": A.AFF_L2,
           "affine_twisted_edwards_add_l3": A.AFF_L3, "twisted_edwards_add_l1": A.PROJ_L1,
           "twisted_edwards_add_l2": A.PROJ_L2, "twisted_edwards_add_l3": A.PROJ_L3,
           "triangle_twisted_edwards_add_l1": A.TRI_L1}
    out = {}
    for name, f in fns.items():
        rows = [[rng.next_fr() for _ in range(f.n_ins)] for _ in range(3)]
        out[name] = {"deg": f.deg, "n_ins": f.n_ins, "n_outs": f.n_outs,
                     "in": [[hx(v) for v in r] for r in rows], "out": [[hx(v) for v in f.exec(r)] for r in rows]}
    # group-law anchors (reference Pattern C): P + Q, 2P, P + O in affine
    pts = F.random_points(4, 9)
    out["group_law"] = {"points": [[hx(p[0]), hx(p[1])] for p in pts],
                        "p0_plus_p1": [hx(v) for v in F.te_add_affine(pts[0], pts[1])],
                        "double_p2": [hx(v) for v in F.te_add_affine(pts[2], pts[2])],
                        "p3_times_12345": [hx(v) for v in F.te_mul_affine(pts[3], 12345)]}
    return out


def poly_fixture():
    rng = F.SplitMix64(5)
    v = [rng.next_fr() for _ in range(16)]
    t = rng.next_fr()
    pt = [rng.next_fr() for _ in range(4)]
    m = rng.next_fr()
    return {"v": [hx(x) for x in v], "t": hx(t), "bind": [hx(x) for x in PL.bind_dense(v, t)],
            "point": [hx(x) for x in pt], "multiplier": hx(m),
            "eq_table": [hx(x) for x in PL.eq_poly_sequence_from_multiplier(m, pt)[-1]],
            "evaluate": hx(PL.evaluate_poly(v, pt)), "eq_sum_k5": hx(PL.eq_sum(pt, 5))}


def msm_fixture(x_log, d_log, nbits, seed):
    y_size = (nbits + d_log - 1) // d_log
    y_log = PL.log2_exact(y_size)
    n = 1 << x_log
    pts = F.random_points(n, seed)
    sc = F.random_scalars(n, nbits, seed + 1)
    sc[1] = 0
    sc[2] = sc[3]
    image, digits, counter, wg = G.pippenger_witness(pts, sc, y_size, y_log, d_log, x_log)
    out = G.pippenger_dense_output(wg, y_log, d_log)
    res = G.pippenger_final_point(out, d_log)
    rng = F.SplitMix64(seed + 2)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = G.pippenger_claims(out, r)
    tape = [rng.next_bits(128) for _ in range(600)]
    tr = SC.TapeTranscript(tape)
    fin = G.prove_image_part(tr, y_log, d_log, x_log, claims, wg)
    nrows = y_size << d_log
    return {
        "x_logsize": x_log, "d_logsize": d_log, "nbits": nbits, "y_size": y_size, "y_logsize": y_log,
        "points": [[hx(p[0]), hx(p[1])] for p in pts], "scalars": [hx(s) for s in sc],
        "digits": digits, "counter": counter,
        "bucket_sums": [[hx(v) for v in col[:nrows]] for col in wg.bucket_sums],
        "window_points": [[hx(v) for v in col] for col in out],
        "msm_result": [hx(res[0]), hx(res[1])],
        "claim_point": [hx(v) for v in r], "claim_evs": [hx(v) for v in claims[1]],
        "tape": [hx(v) for v in tape[:tr.pos]],
        "prover_messages": [hx(v) for m in tr.msgs for v in m],
        "final_point": [hx(v) for v in fin[0]], "final_evs": [hx(v) for v in fin[1]],
    }


def proof_fixture(x_log, d_log, nbits, clm, seed):
    """a whole gen-2 proof (PippengerWG::new + Pippenger::prove of the oracle) with everything a verifier needs: shape, claims,
    every transcript scalar and G1 point in write order, the challenges as drawn, the deferred pairing pair and the mock-setup tau"""
    from pyref import g1 as G1
    from pyref import knuckles as KN
    from pyref import pippenger as PP
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    rng = F.SplitMix64(seed)
    pts = F.random_points(n, seed + 1)
    sc = F.random_scalars(n, nbits, seed + 2)
    nv = x_log + clm
    tau, k = rng.next_fr(), 2
    basis, cur = [], G1.GEN
    for _ in range((2 << nv) - 1):
        basis.append(cur)
        cur = G1.mul(cur, tau)
    st = PP.pippenger_wg(pts, sc, y_size, y_log, d_log, x_log, clm, basis)
    out = G.pippenger_dense_output(st["wg"], y_log, d_log)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = G.pippenger_claims(out, r)
    tape = [rng.next_bits(512) for _ in range(4000)]
    tr = PP.Transcript(tape)
    pair = PP.pippenger_prove(tr, st, claims, y_size, y_log, d_log, x_log, clm, basis, KN.setup_inverses(k, nv), k)
    drawn = [t % F.P if i in tr.wide else t & ((1 << 128) - 1) for i, t in enumerate(tape[: tr.pos])]
    h96 = lambda v: "%096x" % v
    pt = lambda p: None if p is None else [h96(p[0]), h96(p[1])]
    return {"x_logsize": x_log, "d_logsize": d_log, "nbits": nbits, "y_size": y_size, "y_logsize": y_log,
            "commitment_log_multiplicity": clm, "k": k, "tau": hx(tau),
            "points_xy": [[hx(p[0]), hx(p[1])] for p in pts], "scalars_in": [hx(v) for v in sc],
            "claim_point": [hx(v) for v in claims[0]], "claim_evs": [hx(v) for v in claims[1]],
            "transcript_scalars": [hx(v) for m in tr.msgs for v in m], "transcript_points": [pt(p) for p in tr.points],
            "challenges": [hx(v) for v in drawn], "pair": [pt(pair[0]), pt(pair[1])]}


def main():
    os.makedirs(OUT, exist_ok=True)
    dump("proof_x3_d2_n8_clm1.json", proof_fixture(3, 2, 8, 1, 4242))
    dump("field.json", field_fixture())
    dump("layers.json", layer_fixture())
    dump("poly.json", poly_fixture())
    dump("msm_x4_d2_n12.json", msm_fixture(4, 2, 12, 11))
    dump("msm_x5_d3_n24.json", msm_fixture(5, 3, 24, 21))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
