"""CPU tests: the committed golden fixtures (tests/golden, made by scripts/make_golden.py from the Python big-int
restatement) against the C oracle and the host-side code of the product library."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_ffi as O
from gkr_msm_amd import codec, ffi

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def ints(xs):
    return [int(x, 16) for x in xs]


def test_field_fixture_oracle_and_product_host():
    fx = load("field.json")
    a, b = ints(fx["a"]), ints(fx["b"])
    A_, B_ = codec.to_mont_limbs(a), codec.to_mont_limbs(b)
    # reference KAT: COEFF_D limbs (src/utils.rs:35) are d in Montgomery form
    assert codec.to_mont_limbs([int(fx["te_d"], 16)])[0].tolist() == [int(x) for x in fx["coeff_d_montgomery_limbs_u64"]]
    for op, key in [(0, "add"), (1, "sub"), (2, "mul"), (3, "neg"), (7, "mul_by_a"), (8, "mul_by_d")]:
        o = np.zeros_like(A_)
        O.lib().or_fr_batch(op, A_.ctypes.data, B_.ctypes.data, o.ctypes.data, len(a))
        assert codec.from_mont_limbs(o) == ints(fx[key]), "oracle " + key
        o2 = np.zeros_like(A_)
        ffi.check(ffi.lib().gm_fr_host(op, A_.ctypes.data, B_.ctypes.data, o2.ctypes.data, len(a)))
        assert codec.from_mont_limbs(o2) == ints(fx[key]), "product host " + key
    nz = [i for i, x in enumerate(a) if x]
    o = np.zeros_like(A_)
    O.lib().or_fr_batch(4, A_.ctypes.data, None, o.ctypes.data, len(a))
    got = codec.from_mont_limbs(o)
    assert [got[i] for i in nz] == [ints(fx["inv"])[i] for i in nz]
    assert codec.limbs_to_ints(A_) == ints(fx["to_montgomery"])


PRIMS = {"affine_twisted_edwards_add_l1": 1, "affine_twisted_edwards_add_l2": 2, "affine_twisted_edwards_add_l3": 3,
         "twisted_edwards_add_l1": 4, "twisted_edwards_add_l2": 5, "twisted_edwards_add_l3": 6,
         "triangle_twisted_edwards_add_l1": 7}


@pytest.mark.parametrize("name", sorted(PRIMS))
def test_layer_fixture(name):
    fx = load("layers.json")[name]
    f_or, f_gm = O.make_fn((PRIMS[name], 1)), ffi.make_fn((PRIMS[name], 1))
    assert O.lib().or_fn_n_ins(C.byref(f_or)) == fx["n_ins"] and O.lib().or_fn_n_outs(C.byref(f_or)) == fx["n_outs"]
    for row, exp in zip(fx["in"], fx["out"]):
        i = codec.to_mont_limbs(ints(row))
        o = np.zeros((fx["n_outs"], 4), dtype=np.uint64)
        O.lib().or_fn_exec(C.byref(f_or), i.ctypes.data, o.ctypes.data)
        assert codec.from_mont_limbs(o) == ints(exp)
        o2 = np.zeros((fx["n_outs"], 4), dtype=np.uint64)
        ffi.check(ffi.lib().gm_fn_host(C.byref(f_gm), i.ctypes.data, o2.ctypes.data, 1))
        assert codec.from_mont_limbs(o2) == ints(exp)


def test_poly_fixture():
    fx = load("poly.json")
    v, t, pt = ints(fx["v"]), int(fx["t"], 16), ints(fx["point"])
    V, T = codec.to_mont_limbs(v), codec.to_mont_limbs([t])
    o = np.zeros((len(v) // 2, 4), dtype=np.uint64)
    O.lib().or_dense_bind(V.ctypes.data, len(v), T.ctypes.data, o.ctypes.data)
    assert codec.from_mont_limbs(o) == ints(fx["bind"])
    e = np.zeros((1 << len(pt), 4), dtype=np.uint64)
    M, PT = codec.to_mont_limbs([int(fx["multiplier"], 16)]), codec.to_mont_limbs(pt)
    O.lib().or_eq_table(M.ctypes.data, PT.ctypes.data, len(pt), e.ctypes.data)
    assert codec.from_mont_limbs(e) == ints(fx["eq_table"])


@pytest.mark.parametrize("name", ["msm_x4_d2_n12.json", "msm_x5_d3_n24.json"])
def test_msm_and_prover_fixture(name):
    fx = load(name)
    x_log, d_log, y_size, y_log = fx["x_logsize"], fx["d_logsize"], fx["y_size"], fx["y_logsize"]
    pts = codec.points_to_mont([(int(p[0], 16), int(p[1], 16)) for p in fx["points"]])
    sc = codec.ints_to_limbs(ints(fx["scalars"]))
    r = O.msm(pts, sc, x_log, d_log, y_size, threads=2)
    assert r["digits"].tolist() == fx["digits"] and r["counter"].tolist() == fx["counter"]
    for c, k in enumerate(("bx", "by", "bz")):
        assert codec.from_mont_limbs(r[k]) == ints(fx["bucket_sums"][c])
    for c in range(3 * (d_log + 1)):
        assert codec.from_mont_limbs(r["window_cols"][c]) == ints(fx["window_points"][c])[:y_size]
    assert codec.from_mont_limbs(O.msm_combine(r["window_cols"], d_log)) == ints(fx["msm_result"])
    # product host glue: final recombination
    from gkr_msm_amd import harness
    assert list(harness.combine_host(r["window_cols"], d_log)) == ints(fx["msm_result"])
    # prover
    w = O.PipWitness(pts, sc, x_log, d_log, y_size, y_log, 2)
    res = w.prove_image_part(codec.to_mont_limbs(ints(fx["claim_point"])), codec.to_mont_limbs(ints(fx["claim_evs"])),
                             codec.ints_to_limbs(ints(fx["tape"])))
    assert res["tape_used"] == len(fx["tape"])
    assert codec.from_mont_limbs(res["msgs"]) == ints(fx["prover_messages"])
    assert codec.from_mont_limbs(res["point"]) == ints(fx["final_point"])
    assert codec.from_mont_limbs(res["evs"]) == ints(fx["final_evs"])
