"""GPU twins of tests/test_ref_kats_cpu.py: every reference #[test] replayed there on the oracle is replayed here through the
C ABI on the HIP path, with the same literal inputs (tests/golden/ref_kats.json), and compared with the oracle bit for bit.
Test names are the reference's."""
import ctypes as C

import numpy as np
import pytest

from gkr_msm_amd import codec, ffi, harness as H
from pyref import copoly as CP
from pyref import field as F
from pyref import fragmented as FR
from pyref import g1 as G
from pyref import gkr as GK
from pyref import knuckles as K
from pyref import polys as PL
from pyref.algfn import AFF_L1, AFF_L2, AFF_L3, PROJ_L1, PROJ_L2, PROJ_L3, IdAlgFn
from pyref.sumcheck import TapeTranscript
from test_ref_kats_cpu import (KATS, check_triangle_result, frags, rand_points_affine, rand_shape_by_frag_spec, triangle_inputs,
                               tup)

pytestmark = pytest.mark.gpu


def shape_tuples(shape):
    return [tuple(t) for t in tup(shape)]


# ------------------------------------------------------------------ fragmented.rs
def test_split_poly():
    """fragmented.rs:974-1062 on the device: FragmentedPoly::split of the literal polynomial"""
    k = KATS["split_poly"]
    fr_ = [tuple(f) for f in k["fragments"]]
    p = H.FragPoly.from_host(fr_, k["d"], k["c"])
    assert p.to_dense() == k["v"]
    l, r = p.split()
    assert l.to_dense() == k["v"][0::2]
    assert r.to_dense() == k["v"][1::2]
    ol, orr = FR.FragmentedPoly(k["d"], k["c"], FR.Shape(frags(k["fragments"]), k["num_consts"])).split()
    assert l.frags == shape_tuples(ol.shape) and l.data() == ol.data and l.consts() == ol.consts
    assert r.data() == orr.data and r.consts() == orr.consts


def test_split_shape():
    """fragmented.rs:1064-1164: a polynomial over the literal shape, split twice on the device"""
    k = KATS["split_shape"]
    shape = FR.Shape(frags(k["fragments"]), k["num_consts"])
    rng = F.SplitMix64(3)
    data = [rng.next_fr() for _ in range(shape.data_len)]
    consts = [rng.next_fr() for _ in range(k["num_consts"])]
    op = FR.FragmentedPoly(data, consts, shape, mod=F.P)
    gp = H.FragPoly.from_host([tuple(f) for f in k["fragments"]], data, consts)
    ol, orr = op.split()
    gl, gr = gp.split()
    assert [list(f) for f in gl.frags] == k["expected_split"]
    assert (gl.data(), gl.consts(), gr.data(), gr.consts()) == (ol.data, ol.consts, orr.data, orr.consts)
    oll, olr = ol.split()
    gll, glr = gl.split()
    assert [list(f) for f in gll.frags] == k["expected_split_split"]
    assert (gll.data(), glr.data()) == (oll.data, olr.data)


def test_split_rand_poly_and_bind_rand_poly_and_evaluate():
    """fragmented.rs:926-972, 1266-1282: random shapes; split, bind and evaluate (= bind down to one cell) on the device"""
    rng = F.SplitMix64(4321)
    for it in range(40):
        shape = rand_shape_by_frag_spec(rng, 10, 10, 2)
        if len(shape) < 2:
            continue
        data = [rng.next_fr() for _ in range(shape.data_len)]
        consts = [rng.next_fr() for _ in range(2)]
        op = FR.FragmentedPoly(data, consts, shape, mod=F.P)
        gp = H.FragPoly.from_host(shape_tuples(shape), data, consts)
        v = op.into_vec()
        assert gp.to_dense() == v
        gl, gr = gp.split()
        assert gl.to_dense() == v[0::2] and gr.to_dense() == v[1::2]
        pt = [rng.next_fr() for _ in range(op.num_vars())]
        cur_o, cur_g = op, gp
        for t in reversed(pt):
            cur_o, cur_g = cur_o.bind(t), cur_g.bind(t)
            assert cur_g.frags == shape_tuples(cur_o.shape)
            assert cur_g.data() == cur_o.data and cur_g.consts() == cur_o.consts
        assert cur_g.to_dense() == [PL.evaluate_poly(v, pt)]


def test_eq_materialize_with_shape():
    """copoly.rs:492-567 (EqPoly::materialize_eq_with_shape) on the literal KAT shapes and random ones"""
    rng = F.SplitMix64(8)
    shapes = [FR.Shape(frags(KATS["split_poly"]["fragments"]), 2), FR.Shape(frags(KATS["split_shape"]["fragments"]), 4)]
    shapes += [rand_shape_by_frag_spec(rng, 10, 10, 3) for _ in range(10)]
    for shape in shapes:
        n = len(shape)
        if n < 2:
            continue
        nv = n.bit_length() - 1
        point = [rng.next_fr() for _ in range(nv)]
        mult = rng.next_fr()
        want = CP.EqPoly(point, mult).materialize_eq_with_shape(shape)
        got = H.frag_eq_materialize(shape_tuples(shape), shape.num_consts, mult, point)
        assert got == (want[0], want[1])


# ------------------------------------------------------------------ kzg.rs
def test_quotient():
    """kzg.rs:165-172: [1,3,3,7,2,0,2,4] divided by (X - 322), checked at 500"""
    k = KATS["quotient"]
    poly, pt, x = k["poly"], k["pt"], k["check_at"]
    d_poly = H.to_dev(codec.to_mont_limbs(poly))
    d_q = H.dev_empty(4 * (len(poly) - 1))

    def div(d_p, n, at, d_out):
        ev = np.zeros((1, 4), dtype=np.uint64)
        at_arg = H.fr_arg([at])
        ffi.check(ffi.lib().gm_kzg_div_by_linear(H._p(d_p), n, at_arg.ctypes.data, H._p(d_out) if d_out is not None else None,
                                                 ev.ctypes.data, H.cur_stream()))
        return codec.from_mont_limbs(ev)[0]
    rem = div(d_poly, len(poly), pt, d_q)
    q = codec.from_mont_limbs(H.to_host(d_q).reshape(-1, 4))
    oq, orem = K.div_by_linear(poly, pt)
    assert (q, rem) == (oq, orem)
    assert rem == K.ev(poly, pt)                                           # assert!(ev(&poly, pt) == remainder)
    ev_p, ev_q = div(d_poly, len(poly), x, None), div(d_q, len(q), x, None)
    assert ev_p == (ev_q * (x - pt) + rem) % F.P                           # the reference's second assert
    # a long polynomial (several 64-coefficient chunks, ragged tail)
    rng = F.SplitMix64(9)
    big = [rng.next_fr() for _ in range(1000)]
    at = rng.next_fr()
    d_big, d_bq = H.to_dev(codec.to_mont_limbs(big)), H.dev_empty(4 * 999)
    assert div(d_big, 1000, at, d_bq) == K.ev(big, at)
    assert codec.from_mont_limbs(H.to_host(d_bq).reshape(-1, 4)) == K.div_by_linear(big, at)[0]


# ------------------------------------------------------------------ binary_msm.rs, pullback.rs
@pytest.mark.parametrize("name", ["bin_msm", "bin_msm_gamma_3"])
def test_bin_msm(name):
    """binary_msm.rs:63-95 at the literal size (100 bases, gamma 8 / 3)"""
    k = KATS[name]
    num, gamma = k["num"], k["gamma"]
    rng = F.SplitMix64(20 + gamma)
    bits = [bool(rng.next() & 1) for _ in range(num)]
    bases = G.random_points(num, 30 + gamma)
    tables = H.g1_prepare_bases(H.g1_aff_dev(bases), num, gamma)
    got = H.g1_binary_msm(G.prepare_coefs(bits, gamma), tables, gamma)
    exp = None
    for b, p in zip(bits, bases):
        if b:
            exp = G.add(exp, p)
    assert got == exp
    assert got == G.binary_msm(G.prepare_coefs(bits, gamma), G.prepare_bases(bases, gamma))


def test_bucketed_msm():
    """pullback.rs:85-105 at the literal size (1024 bases, 64 image values): bucketed == plain MSM of the pulled-back values"""
    k = KATS["test_bucketed_msm"]
    rng = F.SplitMix64(31)
    mapping = [rng.next() % k["image_size"] for _ in range(k["num_bases"])]
    image = [rng.next_fr() for _ in range(k["image_size"])]
    bases = G.random_points(k["num_bases"], 32)
    d_bases = H.g1_aff_dev(bases)
    got = H.g1_pullback_msm(d_bases, mapping, image)
    assert got == G.pullback_bucketed_msm(mapping, image, bases)
    values = G.pullback_values(mapping, image)
    assert got == H.g1_msm(d_bases, H.to_dev(codec.to_mont_limbs(values)), len(values), mont=True)


# ------------------------------------------------------------------ triangle_add.rs
def test_triangle_witness_gen():
    """triangle_add.rs:277-355 at the literal size (num_vars 12, HI(4)): the device witness == the oracle's, and
    sum_i i P_i == sum_{i >= 1} 2^(i-1) result_i for each of the 16 groups"""
    k = KATS["triangle_witness_gen"]
    nv, hi = k["num_vars"], k["split_hi"]
    pts, s2 = triangle_inputs(nv, hi, 900)
    w = H.GkrWitness.triangle(H.cols_to_dev(s2), nv - 2, hi)
    last, out_vars = w.output()
    assert out_vars == hi and len(last) == 3 * (nv - 2 - hi + 3)
    adv = GK.triangle_witness_build(s2, nv - 2, PL.HI(hi))
    assert last == GK.triangle_last_step(adv[-1][1], nv - 2 - hi)
    check_triangle_result(pts, last, nv, hi)


def test_triangle_prove_and_verify():
    """triangle_add.rs:357-393 (num_vars 8, HI(2)): every prover message == the oracle's; final claims == inputs at the point"""
    k = KATS["triangle_prove_and_verify"]
    nv, hi = k["num_vars"], k["split_hi"]
    _, s2 = triangle_inputs(nv, hi, 910)
    w = H.GkrWitness.triangle(H.cols_to_dev(s2), nv - 2, hi)
    out, _ = w.output()
    rng = F.SplitMix64(911)
    point = [rng.next_fr() for _ in range(hi)]
    evs = [PL.evaluate_poly(o, point) for o in out]
    tape = [rng.next_bits(128) for _ in range(400)]
    res = w.prove(point, evs, tape)
    adv = GK.triangle_witness_build(s2, nv - 2, PL.HI(hi))
    tr = TapeTranscript(tape)
    fin = GK.simple_gkr_prove(tr, GK.triangle_protocol_layers(nv - 2, PL.HI(hi)), adv, (point, evs))
    assert res["msgs"] == [v for m in tr.msgs for v in m] and res["tape_used"] == tr.pos
    assert (res["point"], res["evs"]) == (fin[0], fin[1])
    assert res["evs"] == [PL.evaluate_poly(c, res["point"]) for c in s2]


# ------------------------------------------------------------------ bintree_add.rs
def vv_to_dev(polys):
    return H.VV.from_host([p.data for p in polys], [p.row_pad for p in polys], [p.col_pad for p in polys], polys[0].row_logsize,
                          polys[0].col_logsize)


@pytest.mark.parametrize("num_adds,row_logsize,col_logsize", [tuple(c) for c in KATS["bintree_prove_and_verify"]["cases"]])
def test_bintree_prove_and_verify(num_adds, row_logsize, col_logsize):
    """bintree_add.rs:401-458, cases (5,4,2) and (5,2,4), no bit check"""
    rng = F.SplitMix64(920 + row_logsize)
    points = rand_points_affine(rng, row_logsize, col_logsize, 921)
    inputs = PL.vecvec_map_split(points, IdAlgFn(2), PL.LO(0), 2)
    g_in = vv_to_dev(points).map_split(ffi.make_fn((ffi.FN_ID, 2)), 2)
    w = H.GkrWitness.bintree(g_in, num_adds, False)
    out, out_vars = w.output()
    nv = row_logsize + col_logsize
    adv = GK.bintree_witness_build(("VV", inputs), row_logsize, num_adds, False)
    last = GK.bintree_last_step(adv[-1], num_adds - 1)
    dense_out = [p.to_dense() for p in last[1]] if last[0] == "VV" else last[1]
    assert out_vars == nv - num_adds and out == dense_out
    point = [rng.next_fr() for _ in range(nv - num_adds)]
    evs = [PL.evaluate_poly(o, point) for o in dense_out]
    tape = [rng.next_bits(128) for _ in range(600)]
    res = w.prove(point, evs, tape)
    tr = TapeTranscript(tape)
    fin = GK.simple_gkr_prove(tr, GK.bintree_protocol_layers(nv, num_adds, row_logsize, False), adv, (point, evs))
    assert res["msgs"] == [v for m in tr.msgs for v in m] and res["tape_used"] == tr.pos
    assert (res["point"], res["evs"]) == (fin[0], fin[1])
    assert res["evs"] == [PL.evaluate_poly(p.to_dense(), res["point"]) for p in inputs]


def test_bintree_witness_gen():
    """bintree_add.rs:460-505 (row 6, col 2, 5 adds): every output is the sum of its group of 2^5 input points"""
    k = KATS["bintree_witness_gen"]
    row, col, adds = k["row_logsize"], k["col_logsize"], k["num_adds"]
    rng = F.SplitMix64(930)
    points = rand_points_affine(rng, row, col, 931)
    g_in = vv_to_dev(points).map_split(ffi.make_fn((ffi.FN_ID, 2)), 2)
    w = H.GkrWitness.bintree(g_in, adds, False)
    out, out_vars = w.output()
    assert out_vars == row + col - adds
    dx, dy = points[0].to_dense(), points[1].to_dense()
    group = 1 << adds
    for idx in range(len(out[0])):
        acc = (0, 1)
        for c in range(group):
            p = (dx[idx * group + c], dy[idx * group + c])
            if p != (0, 0):  # the (0, 0) padding of absent cells is not a curve point; its images stay (0, 0, 0) below
                acc = F.te_add_affine(acc, p)
        if out[2][idx] != 0:
            assert F.proj_to_affine(out[0][idx], out[1][idx], out[2][idx]) == acc
    adv = GK.bintree_witness_build(("VV", PL.vecvec_map_split(points, IdAlgFn(2), PL.LO(0), 2)), row, adds, False)
    last = GK.bintree_last_step(adv[-1], adds - 1)
    assert out == ([p.to_dense() for p in last[1]] if last[0] == "VV" else last[1])


@pytest.mark.parametrize("col_logsize", KATS["check_point_addition"]["col_logsize"])
@pytest.mark.parametrize("row_logsize", KATS["check_point_addition"]["row_logsize"])
def test_check_affine_point_addition(col_logsize, row_logsize):
    """bintree_add.rs:508-562: l3(l2(l1(split inputs))) == Affine + Affine on every pair of stored points"""
    rng = F.SplitMix64(940 + 8 * col_logsize + row_logsize)
    points = rand_points_affine(rng, row_logsize, col_logsize, 941)
    g = vv_to_dev(points).map_split(ffi.make_fn((ffi.FN_ID, 2)), 2)
    g = g.map(ffi.make_fn((ffi.FN_AFF_L1, 1))).map(ffi.make_fn((ffi.FN_AFF_L2, 1)))
    o = PL.vecvec_map(PL.vecvec_map(PL.vecvec_map_split(points, IdAlgFn(2), PL.LO(0), 2), AFF_L1), AFF_L2)
    if row_logsize == 2:   # advice_map_split switches to dense when layer_idx + 2 == row_logsize (bintree_add.rs:188-195)
        got = H.cols_to_host(g.map_split_to_dense(ffi.make_fn((ffi.FN_AFF_L3, 1)), 3, 3))
        want = PL.vecvec_map_split_to_dense(o, AFF_L3, PL.LO(0), 3)
    else:
        got = g.map_split(ffi.make_fn((ffi.FN_AFF_L3, 1)), 3).to_dense()
        want = [p.to_dense() for p in PL.vecvec_map_split(o, AFF_L3, PL.LO(0), 3)]
    assert got == want
    dense_pts = [(x, y) for x, y in zip(points[0].to_dense(), points[1].to_dense()) if F.te_on_curve(x, y)]
    sums = [F.te_add_affine(dense_pts[i], dense_pts[i + 1]) for i in range(0, len(dense_pts) - 1, 2)]
    dx, dy = points[0].to_dense(), points[1].to_dense()
    # output pair idx lives in bundle idx % 2 at position idx // 2 -- indexed over the DENSE pairs, pads included
    for pair in range(len(dx) // 2):
        a, b = (dx[2 * pair], dy[2 * pair]), (dx[2 * pair + 1], dy[2 * pair + 1])
        if not (F.te_on_curve(*a) and F.te_on_curve(*b)):
            continue
        X, Y, Z = (got[3 * (pair % 2) + c][pair // 2] for c in range(3))
        assert F.proj_to_affine(X, Y, Z) == F.te_add_affine(a, b)
    assert len(sums) >= 1


@pytest.mark.parametrize("col_logsize", KATS["check_point_addition"]["col_logsize"])
@pytest.mark.parametrize("row_logsize", KATS["check_point_addition"]["row_logsize"])
def test_check_projective_point_addition(col_logsize, row_logsize):
    """bintree_add.rs:564-638: the projective layers l1, l2, l3 == Projective + Projective"""
    rng = F.SplitMix64(950 + 8 * col_logsize + row_logsize)
    aff = rand_points_affine(rng, row_logsize, col_logsize, 951)
    zs = [[rng.next_fr() or 1 for _ in r] for r in aff[0].data]
    xs = PL.VecVec([[x * z % F.P for x, z in zip(r, zr)] for r, zr in zip(aff[0].data, zs)], 0, 0, row_logsize, col_logsize)
    ys = PL.VecVec([[y * z % F.P for y, z in zip(r, zr)] for r, zr in zip(aff[1].data, zs)], 1, 1, row_logsize, col_logsize)
    zz = PL.VecVec(zs, 1, 1, row_logsize, col_logsize)
    polys = [xs, ys, zz]
    g = vv_to_dev(polys).map_split(ffi.make_fn((ffi.FN_ID, 3)), 3)
    g = g.map(ffi.make_fn((ffi.FN_PROJ_L1, 1))).map(ffi.make_fn((ffi.FN_PROJ_L2, 1)))
    o = PL.vecvec_map(PL.vecvec_map(PL.vecvec_map_split(polys, IdAlgFn(3), PL.LO(0), 3), PROJ_L1), PROJ_L2)
    if row_logsize == 2:  # called with layer_idx 1, row_logsize + 1 in the reference, i.e. VecVec -> VecVec unless 3 == row_logsize + 1
        got = H.cols_to_host(g.map_split_to_dense(ffi.make_fn((ffi.FN_PROJ_L3, 1)), 3, 3))
        want = PL.vecvec_map_split_to_dense(o, PROJ_L3, PL.LO(0), 3)
    else:
        got = g.map_split(ffi.make_fn((ffi.FN_PROJ_L3, 1)), 3).to_dense()
        want = [p.to_dense() for p in PL.vecvec_map_split(o, PROJ_L3, PL.LO(0), 3)]
    assert got == want
    dx, dy, dz = xs.to_dense(), ys.to_dense(), zz.to_dense()
    checked = 0
    for pair in range(len(dx) // 2):
        ok = True
        ab = []
        for i in (2 * pair, 2 * pair + 1):
            p = F.proj_to_affine(dx[i], dy[i], dz[i]) if dz[i] else None
            ok = ok and p is not None and F.te_on_curve(*p)
            ab.append(p)
        if not ok:
            continue
        X, Y, Z = (got[3 * (pair % 2) + c][pair // 2] for c in range(3))
        assert F.proj_to_affine(X, Y, Z) == F.te_add_affine(*ab)
        checked += 1
    assert checked >= 1
