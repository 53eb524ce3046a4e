"""The reference's own deterministic tests, replayed on the oracle (and on the library's host-only entry points) with the
reference's LITERAL inputs and expected values (tests/golden/ref_kats.json holds the data; each test names the Rust
#[test] it mirrors).  This is what pins the oracle to values the reference itself holds; the GPU twin of every test here is
in tests/test_ref_kats_gpu.py.

Tests of the reference that draw their inputs from ark_std::test_rng() cannot be replayed value for value (no Rust toolchain,
ark-std is not vendored): for those the literal SIZES and the asserted identity are kept and inputs come from SplitMix64."""
import json
import os

import pytest

from gkr_msm_amd import harness as H
from pyref import copoly as CP
from pyref import field as F
from pyref import fragmented as FR
from pyref import g1 as G
from pyref import gkr as GK
from pyref import knuckles as K
from pyref import polys as PL
from pyref.algfn import IdAlgFn, RepeatedAlgFn
from pyref.sumcheck import TapeTranscript

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats.json")))


def frags(lst):
    return [FR.Fragment(m, ln, c, s) for m, ln, c, s in lst]


def tup(shape):
    return [list(f.key()[:2]) + [f.content, f.start] for f in shape.fragments]


# ------------------------------------------------------------------ fragmented.rs
def test_split_poly():
    """fragmented.rs:974-1062"""
    k = KATS["split_poly"]
    shape = FR.Shape(frags(k["fragments"]), k["num_consts"])
    split, perm = shape.full_split()
    assert perm == k["expected_perm"]
    p = FR.FragmentedPoly(k["d"], k["c"], shape)
    assert p.into_vec() == k["v"]
    l, r = p.split()
    assert l.into_vec() == k["v"][0::2]
    assert r.into_vec() == k["v"][1::2]
    # the library's host-side Shape::full_split gives the same fragments and permutation
    got, gperm, dl = H.frag_shape_full_split([tuple(f) for f in k["fragments"]], k["num_consts"])
    assert [list(g) for g in got] == tup(split) and gperm == perm and dl == split.data_len


def test_split_shape():
    """fragmented.rs:1064-1164"""
    k = KATS["split_shape"]
    s = FR.Shape(frags(k["fragments"]), k["num_consts"])
    l, perm = s.full_split()
    assert perm == k["expected_perm"]
    assert l == FR.Shape(frags(k["expected_split"]), k["num_consts"])
    ll = l.split()
    assert ll == FR.Shape(frags(k["expected_split_split"]), k["num_consts"])
    got, gperm, _ = H.frag_shape_full_split([tuple(f) for f in k["fragments"]], k["num_consts"])
    assert [list(g) for g in got] == k["expected_split"] and gperm == k["expected_perm"]
    got2, _, dl2 = H.frag_shape_full_split(got, k["num_consts"])
    assert [list(g) for g in got2] == k["expected_split_split"] and dl2 == 8


def rand_shape_by_frag_spec(rng, frag_size, n_frags, num_consts):
    """Shape::rand_by_frag_spec (fragmented.rs:206-236) over SplitMix64"""
    s = FR.Shape.empty(num_consts)
    start = 0
    for _ in range(n_frags):
        ln = rng.next() % frag_size + 1
        content = FR.DATA if rng.next() % 2 == 0 else FR.CONSTS
        mem = s.data_len if content == FR.DATA else rng.next() % num_consts
        s.add(FR.Fragment(mem, ln, content, start))
        start += ln
    total = 1 << (start - 1).bit_length() if start & (start - 1) else start
    if total == start:  # `(1 << start.log_2()) - start` with liblasso's log_2 = ceil: a zero-length tail is still added
        pass
    s.add(FR.Fragment(rng.next() % num_consts, total - start, FR.CONSTS, start))
    s.assert_correct()
    return s


def test_split_rand_poly_and_bind_rand_poly():
    """fragmented.rs:926-972 (100 random shapes of 10 fragments of <= 10 cells, one constant): split == even/odd of
    into_vec, bind == dense bound_poly_var_bot; and the library's host Shape::full_split agrees on every shape"""
    rng = F.SplitMix64(1234)
    for _ in range(100):
        shape = rand_shape_by_frag_spec(rng, 10, 10, 1)
        if len(shape) < 2:
            continue
        p = FR.FragmentedPoly([rng.next() % 10 for _ in range(shape.data_len)], [rng.next() % 10], shape)
        v = p.into_vec()
        l, r = p.split()
        assert l.into_vec() == v[0::2] and r.into_vec() == v[1::2]
        f = rng.next_fr()
        pm = FR.FragmentedPoly(p.data, p.consts, shape, mod=F.P)
        assert pm.bind(f).into_vec() == PL.bind_dense(v, f)
        got, gperm, dl = H.frag_shape_full_split([f_.key()[:2] + (f_.content, f_.start) for f_ in shape.fragments], 1)
        tgt, perm = shape.full_split()
        assert [list(g) for g in got] == tup(tgt) and gperm == perm and dl == tgt.data_len


def test_non_native_split():
    """fragmented.rs:1166-1264: split_at on shapes [Data] / [Data, Consts], 5 variables"""
    rng = F.SplitMix64(77)
    nv = 5
    for kind in range(3):
        for _ in range(100):
            if kind == 0:
                split_var = rng.next() % nv
                data_len, consts_len = 1 << nv, 0
            elif kind == 1:
                split_var = rng.next() % nv
                sector = 1 << (nv - 1 - split_var)
                data_len = 2 * (rng.next() % (1 << split_var) + 1) * sector
                consts_len = (1 << nv) - data_len
            else:
                split_var, data_len, consts_len = nv - 1, (1 << nv) - 2, 2
            sector = 1 << (nv - 1 - split_var)
            fr_ = [FR.Fragment(0, data_len, FR.DATA, 0)] + ([FR.Fragment(0, consts_len, FR.CONSTS, data_len)] if consts_len else [])
            d = FR.FragmentedPoly([rng.next() for _ in range(data_len)], [rng.next()], FR.Shape(fr_, 1))
            l, r = d.split_at(split_var)
            assert l.shape == r.shape
            lv, rv = l.into_vec(), r.into_vec()
            inter = []
            for c in range(0, len(lv), sector):
                inter += lv[c:c + sector] + rv[c:c + sector]
            assert d.into_vec() == inter


def test_evaluate():
    """fragmented.rs:1266-1282: FragmentedPoly::evaluate == dense evaluate"""
    rng = F.SplitMix64(5)
    for _ in range(10):
        shape = rand_shape_by_frag_spec(rng, 10, 10, 1)
        p = FR.FragmentedPoly([rng.next() for _ in range(shape.data_len)], [rng.next()], shape, mod=F.P)
        flat = p.into_vec()
        for _ in range(3):
            pt = [rng.next_fr() for _ in range(p.num_vars())]
            assert p.evaluate(pt) == PL.evaluate_poly(flat, pt)


# ------------------------------------------------------------------ copoly.rs
def test_segment_split():
    """copoly.rs:852-869, all 128 x 129 segments; the library's host gm_segment_split gives the same subsets"""
    k = KATS["test_segment_split"]
    for start in range(k["max_start"]):
        for end in range(start, k["max_end"]):
            ss = CP.compute_segment_split(start, end)
            if not ss:
                assert end == start
            else:
                assert ss[0].start == start and ss[-1].end() == end
                for a, b in zip(ss, ss[1:]):
                    assert a.end() == b.start
            if (start + end) % 7 == 0:  # a sample through the C ABI (ctypes call overhead)
                assert H.segment_split(start, end) == [(s.start, s.loglength) for s in ss]


def test_eq_sum_materialize_ip():
    """copoly.rs:871-940 (test_eq_sum, test_eq_materialize, test_eq_ip): 6 variables, every segment"""
    nv = KATS["test_eq_sum_materialize_ip"]["num_vars"]
    rng = F.SplitMix64(6)
    point = [rng.next_fr() for _ in range(nv)]
    mult = rng.next_fr()
    naive = PL.eq_poly_sequence_from_multiplier(1, point)[-1]
    eq = CP.EqPoly(point, mult)
    vals = [rng.next_fr() for _ in range(1 << nv)]
    for start in range(0, 1 << nv):
        for end in range(start, (1 << nv) + 1):
            e = sum(naive[i] for i in range(start, end) if i % 2 == 0) * mult % F.P
            o = sum(naive[i] for i in range(start, end) if i % 2 == 1) * mult % F.P
            assert eq.half_sums_segment(start, end) == (e, o)
            if (start * 3 + end) % 5 == 0:
                assert eq.materialize_segment(start, end) == [naive[i] * mult % F.P for i in range(start, end)]
                assert CP.EqPoly(point).ip_segment(start, end, vals[start:end]) == \
                    sum(v * n for v, n in zip(vals[start:end], naive[start:end])) % F.P


def test_materialize_eq_with_shape_and_split():
    """copoly.rs:492-567, 600-633 on the KAT shape of split_poly and on random shapes: values = eq at the data cells, sums =
    eq summed over each constant's segments; materialize_split = even / odd halves on the split shape"""
    rng = F.SplitMix64(8)
    shapes = [FR.Shape(frags(KATS["split_poly"]["fragments"]), 2), FR.Shape(frags(KATS["split_shape"]["fragments"]), 4)]
    shapes += [rand_shape_by_frag_spec(rng, 10, 10, 3) for _ in range(20)]
    for shape in shapes:
        n = len(shape)
        if n < 2:
            continue
        nv = n.bit_length() - 1
        point = [rng.next_fr() for _ in range(nv)]
        mult = rng.next_fr()
        full = [v * mult % F.P for v in PL.eq_poly_sequence_from_multiplier(1, point)[-1]]

        def expect(sh, table):
            vals = [None] * sh.data_len
            sums = [0] * sh.num_consts
            for f in sh.fragments:
                for i in range(f.len):
                    if f.content == FR.DATA:
                        vals[f.mem_idx + i] = table[f.start + i]
                    else:
                        sums[f.mem_idx] = (sums[f.mem_idx] + table[f.start + i]) % F.P
            return vals, sums
        eq = CP.EqPoly(point, mult)
        assert eq.materialize_eq_with_shape(shape) == expect(shape, full)
        eq.take_shape(shape)
        a, b = eq.materialize_split()
        assert a == expect(shape.split(), full[0::2]) and b == expect(shape.split(), full[1::2])


# ------------------------------------------------------------------ kzg.rs
def test_quotient():
    """kzg.rs:165-172"""
    k = KATS["quotient"]
    poly, pt, x = k["poly"], k["pt"], k["check_at"]
    q, rem = K.div_by_linear(poly, pt)
    assert K.ev(poly, pt) == rem
    assert K.ev(poly, x) == (K.ev(q, x) * (x - pt) + rem) % F.P
    # integer cross-check of the literal numbers (no modular wrap below p for the remainder: 4*322^7 < 2^64)
    assert rem == sum(c * pt ** i for i, c in enumerate(poly)) % F.P


# ------------------------------------------------------------------ binary_msm.rs, pullback.rs
@pytest.mark.parametrize("name", ["bin_msm", "bin_msm_gamma_3"])
def test_bin_msm(name):
    """binary_msm.rs:63-95"""
    k = KATS[name]
    rng = F.SplitMix64(20 + k["gamma"])
    bits = [bool(rng.next() & 1) for _ in range(k["num"])]
    bases = G.random_points(k["num"], 30 + k["gamma"])
    res = G.binary_msm(G.prepare_coefs(bits, k["gamma"]), G.prepare_bases(bases, k["gamma"]))
    exp = None
    for b, p in zip(bits, bases):
        if b:
            exp = G.add(exp, p)
    assert res == exp


def test_bucketed_msm():
    """pullback.rs:85-105 (1024 bases, 64 image values)"""
    k = KATS["test_bucketed_msm"]
    rng = F.SplitMix64(31)
    mapping = [rng.next() % k["image_size"] for _ in range(k["num_bases"])]
    image = [rng.next_fr() for _ in range(k["image_size"])]
    bases = G.random_points(k["num_bases"], 32)
    lhs = G.msm_nonaff(bases, G.pullback_values(mapping, image))
    assert lhs == G.pullback_bucketed_msm(mapping, image, bases)


# ------------------------------------------------------------------ triangle_add.rs, bintree_add.rs (oracle side)
def rand_points_affine(rng, row_logsize, col_logsize, seed):
    """VecVecPolynomial::rand_points_affine (vecvec.rs:347-377): random number of rows of random lengths; pads (0,0), (1,1)"""
    nrows = rng.next() % (1 << col_logsize) + 1
    lens = [rng.next() % (1 << row_logsize) + 1 for _ in range(nrows)]
    pts = F.random_points(sum(lens), seed)
    rows, at = [], 0
    for ln in lens:
        rows.append(pts[at:at + ln])
        at += ln
    xs = PL.VecVec([[p[0] for p in r] for r in rows], 0, 0, row_logsize, col_logsize)
    ys = PL.VecVec([[p[1] for p in r] for r in rows], 1, 1, row_logsize, col_logsize)
    return [xs, ys]


def triangle_inputs(num_vars, split_hi, seed, projective=True):
    """2^num_vars random points -> the 12 columns TriangleAddWG::new takes (triangle_add.rs:296-312)"""
    rng = F.SplitMix64(seed)
    pts = F.random_points(1 << num_vars, seed + 1)
    cols = [[], [], []]
    for (x, y) in pts:
        z = (rng.next_fr() or 1) if projective else 1
        cols[0].append(x * z % F.P)
        cols[1].append(y * z % F.P)
        cols[2].append(z)
    s1 = PL.dense_algfn_map_split(cols, IdAlgFn(3), PL.HI(split_hi), 3)
    s2 = PL.dense_algfn_map_split(s1, RepeatedAlgFn(IdAlgFn(3), 2), PL.HI(split_hi), 3)
    return pts, s2


def check_triangle_result(pts, last, num_vars, split_hi):
    """triangle_add.rs:330-350: sum_i i * P[idx*chunk + i] == sum_{i>=1} 2^(i-1) * result_i[idx]"""
    chunk = 1 << (num_vars - split_hi)
    npts = len(last) // 3
    for idx in range(1 << split_hi):
        target = (0, 1)
        for i in range(chunk):
            target = F.te_add_affine(target, F.te_mul_affine(pts[idx * chunk + i], i))
        got = (0, 1)
        for i in range(1, npts):
            p = F.proj_to_affine(last[3 * i][idx], last[3 * i + 1][idx], last[3 * i + 2][idx])
            got = F.te_add_affine(got, F.te_mul_affine(p, 1 << (i - 1)))
        assert got == target


def test_triangle_witness_gen_oracle():
    """triangle_add.rs:277-355 at num_vars = 8, HI(2) on the CPU (the literal 12 / HI(4) case runs on the GPU test)"""
    nv, hi = 8, 2
    pts, s2 = triangle_inputs(nv, hi, 900)
    adv = GK.triangle_witness_build(s2, nv - 2, PL.HI(hi))
    last = GK.triangle_last_step(adv[-1][1], nv - 2 - hi)
    check_triangle_result(pts, last, nv, hi)


def test_triangle_prove_and_verify_oracle():
    """triangle_add.rs:357-393 (num_vars 8, HI(2)): final claims == the 12 input columns at the final point"""
    k = KATS["triangle_prove_and_verify"]
    nv, hi = k["num_vars"], k["split_hi"]
    _, s2 = triangle_inputs(nv, hi, 910)
    adv = GK.triangle_witness_build(s2, nv - 2, PL.HI(hi))
    out = GK.triangle_last_step(adv[-1][1], nv - 2 - hi)
    rng = F.SplitMix64(911)
    point = [rng.next_fr() for _ in range(hi)]
    claims = (point, [PL.evaluate_poly(o, point) for o in out])
    tr = TapeTranscript([rng.next_bits(128) for _ in range(400)])
    fin = GK.simple_gkr_prove(tr, GK.triangle_protocol_layers(nv - 2, PL.HI(hi)), adv, claims)
    assert fin[1] == [PL.evaluate_poly(c, fin[0]) for c in s2]


@pytest.mark.parametrize("num_adds,row_logsize,col_logsize", [tuple(c) for c in KATS["bintree_prove_and_verify"]["cases"]])
def test_bintree_prove_and_verify_oracle(num_adds, row_logsize, col_logsize):
    """bintree_add.rs:401-458: final claims == the 4 input polynomials at the final point"""
    rng = F.SplitMix64(920 + row_logsize)
    points = rand_points_affine(rng, row_logsize, col_logsize, 921)
    inputs = PL.vecvec_map_split(points, IdAlgFn(2), PL.LO(0), 2)
    adv = GK.bintree_witness_build(("VV", inputs), row_logsize, num_adds, False)
    last = GK.bintree_last_step(adv[-1], num_adds - 1)
    nv = row_logsize + col_logsize
    dense_out = [p.to_dense() for p in last[1]] if last[0] == "VV" else last[1]
    point = [rng.next_fr() for _ in range(nv - num_adds)]
    claims = (point, [PL.evaluate_poly(o, point) for o in dense_out])
    tr = TapeTranscript([rng.next_bits(128) for _ in range(600)])
    fin = GK.simple_gkr_prove(tr, GK.bintree_protocol_layers(nv, num_adds, row_logsize, False), adv, claims)
    assert fin[1] == [PL.evaluate_poly(p.to_dense(), fin[0]) for p in inputs]
