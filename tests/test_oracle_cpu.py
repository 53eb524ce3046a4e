"""CPU tests (no GPU): pin the C oracle (oracle/gkrmsm_oracle.c) against
  * the reference's integer KAT COEFF_D (/root/reference/src/utils.rs:35),
  * the independent Python big-int restatement (oracle/pyref),
  * the algebraic identities the reference's own tests assert (Pattern C: layers == affine group law,
    /root/reference/src/cleanup/protocols/gkrs/bintree_add.rs:507-638; MSM == naive sum, pippenger.rs:598-605)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gkr_msm_amd import codec
from pyref import algfn as A
from pyref import field as F
from pyref import gkr as G
from pyref.polys import log2_exact


def test_coeff_d_kat():
    out = np.zeros(4, dtype=np.uint64)
    O.lib().or_coeff_d(out.ctypes.data)
    assert out.tolist() == F.COEFF_D_MONT_LIMBS
    assert codec.from_mont_limbs(out)[0] == F.TE_D
    # d is what the curve equation needs: the arkworks generator is on a*x^2 + y^2 = 1 + d*x^2*y^2
    gx = 18886178867200960497001835917649091219057080094937609519140440539760939937304
    gy = 19188667384257783945677642223292697773471335439753913231509108946878080696678
    assert F.te_on_curve(gx, gy) and F.te_mul_affine((gx, gy), F.BS_ORDER) == (0, 1)


def test_field_ops_vs_bigint():
    rng = F.SplitMix64(11)
    n = 500
    a = [rng.next_fr() for _ in range(n)]
    b = [rng.next_fr() for _ in range(n)]
    a[:4] = [0, F.P - 1, 1, F.P - 2]
    b[:4] = [0, F.P - 1, F.P - 1, 2]
    A_, B_ = codec.to_mont_limbs(a), codec.to_mont_limbs(b)
    o = np.zeros_like(A_)

    def run(op):
        O.lib().or_fr_batch(op, A_.ctypes.data, B_.ctypes.data, o.ctypes.data, n)
        return codec.from_mont_limbs(o)
    assert run(0) == [(x + y) % F.P for x, y in zip(a, b)]
    assert run(1) == [(x - y) % F.P for x, y in zip(a, b)]
    assert run(2) == [(x * y) % F.P for x, y in zip(a, b)]
    assert run(3) == [(-x) % F.P for x in a]
    assert run(7) == [(-5 * x) % F.P for x in a]
    assert run(8) == [(F.TE_D * x) % F.P for x in a]
    inv = run(4)
    assert all(x * y % F.P == 1 for x, y in zip(a[1:], inv[1:]))


FNS = {
    "aff_l1": ([(1, 1)], A.AFF_L1), "aff_l2": ([(2, 1)], A.AFF_L2), "aff_l3": ([(3, 1)], A.AFF_L3),
    "proj_l1": ([(4, 1)], A.PROJ_L1), "proj_l2": ([(5, 1)], A.PROJ_L2), "proj_l3": ([(6, 1)], A.PROJ_L3),
    "tri+rep2": ([(7, 1), (4, 2)], A.StackedAlgFn(A.TRI_L1, A.RepeatedAlgFn(A.PROJ_L1, 2))),
    "aff_l1+bitcheck": ([(1, 1), (9, 2)], A.StackedAlgFn(A.AFF_L1, A.RepeatedAlgFn(A.BitCheckFn(), 2))),
    "id3x2": ([(8, 6)], A.RepeatedAlgFn(A.IdAlgFn(3), 2)),
}


@pytest.mark.parametrize("name", sorted(FNS))
def test_algfn_vs_pyref(name):
    segs, pyf = FNS[name]
    f = O.make_fn(*segs)
    L = O.lib()
    assert L.or_fn_n_ins(C.byref(f)) == pyf.n_ins and L.or_fn_n_outs(C.byref(f)) == pyf.n_outs
    rng = F.SplitMix64(len(name))
    for _ in range(5):
        row = [rng.next_fr() for _ in range(pyf.n_ins)]
        i = codec.to_mont_limbs(row)
        o = np.zeros((pyf.n_outs, 4), dtype=np.uint64)
        L.or_fn_exec(C.byref(f), i.ctypes.data, o.ctypes.data)
        assert codec.from_mont_limbs(o) == pyf.exec(row)


def test_layers_equal_group_law():
    """Pattern C: l3(l2(l1(.))) is the twisted-Edwards sum (affine and projective inputs)."""
    pts = F.random_points(8, 5)
    L = O.lib()
    aff = O.make_fn((1, 1)), O.make_fn((2, 1)), O.make_fn((3, 1))
    prj = O.make_fn((4, 1)), O.make_fn((5, 1)), O.make_fn((6, 1))

    def chain(fns, row):
        cur = codec.to_mont_limbs(row)
        for f in fns:
            o = np.zeros((L.or_fn_n_outs(C.byref(f)), 4), dtype=np.uint64)
            L.or_fn_exec(C.byref(f), cur.ctypes.data, o.ctypes.data)
            cur = o
        return codec.from_mont_limbs(cur)
    rng = F.SplitMix64(3)
    for i in range(0, 8, 2):
        p, q = pts[i], pts[i + 1]
        X, Y, Z = chain(aff, [p[0], p[1], q[0], q[1]])
        assert F.proj_to_affine(X, Y, Z) == F.te_add_affine(p, q)
        z1, z2 = rng.next_fr(), rng.next_fr()
        X, Y, Z = chain(prj, [p[0] * z1 % F.P, p[1] * z1 % F.P, z1, q[0] * z2 % F.P, q[1] * z2 % F.P, z2])
        assert F.proj_to_affine(X, Y, Z) == F.te_add_affine(p, q)
        # doubling and identity go through the same unified formulas
        X, Y, Z = chain(aff, [p[0], p[1], p[0], p[1]])
        assert F.proj_to_affine(X, Y, Z) == F.te_add_affine(p, p)
        X, Y, Z = chain(aff, [p[0], p[1], 0, 1])
        assert F.proj_to_affine(X, Y, Z) == p


def test_dense_fold_and_eq_table():
    from pyref import polys as PL
    rng = F.SplitMix64(21)
    n = 64
    v = [rng.next_fr() for _ in range(n)]
    t = rng.next_fr()
    V, T = codec.to_mont_limbs(v), codec.to_mont_limbs([t])
    o = np.zeros((n // 2, 4), dtype=np.uint64)
    L = O.lib()
    L.or_dense_bind(V.ctypes.data, n, T.ctypes.data, o.ctypes.data)
    assert codec.from_mont_limbs(o) == PL.bind_dense(v, t)
    w = list(v)
    PL.dense_make_21(w)
    V2 = V.copy()
    L.or_dense_make21(V2.ctypes.data, n)
    assert codec.from_mont_limbs(V2) == w
    L.or_dense_bind21(V2.ctypes.data, n, T.ctypes.data, o.ctypes.data)
    assert codec.from_mont_limbs(o) == PL.dense_bind_21(w, t) == PL.bind_dense(v, t)
    pt = [rng.next_fr() for _ in range(6)]
    mult = rng.next_fr()
    e = np.zeros((64, 4), dtype=np.uint64)
    PT, M = codec.to_mont_limbs(pt), codec.to_mont_limbs([mult])
    L.or_eq_table(M.ctypes.data, PT.ctypes.data, 6, e.ctypes.data)
    assert codec.from_mont_limbs(e) == PL.eq_poly_sequence_from_multiplier(mult, pt)[-1]


MSM_SHAPES = [(2, 2, 12), (4, 2, 12), (5, 3, 16), (6, 2, 10), (3, 3, 24), (8, 4, 32), (7, 6, 128), (9, 8, 64),
              (10, 9, 27), (10, 10, 30)]


@pytest.mark.parametrize("x_log,d_log,nbits", MSM_SHAPES)
def test_msm_vs_pyref(x_log, d_log, nbits):
    y_size = (nbits + d_log - 1) // d_log
    y_log = log2_exact(y_size)
    n = 1 << x_log
    pts = F.random_points(n, 100 + x_log)
    sc = F.random_scalars(n, nbits, 200 + d_log)
    if n >= 4:
        sc[1] = 0
        sc[2] = sc[3]
    image, digits, counter, wg = G.pippenger_witness(pts, sc, y_size, y_log, d_log, x_log)
    out = G.pippenger_dense_output(wg, y_log, d_log)
    expect_pt = G.pippenger_final_point(out, d_log)
    r = O.msm(codec.points_to_mont(pts), codec.ints_to_limbs(sc), x_log, d_log, y_size, threads=2)
    assert r["digits"].tolist() == digits
    assert r["counter"].tolist() == counter
    nrows = y_size << d_log
    for c, k in enumerate(("bx", "by", "bz")):
        assert codec.from_mont_limbs(r[k]) == wg.bucket_sums[c][:nrows]
    for c in range(3 * (d_log + 1)):
        assert codec.from_mont_limbs(r["window_cols"][c]) == out[c][:y_size]
    xy = codec.from_mont_limbs(O.msm_combine(r["window_cols"], d_log))
    assert tuple(xy) == expect_pt
    if n <= 64:
        acc = (0, 1)
        for p, s in zip(pts, sc):
            acc = F.te_add_affine(acc, F.te_mul_affine(p, s))
        assert tuple(xy) == acc


def test_msm_window_shards_concatenate():
    x_log, d_log, y_size = 7, 4, 8
    n = 1 << x_log
    pts = codec.points_to_mont(F.random_points(n, 1))
    sc = codec.ints_to_limbs(F.random_scalars(n, 32, 2))
    full = O.msm(pts, sc, x_log, d_log, y_size)
    parts = [O.msm(pts, sc, x_log, d_log, y_size, a, b)["window_cols"] for (a, b) in [(0, 3), (3, 4), (4, 8)]]
    assert np.array_equal(np.concatenate(parts, axis=1), full["window_cols"])


def test_msm_rejects_bad_shapes():
    pts = np.zeros((4, 8), dtype=np.uint64)
    sc = np.zeros((4, 4), dtype=np.uint64)
    with pytest.raises(ValueError):
        O.msm(pts, sc, 2, 8, 33)     # y_size * d > 256 (pushforward.rs:358 would panic)
    with pytest.raises(ValueError):
        O.msm(pts, sc, 2, 1, 4)      # d_logsize < 2
