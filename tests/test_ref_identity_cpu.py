"""The reference's own identity tests for the round polynomials, replayed on the oracle (no GPU): the optimised sumcheck objects
of oracle/pyref/sumcheck.py against the reference's NAIVE object restated independently in oracle/pyref/naive.py
(ExampleSumcheckObjectSO, sumcheck.rs:132-235), at the reference's literal shapes:

  check_univars  sumchecks/dense_eq.rs:258-344    num_vars 6, twisted_edwards_add_l1, gamma = 2, 100 iterations
  check_univars  sumchecks/vecvec_eq.rs:511-600   num_vars 6, num_vertical_vars in {0, 1, 3} x {Full, Rows, Nothing}, 100 iterations each
                                                   (here 100 / 40 / 40 iterations: the pure-Python objects take ~30 ms per run)

Per round: the two round polynomials agree at 0, 1, 2, 3 and coefficient by coefficient; after every bind the naive object's
claim (the plain sum) equals the optimised object's running claim; the final evaluations agree.
tests/test_ref_identity_gpu.py holds the HIP objects against the same naive object."""
import pytest

from pyref import algfn as A
from pyref import field as F
from pyref import naive as N
from pyref import polys as PL
from pyref import sumcheck as SC

from ref_identity_common import dense_rand_points, vecvec_py, vecvec_rand_points

P = F.P


def _vec_claims(f, dense_cols, eq):
    """vec_claim of the reference tests: sum_i eq[i] f_o(cols[.][i]) per output o"""
    acc = [0] * f.n_outs
    for i in range(len(eq)):
        out = f.exec([c[i] for c in dense_cols])
        for o in range(f.n_outs):
            acc[o] = (acc[o] + out[o] * eq[i]) % P
    return acc


def _compare_rounds(opt, naive, num_vars, rng, opt_claim):
    for _ in range(num_vars):
        a, b = opt.unipoly(), naive.unipoly()
        for x in (0, 1, 2, 3):
            assert N.evaluate(a, x) == N.evaluate(b, x)
        assert list(a) == list(b)
        t = rng.next_fr()
        opt.bind(t)
        naive.bind(t)
        assert naive.claim() == opt_claim(opt)


def test_interpolation_agrees_with_the_lagrange_construction():
    rng = F.SplitMix64(77)
    for n in (1, 2, 3, 4, 5):
        ev = [rng.next_fr() for _ in range(n)]
        assert N.interpolate_gauss(ev) == list(SC.unipoly_from_evals(ev))
    pt = [rng.next_fr() for _ in range(5)]
    assert N.eq_table(pt) == PL.eq_poly_sequence_last(pt)


def test_dense_check_univars():
    """dense_eq.rs:258-344"""
    rng = F.SplitMix64(0xD15E)
    num_vars, gamma, f = 6, 2, A.PROJ_L1
    for _ in range(100):
        cols = dense_rand_points(rng, num_vars)
        data = cols + [list(c) for c in cols]                     # data_l ++ data_r: six columns, every pair is P + P's inputs
        point = [rng.next_fr() for _ in range(num_vars)]
        eq = N.eq_table(point)
        naive = N.ExampleSumcheckObjectSO(data + [eq], N.GammaEq(f, gamma), num_vars)
        sum_claim = sum(N.GammaEq(f, gamma).exec([c[i] for c in data] + [eq[i]]) for i in range(1 << num_vars)) % P
        assert naive.claim() == sum_claim
        opt = SC.DenseDeg2SumcheckObjectSO.rlc(data, f, _vec_claims(f, data, eq), point, gamma)
        assert opt.claim == sum_claim
        _compare_rounds(opt, naive, num_vars, rng, lambda o: o.claim)
        assert opt.final_evals() == naive.final_evals()[:-1]


@pytest.mark.parametrize("num_vertical_vars", [0, 1, 3])
@pytest.mark.parametrize("denseness,iters", [("full", 40), ("rows", 40), ("nothing", 100)])
def test_vecvec_check_univars(num_vertical_vars, denseness, iters):
    """vecvec_eq.rs:511-600"""
    rng = F.SplitMix64(0xBEC0 + 16 * num_vertical_vars + len(denseness))
    num_vars, gamma, f = 6, 2, A.PROJ_L1
    for _ in range(iters):
        data3, pads = vecvec_rand_points(rng, num_vars - num_vertical_vars, num_vertical_vars, denseness)
        py = vecvec_py(data3, pads, num_vars - num_vertical_vars, num_vertical_vars)
        py = py + [p.clone() for p in py]
        dense = [p.to_dense() for p in py]
        point = [rng.next_fr() for _ in range(num_vars)]
        eq = N.eq_table(point)
        naive = N.ExampleSumcheckObjectSO(dense + [eq], N.GammaEq(f, gamma), num_vars)
        opt = SC.VecVecDeg2SumcheckObjectSO.rlc(py, f, _vec_claims(f, dense, eq), point, num_vertical_vars, gamma)
        assert opt.claim() == naive.claim()
        _compare_rounds(opt, naive, num_vars, rng, lambda o: o.claim())
        assert opt.final_evals() == naive.final_evals()
