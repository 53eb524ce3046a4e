"""CPU tests of the pushforward-argument oracle (oracle/pyref/pushforward.py): the protocol's own consistency checks
(logup total, per-round sum checks, asserted inside the restatement as in the reference) on real bucketing data, and
Pattern A -- every final claim equals the direct evaluation of the polynomial it is about (pushforward.rs tests,
`evaluate_poly` sanity asserts :811, the verifier's final combinator checks)."""
import pytest

from pyref import field as F
from pyref import gkr as G
from pyref import polys as PL
from pyref import pushforward as PF
from pyref.sumcheck import TapeTranscript


def _setup(x_log, d_log, nbits, seed):
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    pts = F.random_points(n, seed)
    sc = F.random_scalars(n, nbits, seed + 1)
    sc[0] = 0
    _, digits, counter = G.bucketing_image(pts, sc, y_size, y_log, d_log, x_log)
    return y_size, y_log, pts, digits, counter


@pytest.mark.parametrize("x_log,d_log,nbits", [(3, 2, 8), (4, 2, 6), (3, 3, 15), (5, 2, 4)])
def test_pushforward_prover_claims_are_evaluations(x_log, d_log, nbits):
    y_size, y_log, pts, digits, counter = _setup(x_log, d_log, nbits, 3)
    rng = F.SplitMix64(17)
    r = [rng.next_fr() for _ in range(y_log + d_log + x_log)]
    p1 = PF.phase1_data(pts, digits, counter, x_log, d_log)
    p2 = PF.phase2_data(digits, counter, r, y_log, d_log, x_log)
    # the claims the image part hands over: evaluations of the image polynomials (x, y, z) at r
    img, _, _ = G.bucketing_image(pts, _scalars(digits, d_log), y_size, y_log, d_log, x_log)
    evs = [PL.evaluate_poly(p.to_dense(), r) for p in img]
    tape = [rng.next_bits(512) for _ in range(3000)]
    tr = TapeTranscript(tape)
    res = PF.pushforward_prove(tr, x_log, y_log, y_size, d_log, (r, evs), p1, p2)
    mpt, mevs = res["matrix"]
    mlog = x_log + y_log
    pad = lambda v: PF.pad_vector(v, mlog, 0)
    g = PF.make_gamma_pows(res["gamma"], 5)
    p_folded = [(a + g[1] * (b - 1) + g[2]) % F.P for a, b in zip(p1["p_0"], p1["p_1"])]
    assert mevs[0] == (PL.evaluate_poly(p_folded, mpt[y_log:]) + res["gamma"]) % F.P
    assert mevs[1] == PL.evaluate_poly(pad(p2["c_pull"]), mpt)
    assert mevs[2] == PL.evaluate_poly(pad(p2["d_pull"]), mpt)
    assert mevs[3] == PL.evaluate_poly(pad(p1["c"]), mpt)
    assert mevs[4] == PL.evaluate_poly(pad(p1["d"]), mpt)
    pt_c, ev_c = res["ac_c"]
    assert len(pt_c) == x_log and ev_c[0] == PL.evaluate_poly(p1["ac_c"], pt_c)
    pt_d, ev_d = res["ac_d"]
    assert len(pt_d) == d_log and ev_d[0] == PL.evaluate_poly(p1["ac_d"], pt_d)


def _scalars(digits, d_log):
    n = len(digits[0])
    return [sum(int(digits[y][x]) << (y * d_log) for y in range(len(digits))) for x in range(n)]


def test_eq_trunc_and_selector_match_their_tables():
    """verifier_polys.rs:172-196 (truncated_eq_tests, selector_poly_tests) at 5 variables"""
    rng = F.SplitMix64(4)
    nv = 5
    u = [rng.next_fr() for _ in range(nv)]
    v = [rng.next_fr() for _ in range(nv)]
    eq_u, eq_v = PL.eq_poly_sequence_last(u), PL.eq_poly_sequence_last(v)
    for k in range((1 << nv) + 1):
        assert PF.eq_trunc_evaluate(nv, k, u, v) == sum(a * b for a, b in zip(eq_u[:k], eq_v[:k])) % F.P
        assert PF.selector_evaluate(nv, k, v) == sum(eq_v[:k]) % F.P
        assert PF.eq_trunc_evals(nv, k, u) == [e if i < k else 0 for i, e in enumerate(eq_u)]
