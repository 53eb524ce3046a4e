"""CPU tests: the C oracle's image-part prover (oracle/gkrmsm_oracle_prover.c) against the Python restatement with
the same challenge tape: dense output, every prover message, final claims; plus Pattern A (final claims are the
image polynomials at the final point; /root/reference/src/cleanup/protocols/pippenger_ending.rs:176-275)."""
import pytest

import oracle_ffi as O
from gkr_msm_amd import codec
from pyref import field as F
from pyref import gkr as G
from pyref import polys as PL
from pyref.sumcheck import TapeTranscript


@pytest.mark.parametrize("x_log,d_log,nbits,threads", [(4, 2, 12, 1), (5, 3, 16, 2), (6, 2, 10, 1), (3, 3, 24, 3),
                                                       (8, 4, 32, 4), (7, 6, 128, 2)])
def test_c_prover_matches_pyref(x_log, d_log, nbits, threads):
    y_size = (nbits + d_log - 1) // d_log
    y_log = PL.log2_exact(y_size)
    n = 1 << x_log
    pts = F.random_points(n, 7 + x_log)
    sc = F.random_scalars(n, nbits, 70 + d_log)
    sc[0] = 0
    image, digits, counter, wg = G.pippenger_witness(pts, sc, y_size, y_log, d_log, x_log)
    out = G.pippenger_dense_output(wg, y_log, d_log)
    rng = F.SplitMix64(99)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = G.pippenger_claims(out, r)
    tape = [rng.next_bits(128) for _ in range(3000)]
    tr = TapeTranscript(tape)
    fin = G.prove_image_part(tr, y_log, d_log, x_log, claims, wg)
    exp_msgs = [v for m in tr.msgs for v in m]

    w = O.PipWitness(codec.points_to_mont(pts), codec.ints_to_limbs(sc), x_log, d_log, y_size, y_log, threads)
    got_out = w.output()
    assert [codec.from_mont_limbs(got_out[c]) for c in range(got_out.shape[0])] == out
    res = w.prove_image_part(codec.to_mont_limbs(claims[0]) if y_log else codec.to_mont_limbs([0]),
                             codec.to_mont_limbs(claims[1]), codec.ints_to_limbs(tape))
    assert res["tape_used"] == tr.pos
    assert codec.from_mont_limbs(res["msgs"]) == exp_msgs
    assert codec.from_mont_limbs(res["point"]) == fin[0]
    assert codec.from_mont_limbs(res["evs"]) == fin[1]
    for i in range(3):
        assert PL.evaluate_poly(image[i].to_dense(), fin[0]) == fin[1][i]


@pytest.mark.parametrize("lp,lb,threads", [(1, 1, 1), (2, 1, 2), (3, 2, 1), (4, 3, 3), (3, 5, 2)])
def test_c_gen1_prover_matches_pyref(lp, lb, threads):
    """gen-1 gkr_msm_prove: C oracle vs the Python restatement (+ Pattern A on the final claim)"""
    import numpy as np
    from pyref import gen1 as G1
    pts = F.random_points(1 << lp, 3 + lp)
    rng = F.SplitMix64(40 + lb)
    bits = [[bool(rng.next() & 1) for _ in range(1 << lb)] for _ in range(1 << lp)]
    tape = [rng.next_fr() for _ in range(3000)]
    claim, out, tr = G1.gkr_msm_prove(bits, pts, lp, lb, tape)
    b8 = np.array([[1 if b else 0 for b in s] for s in bits], dtype=np.uint8).reshape(-1)
    res = O.gkr_msm_prove(codec.points_to_mont(pts), b8, lp, lb, codec.ints_to_limbs(tape), threads)
    assert res["tape_used"] == tr.pos
    assert codec.from_mont_limbs(res["msgs"]) == tr.msgs
    assert codec.from_mont_limbs(res["output"]) == [v for p in out for v in p]
    assert codec.from_mont_limbs(res["point"]) == claim[0] and codec.from_mont_limbs(res["evs"]) == claim[1]
    base = G1.base_layer(bits, pts, lp, lb)
    assert [G1.evaluate(b, claim[0]) for b in base] == claim[1]
