"""GPU parity of "prove pushforward" (PushforwardProtocol::prove + the logup main phase) through the C ABI vs the Python
oracle with the same challenge tape: every prover message, gamma, and the three final claims; chained after the image part
exactly as Pippenger::prove does (pippenger.rs:138-160)."""
import pytest

from gkr_msm_amd import codec, harness as H
from pyref import field as F
from pyref import gkr as G
from pyref import polys as PL
from pyref import pushforward as PF
from pyref.sumcheck import TapeTranscript

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("x_log,d_log,nbits", [(3, 2, 8), (4, 2, 6), (3, 3, 15), (5, 2, 4), (6, 3, 24), (5, 4, 16),
                                              (10, 8, 40)])   # pushforward_works (pushforward.rs:1050-1189): x_logsize 10, y_size 5, d_logsize 8
def test_pushforward_matches_oracle(x_log, d_log, nbits):
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    pts = F.random_points(n, 11 + x_log)
    sc = F.random_scalars(n, nbits, 12 + d_log)
    sc[0] = 0
    image, digits, counter = G.bucketing_image(pts, sc, y_size, y_log, d_log, x_log)
    rng = F.SplitMix64(31)
    r = [rng.next_fr() for _ in range(y_log + d_log + x_log)]
    evs = [PL.evaluate_poly(p.to_dense(), r) for p in image]
    tape = [rng.next_bits(512) % F.P for _ in range(3000)]
    tr = TapeTranscript(tape)
    p1 = PF.phase1_data(pts, digits, counter, x_log, d_log)
    p2 = PF.phase2_data(digits, counter, r, y_log, d_log, x_log)
    want = PF.pushforward_prove(tr, x_log, y_log, y_size, d_log, (r, evs), p1, p2)

    d_pts = H.to_dev(codec.points_to_mont(pts))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(sc)))
    # the device draws challenge(128) as the low 128 bits too: feed canonical values, masked where the oracle masks
    got = H.pushforward_prove(plan, d_pts, y_log, r, evs, _device_tape(tape, tr))
    assert got["tape_used"] == tr.pos
    assert got["msgs"] == [v for m in tr.msgs for v in m]
    assert got["gamma"] == want["gamma"]
    assert (list(got["matrix"][0]), list(got["matrix"][1])) == (want["matrix"][0], want["matrix"][1])
    assert (list(got["ac_c"][0]), list(got["ac_c"][1])) == (list(want["ac_c"][0]), list(want["ac_c"][1]))
    assert (list(got["ac_d"][0]), list(got["ac_d"][1])) == (list(want["ac_d"][0]), list(want["ac_d"][1]))
    assert got["rounds"] == sum(range(x_log + y_log)) + x_log + d_log + x_log + y_log


@pytest.mark.parametrize("kind", ["all_same", "zero", "max"])
def test_pushforward_with_every_point_in_one_bucket_per_window(kind):
    """collisions at their maximum: every scalar equal, so one bucket per window holds all 2^x_logsize points -- the counter column
    reaches X - 1, the access counts are one entry of X and zeros elsewhere (the logup fractions of the untouched rows)"""
    x_log, d_log, nbits = 4, 2, 6
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    pts = F.random_points(n, 77)
    sc = {"all_same": [0b100111] * n, "zero": [0] * n, "max": [(1 << nbits) - 1] * n}[kind]
    image, digits, counter = G.bucketing_image(pts, sc, y_size, y_log, d_log, x_log)
    assert max(max(c) for c in counter) == n - 1
    rng = F.SplitMix64(31)
    r = [rng.next_fr() for _ in range(y_log + d_log + x_log)]
    evs = [PL.evaluate_poly(p.to_dense(), r) for p in image]
    tape = [rng.next_bits(512) % F.P for _ in range(3000)]
    tr = TapeTranscript(tape)
    p1 = PF.phase1_data(pts, digits, counter, x_log, d_log)
    p2 = PF.phase2_data(digits, counter, r, y_log, d_log, x_log)
    want = PF.pushforward_prove(tr, x_log, y_log, y_size, d_log, (r, evs), p1, p2)
    d_pts = H.to_dev(codec.points_to_mont(pts))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(sc)))
    got = H.pushforward_prove(plan, d_pts, y_log, r, evs, _device_tape(tape, tr))
    assert got["tape_used"] == tr.pos
    assert got["msgs"] == [v for m in tr.msgs for v in m]
    assert got["gamma"] == want["gamma"]
    assert (list(got["matrix"][0]), list(got["matrix"][1])) == (want["matrix"][0], want["matrix"][1])
    assert (list(got["ac_c"][0]), list(got["ac_c"][1])) == (list(want["ac_c"][0]), list(want["ac_c"][1]))
    assert (list(got["ac_d"][0]), list(got["ac_d"][1])) == (list(want["ac_d"][0]), list(want["ac_d"][1]))
    plan.close()


def _device_tape(tape, tr):
    """the challenges as the oracle consumed them (4 x 512-bit reduced mod p, then 128-bit truncations)"""
    return [t % F.P if i < 4 else t & ((1 << 128) - 1) for i, t in enumerate(tape[: tr.pos])] + [0] * 8


@pytest.mark.parametrize("nvars,nargs", [(1, 1), (3, 4), (6, 4), (5, 8), (9, 2)])
def test_multiopen_reduction_matches_oracle(nvars, nargs):
    """MultiOpenReduction::prove (multiopen_reduction.rs:65-93) vs the oracle; Pattern A on the output claims"""
    rng = F.SplitMix64(50 + nvars)
    n = 1 << nvars
    polys = [[rng.next_fr() for _ in range(n)] for _ in range(nargs)]
    polys[0][n // 2:] = [0] * (n - n // 2)            # zero-padded tail, as the caller's witness has (pippenger.rs:233)
    points = [[rng.next_fr() for _ in range(nvars)] for _ in range(nargs)]
    claims = [(pt, PL.evaluate_poly(p, pt)) for p, pt in zip(polys, points)]
    tape = [rng.next_bits(128) for _ in range(200)]
    tr = TapeTranscript(tape)
    want_pt, want_evs = PF.multiopen_prove(tr, nvars, claims, polys)
    cols = [H.to_dev(codec.to_mont_limbs(p)).reshape(-1) for p in polys]
    got = H.multiopen_prove(cols, nvars, points, [ev for _, ev in claims], tape)
    assert got["msgs"] == [v for m in tr.msgs for v in m]
    assert got["point"] == want_pt and got["evs"] == want_evs
    assert got["tape_used"] == tr.pos and got["rounds"] == nvars
    assert got["evs"] == [PL.evaluate_poly(p, got["point"]) for p in polys]
