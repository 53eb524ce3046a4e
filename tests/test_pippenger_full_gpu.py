"""GPU test of the whole gen-2 prover, gm_pippenger_wg_create + gm_pippenger_prove = PippengerWG::new + Pippenger::prove
(pippenger.rs:37-70, 118-290), against the Python oracle's orchestration of the same protocol with the same challenge tape:
every scalar and every G1 point written to the transcript, in order, and the deferred pairing pair.  The SRS is built from a
known tau, so the final pairing equation <A, H0> = <B, H1> is checked in the exponent (A = tau * B): the proof verifies."""
import pytest

from gkr_msm_amd import codec, harness as H
from pyref import field as F
from pyref import g1 as G
from pyref import gkr as GK
from pyref import knuckles as KN
from pyref import pippenger as PP

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("x_log,d_log,nbits,clm", [(3, 2, 8, 0), (3, 2, 8, 1), (4, 2, 6, 0), (4, 3, 12, 2),
                                                  (3, 2, 40, 4)])   # clm = 4 (BASELINE.json configs[4]): 16 windows per commitment matrix, the last matrix partial
def test_full_prover_matches_oracle_and_verifies(x_log, d_log, nbits, clm):
    _full_prover_against_oracle(x_log, d_log, nbits, clm, None)


@pytest.mark.parametrize("kind", ["all_same", "zero"])
def test_full_prover_with_every_point_in_one_bucket_per_window(kind):
    """every scalar equal: ONE outer bucket per window holds every key point (a G1 row of 2^x_logsize points beside empty rows),
    the counter column reaches X - 1, the access counts are X in one place: commitments, messages and the pairing pair as the oracle's"""
    _full_prover_against_oracle(3, 2, 8, 1, [0b10011011] * 8 if kind == "all_same" else [0] * 8)


def _full_prover_against_oracle(x_log, d_log, nbits, clm, scalars):
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    rng = F.SplitMix64(90 + x_log + clm)
    pts = F.random_points(n, 2)
    sc = F.random_scalars(n, nbits, 3)
    sc[0] = 0
    if scalars is not None:
        sc = list(scalars)
    nv = x_log + clm
    N = 1 << nv
    tau, k = rng.next_fr(), 2
    basis, cur = [], G.GEN
    for _ in range(2 * N - 1):
        basis.append(cur)
        cur = G.mul(cur, tau)
    st = PP.pippenger_wg(pts, sc, y_size, y_log, d_log, x_log, clm, basis)
    out = GK.pippenger_dense_output(st["wg"], y_log, d_log)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = GK.pippenger_claims(out, r)
    tape = [rng.next_bits(512) for _ in range(8000)]
    tr = PP.Transcript(tape)
    inv = KN.setup_inverses(k, nv)
    want_pair = PP.pippenger_prove(tr, st, claims, y_size, y_log, d_log, x_log, clm, basis, inv, k)
    assert want_pair[0] == G.mul(want_pair[1], tau)

    d_pts = H.to_dev(codec.points_to_mont(pts))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(sc)))
    d_basis = H.g1_aff_dev(basis)
    wg = H.PippengerWG(plan, d_pts, y_log, clm, d_basis)
    assert wg.dense_output() == out
    d_inv = H.knuckles_setup(k, nv)
    # the tape as the oracle consumed it: 512-bit draws reduced mod p, 128-bit draws truncated (the oracle records which is which)
    dev_tape = _consumed(tape, tr)
    got = wg.prove(claims[0], claims[1], d_inv, k, dev_tape)
    assert got["tape_used"] == tr.pos
    assert got["points"] == tr.points
    assert got["msgs"] == [v for m in tr.msgs for v in m]
    assert got["pair"] == want_pair
    assert got["pair"][0] == G.mul(got["pair"][1], tau)          # the proof verifies


def _consumed(tape, tr):
    """replay which draws were 512-bit: psi, tau_c, tau_d, tau_suppression (the first 4 draws of the pushforward) and u"""
    wide = getattr(tr, "wide", None)
    return [t % F.P if (wide and i in wide) else t & ((1 << 128) - 1) for i, t in enumerate(tape[: tr.pos])] + [0] * 8


def test_full_prover_live_transcript_requests():
    """gm_pippenger_prove_tr: the live transcript sees the same scalars and points as the tape run, and the challenge requests
    carry the reference's sizes: one challenge_vec(4, 512) (pushforward.rs:684), one challenge(512) (pippenger.rs:197), the
    rest challenge(128)"""
    import ctypes as C
    import numpy as np
    from gkr_msm_amd import ffi
    x_log, d_log, nbits, clm = 3, 2, 8, 1
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    rng = F.SplitMix64(123)
    pts = F.random_points(n, 2)
    sc = F.random_scalars(n, nbits, 3)
    nv = x_log + clm
    basis = G.random_points((2 << nv) - 1, 4)        # any SRS works for transcript parity
    d_pts = H.to_dev(codec.points_to_mont(pts))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(sc)))
    wg = H.PippengerWG(plan, d_pts, y_log, clm, H.g1_aff_dev(basis))
    out = wg.dense_output()
    r = [rng.next_fr() for _ in range(y_log)]
    claims = GK.pippenger_claims(out, r)
    d_inv = H.knuckles_setup(2, nv)
    drawn = []

    def draw():
        v = rng.next_bits(512)
        drawn.append(v)
        return v
    live = H.LiveTranscript(draw)
    cp, ce, kk = H.fr_arg(claims[0]), H.fr_arg(claims[1]), H.fr_arg([2])
    pair = np.zeros(24, dtype=np.uint64)
    used, rounds = C.c_uint64(), C.c_uint64()
    ffi.check(ffi.lib().gm_pippenger_prove_tr(wg.h, cp.ctypes.data, ce.ctypes.data, C.c_void_p(d_inv.data_ptr()), kk.ctypes.data,
                                              C.byref(live.c), pair.ctypes.data, C.byref(used), C.byref(rounds)))
    assert live.requests.count((4, 512)) == 1 and live.requests.count((1, 512)) == 1
    assert all(rq in ((4, 512), (1, 512), (1, 128)) for rq in live.requests)
    assert sum(c for c, _ in live.requests) == used.value == len(drawn)
    # replay as a tape: identical transcript and pair
    it = iter(drawn)
    tape = []
    for cnt, bits in live.requests:
        for _ in range(cnt):
            v = next(it)
            tape.append(v % F.P if bits >= 255 else v & ((1 << bits) - 1))
    ref = wg.prove(claims[0], claims[1], d_inv, 2, tape + [0] * 8)
    assert [v for m in live.writes for v in m] == ref["msgs"]
    assert live.points == ref["points"]
    assert tuple(codec.g1_aff_from_limbs(pair)) == ref["pair"] and rounds.value == ref["rounds"]


def test_full_prover_with_builtin_merlin_transcript_produces_a_verifying_proof():
    """gm_pippenger_prove_tr driven by the library's ProofTranscript2 clone (csrc/merlin.hip): real Fiat-Shamir challenges; the
    proof bytes have the expected size, are reproducible, and the deferred pairing pair satisfies A = tau * B"""
    import ctypes as C
    import numpy as np
    from gkr_msm_amd import ffi
    L = ffi.lib()
    x_log, d_log, nbits, clm = 4, 2, 8, 1
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    rng = F.SplitMix64(321)
    pts = F.random_points(n, 12)
    sc = F.random_scalars(n, nbits, 13)
    nv = x_log + clm
    tau = rng.next_fr()
    basis, cur = [], G.GEN
    for _ in range((2 << nv) - 1):
        basis.append(cur)
        cur = G.mul(cur, tau)
    d_pts = H.to_dev(codec.points_to_mont(pts))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(sc)))
    wg = H.PippengerWG(plan, d_pts, y_log, clm, H.g1_aff_dev(basis))
    out = wg.dense_output()
    r = [rng.next_fr() for _ in range(y_log)]
    claims = GK.pippenger_claims(out, r)
    d_inv = H.knuckles_setup(2, nv)
    cp, ce, kk = H.fr_arg(claims[0]), H.fr_arg(claims[1]), H.fr_arg([2])
    proofs, pairs = [], []
    for _ in range(2):
        h = C.c_void_p()
        ffi.check(L.gm_merlin_create(b"pippenger-gpu", 13, C.byref(h)))
        tr = ffi.GmTranscript()
        ffi.check(L.gm_merlin_transcript(h, C.byref(tr)))
        pair = np.zeros(24, dtype=np.uint64)
        used, rounds = C.c_uint64(), C.c_uint64()
        ffi.check(L.gm_pippenger_prove_tr(wg.h, cp.ctypes.data, ce.ctypes.data, C.c_void_p(d_inv.data_ptr()), kk.ctypes.data,
                                          C.byref(tr), pair.ctypes.data, C.byref(used), C.byref(rounds)))
        pp, pn = C.c_void_p(), C.c_uint64()
        ffi.check(L.gm_merlin_proof(h, C.byref(pp), C.byref(pn)))
        proofs.append(C.string_at(pp, pn.value))
        pairs.append(tuple(codec.g1_aff_from_limbs(pair)))
        L.gm_merlin_destroy(h)
    assert proofs[0] == proofs[1] and pairs[0] == pairs[1]
    assert pairs[0][0] == G.mul(pairs[0][1], tau)                 # the proof verifies under real Fiat-Shamir challenges
    # size: every scalar 32 bytes, every point 48 bytes -- counts from a tape run of the same shape
    ref = wg.prove(claims[0], claims[1], d_inv, 2, [rng.next_bits(128) for _ in range(4000)])
    assert len(proofs[0]) == 32 * len(ref["msgs"]) + 48 * len(ref["points"])


def test_whole_proofs_on_concurrent_host_threads():
    """three host threads, each with its own stream, plan and PippengerWG (two shapes; ONE proving key on the device, read by all of
    them), run PippengerWG::new + Pippenger::prove at the same time, four proofs each: every proof -- transcript scalars, G1 points,
    pairing pair -- equals the one the same inputs gave on a single thread, and A = tau B.  The G1 engine's scratch and the
    fixed-base registry are shared state behind locks (one G1 call at a time per device), the sumcheck state is per thread
    (include/gkrmsm.h "Threads")."""
    import threading
    import torch
    shapes = [(9, 4, 32, 0), (8, 4, 16, 1), (9, 4, 32, 0)]
    nv_max = max(x + c for x, _, _, c in shapes)
    rng = F.SplitMix64(777)
    tau = rng.next_fr()
    d_basis = H.g1_mock_srs(tau, (2 << nv_max) - 1, G.GEN)
    H.g1_fixed_base_register(d_basis, (2 << nv_max) - 1)
    torch.cuda.synchronize()
    inputs = {}
    for x_log, d_log, nbits, clm in set(shapes):
        y_size = (nbits + d_log - 1) // d_log
        y_log = (y_size - 1).bit_length()
        n = 1 << x_log
        inputs[(x_log, d_log, nbits, clm)] = dict(
            pts=codec.points_to_mont(F.random_points(n, 40 + x_log)), sc=codec.ints_to_limbs(F.random_scalars(n, nbits, 41 + clm)),
            y_size=y_size, y_log=y_log, r=[rng.next_fr() for _ in range(y_log)], tape=[rng.next_bits(128) for _ in range(6000)])

    def one_proof(shape, reps):
        x_log, d_log, nbits, clm = shape
        c = inputs[shape]
        d_pts = H.to_dev(c["pts"])
        plan = H.MsmPlan(x_log, d_log, c["y_size"])
        d_sc = H.to_dev(c["sc"])
        d_inv = H.knuckles_setup(2, x_log + clm)
        outs = []
        for _ in range(reps):
            plan.run(d_pts, d_sc)
            wg = H.PippengerWG(plan, d_pts, c["y_log"], clm, d_basis)
            claims = GK.pippenger_claims(wg.dense_output(), c["r"])
            outs.append(wg.prove(claims[0], claims[1], d_inv, 2, c["tape"]))
            wg.close()
        plan.close()
        return outs

    alone = {s: one_proof(s, 1)[0] for s in set(shapes)}
    for s, res in alone.items():
        assert res["pair"][0] == G.mul(res["pair"][1], tau), "shape %r: the single-thread proof does not verify" % (s,)
    bar = threading.Barrier(len(shapes))
    errors = []

    def prover(k):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                bar.wait(timeout=120)
                for j, res in enumerate(one_proof(shapes[k], 4)):
                    want = alone[shapes[k]]
                    for key in ("msgs", "points", "pair", "tape_used", "rounds"):
                        assert res[key] == want[key], "thread %d proof %d: %s differ from the single-thread proof" % (k, j, key)
        except Exception as e:  # noqa: BLE001 - reported to the test thread
            errors.append("thread %d: %r" % (k, e))
            try:
                bar.abort()
            except Exception:
                pass
    ths = [threading.Thread(target=prover, args=(k,)) for k in range(len(shapes))]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=600)
    torch.cuda.synchronize()
    H.g1_fixed_base_release(d_basis)
    assert not errors, errors
