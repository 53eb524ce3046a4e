"""GPU parity for the whole "prove image part" (triangle GKR, two splits, bintree GKR, GlueSplit) through the C ABI
driver vs the Python oracle with the same challenge tape: every prover message, the final claims, and (Pattern A,
pippenger.rs:621-645 / pippenger_ending.rs:176-275) final claims == the image polynomials at the final point."""
import pytest

from gkr_msm_amd import codec, harness as H
from pyref import field as F
from pyref import gkr as G
from pyref import polys as PL
from pyref.sumcheck import TapeTranscript

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("x_log,d_log,nbits", [(4, 2, 12), (5, 3, 16), (6, 2, 10), (3, 3, 24), (8, 4, 32), (7, 6, 128),
                                               (9, 8, 64)])
def test_prove_image_part_matches_oracle(x_log, d_log, nbits):
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
    tape = [rng.next_bits(128) for _ in range(4000)]
    tr = TapeTranscript(tape)
    fin = G.prove_image_part(tr, y_log, d_log, x_log, claims, wg)
    exp_msgs = [v for m in tr.msgs for v in m]

    d_pts = H.to_dev(codec.points_to_mont(pts))
    d_sc = H.to_dev(codec.ints_to_limbs(sc))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    w = H.PipWitness(plan, d_pts, y_log)
    g_out, g_bs = w.outputs()
    assert g_out == out
    assert g_bs == wg.bucket_sums
    res = w.prove_image_part(claims[0], claims[1], tape)
    assert res["tape_used"] == tr.pos
    assert res["msgs"] == exp_msgs
    assert res["point"] == fin[0] and res["evs"] == fin[1]
    # Pattern A: the final claims are evaluations of the image polynomials (x, y, z)
    for i in range(3):
        assert PL.evaluate_poly(image[i].to_dense(), res["point"]) == res["evs"][i]


def test_live_transcript_matches_tape():
    """gm_pip_prove_image_part_tr: the callbacks see exactly the tape run's messages, in order, and a challenge that
    depends on everything written so far (a hash chain, as a real Fiat-Shamir transcript) is honoured round by round."""
    import hashlib
    x_log, d_log, nbits = 6, 3, 18
    y_size = (nbits + d_log - 1) // d_log
    y_log = PL.log2_exact(y_size)
    n = 1 << x_log
    pts = F.random_points(n, 21)
    sc = F.random_scalars(n, nbits, 22)
    d_pts = H.to_dev(codec.points_to_mont(pts))
    d_sc = H.to_dev(codec.ints_to_limbs(sc))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    w = H.PipWitness(plan, d_pts, y_log)
    outs, _ = w.outputs()
    rng = F.SplitMix64(5)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = G.pippenger_claims(outs, r)
    state = hashlib.sha256(b"live")
    drawn = []

    def on_write(vals):
        for v in vals:
            state.update(v.to_bytes(32, "little"))

    def draw():
        c = int.from_bytes(state.digest()[:16], "little")
        state.update(b"c")
        drawn.append(c)
        return c
    live = H.LiveTranscript(draw, on_write)
    res = H.prove_image_part_tr(w, claims[0], claims[1], live)
    assert res["n_challenges"] == len(drawn) == live.n_challenges
    # replay the drawn challenges as a tape: same messages, same final claim
    ref = w.prove_image_part(claims[0], claims[1], drawn)
    assert [v for m in live.writes for v in m] == ref["msgs"]
    assert (res["point"], res["evs"], res["rounds"]) == (ref["point"], ref["evs"], ref["rounds"])
    # a failing callback aborts with an error code instead of a wrong proof
    bad = H.LiveTranscript(lambda: None)
    with pytest.raises(Exception):
        H.prove_image_part_tr(w, claims[0], claims[1], bad)


def test_slow_and_failing_transcripts_with_kernels_waiting_on_the_device():
    """Small rounds keep kernels waiting on the device for the next challenge (gate kernels, the persistent tail kernel).
    A transcript that takes its time (sleeps) must not time them out, and one that fails in the middle -- at every kind of
    round -- must leave nothing waiting: the next proof on the same witness is the normal one."""
    import time
    x_log, d_log, nbits = 11, 4, 32     # dense stages of 7 variables (tail only) and sparse rounds above and below the split size
    y_size = (nbits + d_log - 1) // d_log
    y_log = PL.log2_exact(y_size)
    n = 1 << x_log
    pts = F.random_points(n, 31)
    sc = F.random_scalars(n, nbits, 32)
    d_pts = H.to_dev(codec.points_to_mont(pts))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(sc)))
    w = H.PipWitness(plan, d_pts, y_log)
    outs, _ = w.outputs()
    rng = F.SplitMix64(9)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = G.pippenger_claims(outs, r)
    tape = [rng.next_bits(128) for _ in range(3000)]
    ref = w.prove_image_part(claims[0], claims[1], tape)
    used = ref["tape_used"]

    def make(fail_at=None, sleep_every=0):
        it = iter(tape)
        count = [0]

        def draw():
            count[0] += 1
            if fail_at is not None and count[0] > fail_at:
                return None
            if sleep_every and count[0] % sleep_every == 0:
                time.sleep(0.02)
            return next(it)
        return H.LiveTranscript(draw)
    slow = H.prove_image_part_tr(w, claims[0], claims[1], make(sleep_every=37))
    assert (slow["point"], slow["evs"]) == (ref["point"], ref["evs"])
    for fail_at in (3, used // 5, used // 2, used - 4):
        with pytest.raises(Exception):
            H.prove_image_part_tr(w, claims[0], claims[1], make(fail_at=fail_at))
        again = w.prove_image_part(claims[0], claims[1], tape)
        assert again["msgs"] == ref["msgs"] and again["evs"] == ref["evs"]


def test_a_timed_out_wait_does_not_poison_later_proofs():
    """gm_set_wait_timeout_ms: with a 100 ms bound a transcript that sleeps longer makes the waiting kernels (gate / persistent
    tail) give up -- the proof call fails with an error, nothing hangs -- and the NEXT proof on the same host thread, with the
    default bound restored, is the normal one (the per-thread pinned stages carry no stale time-out flag)"""
    import time
    from gkr_msm_amd import ffi
    x_log, d_log, nbits = 11, 4, 32
    y_size = (nbits + d_log - 1) // d_log
    y_log = PL.log2_exact(y_size)
    n = 1 << x_log
    pts = F.random_points(n, 31)
    sc = F.random_scalars(n, nbits, 32)
    d_pts = H.to_dev(codec.points_to_mont(pts))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(sc)))
    w = H.PipWitness(plan, d_pts, y_log)
    outs, _ = w.outputs()
    rng = F.SplitMix64(9)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = G.pippenger_claims(outs, r)
    tape = [rng.next_bits(128) for _ in range(3000)]
    ref = w.prove_image_part(claims[0], claims[1], tape)
    L = ffi.lib()
    timeouts = []
    # not every draw has a kernel waiting behind it (gammas, split challenges, the rounds the host finishes itself): walk over
    # the draws until three waits have timed out
    for sleep_at in range(2, 60):
        if len(timeouts) >= 3:
            break
        ffi.check(L.gm_set_wait_timeout_ms(100))
        it = iter(tape)
        count = [0]

        def draw():
            count[0] += 1
            if count[0] == sleep_at:
                time.sleep(0.35)
            return next(it)
        try:   # a draw with no kernel waiting behind it just takes longer
            res = H.prove_image_part_tr(w, claims[0], claims[1], H.LiveTranscript(draw))
            assert (res["point"], res["evs"]) == (ref["point"], ref["evs"])
        except ffi.GmError as e:
            assert "timed out" in str(e) or "did not arrive" in str(e), str(e)
            timeouts.append(sleep_at)
        finally:
            ffi.check(L.gm_set_wait_timeout_ms(0))
        again = w.prove_image_part(claims[0], claims[1], tape)
        assert again["msgs"] == ref["msgs"] and again["evs"] == ref["evs"]
    assert timeouts, "no wait timed out: the bound is not honoured"


def test_provers_on_concurrent_host_threads():
    """four host threads, each with its own stream, plan and witness (two shapes, so that large and small stage kernels meet on
    the device), prove at the same time: every proof equals the one the same thread made alone, which equals the oracle's for
    that shape (include/gkrmsm.h "Threads"; the reference's provers run under rayon in the same way)"""
    import threading
    import torch
    shapes = [(9, 8, 64), (8, 4, 32), (9, 8, 64), (7, 6, 128)]
    ref = {}
    for x_log, d_log, nbits in set(shapes):
        y_size = (nbits + d_log - 1) // d_log
        y_log = PL.log2_exact(y_size)
        n = 1 << x_log
        pts = F.random_points(n, 7 + x_log)
        sc = F.random_scalars(n, nbits, 70 + d_log)
        image, digits, counter, wg = G.pippenger_witness(pts, sc, y_size, y_log, d_log, x_log)
        out = G.pippenger_dense_output(wg, y_log, d_log)
        rng = F.SplitMix64(99)
        r = [rng.next_fr() for _ in range(y_log)]
        claims = G.pippenger_claims(out, r)
        tape = [rng.next_bits(128) for _ in range(4000)]
        tr = TapeTranscript(tape)
        G.prove_image_part(tr, y_log, d_log, x_log, claims, wg)
        ref[(x_log, d_log, nbits)] = dict(pts=codec.points_to_mont(pts), sc=codec.ints_to_limbs(sc), claims=claims, tape=tape,
                                          msgs=[v for m in tr.msgs for v in m], y_size=y_size, y_log=y_log)
    bar = threading.Barrier(len(shapes))
    errors = []

    def prover(k):
        try:
            x_log, d_log, nbits = shapes[k]
            c = ref[shapes[k]]
            with torch.cuda.stream(torch.cuda.Stream()):
                d_pts, d_sc = H.to_dev(c["pts"]), H.to_dev(c["sc"])
                plan = H.MsmPlan(x_log, d_log, c["y_size"])
                plan.run(d_pts, d_sc)
                w = H.PipWitness(plan, d_pts, c["y_log"])
                bar.wait(timeout=120)
                for _ in range(6):
                    res = w.prove_image_part(c["claims"][0], c["claims"][1], c["tape"])
                    assert res["msgs"] == c["msgs"], "thread %d: proof differs from the oracle's" % k
                w.close()
                plan.close()
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
        th.join(timeout=300)
    assert not errors, errors


def test_stage_launch_denied_falls_back_to_ordinary_rounds():
    """a stage launch that may only TRY for its share of the device (one of this thread's own gates is waiting in the stream) and
    finds it busy leaves the layer on ordinary round kernels: GM_STAGE_FORCE_BUSY=1 makes every such launch find the device busy
    (read once per process, hence the child process); the proofs must still equal the oracle's"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_STAGE_FORCE_BUSY="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_prover_gpu.py"), "-x", "-q", "-k",
                          "matches_oracle and (9-8-64 or 7-6-128 or 5-3-16)"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "3 passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_stage_launch_not_resident_falls_back_to_ordinary_rounds():
    """a k_stage launch whose grid does not become resident as a whole (kernels of other streams / processes hold compute units)
    is abandoned at its residency barrier and the layer's rounds run as ordinary kernels: GM_STAGE_FORCE_NONRESIDENT=1 makes every
    launch's barrier wait for one block more than the grid has (child process: read once); the proofs must still equal the oracle's"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_STAGE_FORCE_NONRESIDENT="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_prover_gpu.py"), "-x", "-q", "-k",
                          "matches_oracle and (9-8-64 or 7-6-128 or 5-3-16)"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "3 passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
