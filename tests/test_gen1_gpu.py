"""GPU parity for the gen-1 prover gkr_msm_prove (gkr_msm_simple.rs:86-338, Fr part) through the C ABI vs the Python
oracle with the same challenge tape: outputs, every transcript message, final claim; Pattern A on the final claim."""
import numpy as np
import pytest
import torch

from gkr_msm_amd import codec, harness as H
from pyref import field as F
from pyref import gen1 as G1

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lp,lb", [(1, 1), (2, 1), (3, 2), (4, 3), (5, 4), (3, 6), (6, 2)])
def test_gkr_msm_prove_matches_oracle(lp, lb):
    pts = F.random_points(1 << lp, 3 + lp)
    rng = F.SplitMix64(40 + lb)
    bits = [[bool(rng.next() & 1) for _ in range(1 << lb)] for _ in range(1 << lp)]
    tape = [rng.next_fr() for _ in range(4000)]
    claim, out, tr = G1.gkr_msm_prove(bits, pts, lp, lb, tape)
    d_pts = H.to_dev(codec.points_to_mont(pts))
    d_bits = torch.from_numpy(np.array([[1 if b else 0 for b in s] for s in bits], dtype=np.uint8).reshape(-1)).cuda()
    res = H.gkr_msm_prove(d_pts, d_bits, lp, lb, tape)
    assert res["output"] == out
    assert res["tape_used"] == tr.pos
    assert res["msgs"] == tr.msgs
    assert res["point"] == claim[0] and res["evs"] == claim[1]
    base = G1.base_layer(bits, pts, lp, lb)
    assert [G1.evaluate(b, res["point"]) for b in base] == res["evs"]


@pytest.mark.parametrize("kind", ["zeros", "ones", "identities"])
def test_gkr_msm_prove_on_degenerate_inputs(kind):
    """all-zero and all-one bit columns (every scalar 0 / 2^(2^lb) - 1) and identity points: the layer functions see (0, 1, 1) cells and
    doublings of equal operands; outputs and every message as the oracle's"""
    lp, lb = 3, 2
    pts = F.random_points(1 << lp, 5)
    rng = F.SplitMix64(41)
    bits = [[bool(rng.next() & 1) for _ in range(1 << lb)] for _ in range(1 << lp)]
    if kind == "zeros":
        bits = [[False] * (1 << lb) for _ in range(1 << lp)]
    elif kind == "ones":
        bits = [[True] * (1 << lb) for _ in range(1 << lp)]
    else:
        pts = [(0, 1)] * (1 << lp)
    tape = [rng.next_fr() for _ in range(4000)]
    claim, out, tr = G1.gkr_msm_prove(bits, pts, lp, lb, tape)
    d_pts = H.to_dev(codec.points_to_mont(pts))
    d_bits = torch.from_numpy(np.array([[1 if b else 0 for b in s] for s in bits], dtype=np.uint8).reshape(-1)).cuda()
    res = H.gkr_msm_prove(d_pts, d_bits, lp, lb, tape)
    assert res["output"] == out
    assert res["msgs"] == tr.msgs
    assert res["point"] == claim[0] and res["evs"] == claim[1]


def test_gen1_live_transcript_matches_tape():
    lp, lb = 4, 2
    pts = F.random_points(1 << lp, 9)
    rng = F.SplitMix64(77)
    bits = [[bool(rng.next() & 1) for _ in range(1 << lb)] for _ in range(1 << lp)]
    d_pts = H.to_dev(codec.points_to_mont(pts))
    d_bits = torch.from_numpy(np.array([[1 if b else 0 for b in s] for s in bits], dtype=np.uint8).reshape(-1)).cuda()
    acc = [12345]
    drawn = []

    def on_write(vals):
        for v in vals:
            acc[0] = (acc[0] * 1000003 + v) % F.P

    def draw():
        acc[0] = (acc[0] * 7 + 1) % F.P
        drawn.append(acc[0])
        return acc[0]
    live = H.LiveTranscript(draw, on_write)
    res = H.gkr_msm_prove_tr(d_pts, d_bits, lp, lb, live)
    ref = H.gkr_msm_prove(d_pts, d_bits, lp, lb, drawn)
    assert [v for m in live.writes for v in m] == ref["msgs"]
    assert (res["output"], res["point"], res["evs"]) == (ref["output"], ref["point"], ref["evs"])
    assert res["n_challenges"] == len(drawn) == ref["tape_used"]


def test_gpu_gen1_stream_verifies():
    """the library's gen-1 verifier (host) accepts the GPU prover's transcript stream at a size beyond the Python oracle and ends
    on the prover's final claim; a flipped message is rejected"""
    from gkr_msm_amd import verifier as VF
    lp, lb = 10, 4
    n = 1 << lp
    d_pts = H.dev_empty(n * 8)
    import ctypes as C
    from gkr_msm_amd import ffi
    ffi.check(ffi.lib().gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 5, H.cur_stream()))
    d_bits = torch.from_numpy(np.random.default_rng(3).integers(0, 2, size=n << lb, dtype=np.uint8)).cuda()
    rng = F.SplitMix64(8)
    tape = [rng.next_fr() for _ in range(1000)]
    res = H.gkr_msm_prove(d_pts, d_bits, lp, lb, tape)
    got = VF.gkr_msm_verify(lp, lb, res["msgs"], tape[: res["tape_used"]])
    assert got["point"] == res["point"] and got["evs"] == res["evs"] and got["rounds"] == res["rounds"]
    bad = list(res["msgs"])
    bad[len(bad) // 3] = (bad[len(bad) // 3] + 1) % F.P
    with pytest.raises(VF.Rejected):
        VF.gkr_msm_verify(lp, lb, bad, tape[: res["tape_used"]])


def test_c_example_gen1():
    """examples/gkr_msm_simple.c: the reference test's flow (commitments, gkr_msm_prove) plus the verifier, in plain C"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "examples", "gkr_msm_simple")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", root, "examples"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "transcript verified" in r.stdout and "tampered transcript rejected" in r.stdout
