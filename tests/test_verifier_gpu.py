"""Proofs made on the GPU (gm_pippenger_wg_create + gm_pippenger_prove) go through the library's host verifier
(gm_pippenger_verify = Pippenger::verify, pippenger.rs:296-406) and the real pairing check (gm_kzg_verify_pair): accepted as
made, rejected when altered.  Both transcript forms: recorded messages + tape, and the built-in merlin transcript over proof
bytes (prover mode -> bytes -> verifier mode)."""
import ctypes as C

import numpy as np
import pytest

from gkr_msm_amd import codec, ffi, harness as H, verifier as VF
from pyref import field as F
from pyref import g1 as G
from pyref import gkr as GK
from pyref import pairing as PR

pytestmark = pytest.mark.gpu


def _setup(x_log, d_log, nbits, clm, seed, device_srs=False, scalars_u64x4=None):
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    rng = F.SplitMix64(seed)
    nv = x_log + clm
    tau = rng.next_fr()
    if device_srs:
        d_pts = H.dev_empty(n * 8)
        ffi.check(ffi.lib().gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 77, H.cur_stream()))
        sc = np.random.default_rng(seed).integers(0, 2**63, size=(n, 4), dtype=np.uint64)
        sc[:, 3] &= np.uint64((1 << 60) - 1)
        for w in range(4):   # nbits bits
            lo = max(0, min(64, nbits - 64 * w))
            sc[:, w] &= np.uint64((1 << lo) - 1) if lo < 64 else np.uint64(2**64 - 1)
        if scalars_u64x4 is not None:
            sc = np.ascontiguousarray(scalars_u64x4, dtype=np.uint64).reshape(n, 4)
        d_sc = H.to_dev(sc)
        d_basis = H.g1_mock_srs(tau, (2 << nv) - 1, G.GEN)
    else:
        pts = F.random_points(n, 2)
        d_pts = H.to_dev(codec.points_to_mont(pts))
        d_sc = H.to_dev(codec.ints_to_limbs(F.random_scalars(n, nbits, 3)))
        basis, cur = [], G.GEN
        for _ in range((2 << nv) - 1):
            basis.append(cur)
            cur = G.mul(cur, tau)
        d_basis = H.g1_aff_dev(basis)
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    wg = H.PippengerWG(plan, d_pts, y_log, clm, d_basis)
    out = wg.dense_output()
    r = [rng.next_fr() for _ in range(y_log)]
    claims = GK.pippenger_claims(out, r)
    return dict(shape=(x_log, d_log, y_size, y_log, clm), wg=wg, claims=claims, d_inv=H.knuckles_setup(2, nv), tau=tau, rng=rng,
                keep=(plan, d_pts, d_sc, d_basis))


@pytest.mark.parametrize("x_log,d_log,nbits,clm", [(3, 2, 8, 0), (4, 3, 12, 2), (6, 3, 16, 1)])
def test_gpu_proof_verifies_tape_form(x_log, d_log, nbits, clm):
    s = _setup(x_log, d_log, nbits, clm, 5 + x_log)
    tape = [s["rng"].next_bits(128) for _ in range(6000)]
    res = s["wg"].prove(s["claims"][0], s["claims"][1], s["d_inv"], 2, tape)
    used = tape[: res["tape_used"]]
    got = VF.pippenger_verify(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, res["msgs"], res["points"], used)
    assert got["pair"] == res["pair"] and got["tape_used"] == res["tape_used"]
    assert VF.kzg_verify_pair(got["pair"], PR.G2_GEN, PR.g2_mul(PR.G2_GEN, s["tau"]))
    assert not VF.kzg_verify_pair(got["pair"], PR.G2_GEN, PR.g2_mul(PR.G2_GEN, s["tau"] + 1))
    for idx in (0, len(res["msgs"]) // 2, len(res["msgs"]) - 1):
        bad = list(res["msgs"])
        bad[idx] = (bad[idx] + 1) % F.P
        with pytest.raises(VF.Rejected):
            VF.pippenger_verify(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, bad, res["points"], used)
    with pytest.raises(VF.Rejected):    # a different claimed MSM output
        VF.pippenger_verify(*s["shape"], s["claims"][0], [(s["claims"][1][0] + 1) % F.P] + list(s["claims"][1][1:]), G.GEN, 2,
                            res["msgs"], res["points"], used)


def _prove_merlin(s, label):
    L = ffi.lib()
    cp, ce, kk = H.fr_arg(s["claims"][0]), H.fr_arg(s["claims"][1]), H.fr_arg([2])
    h = C.c_void_p()
    ffi.check(L.gm_merlin_create(label, len(label), C.byref(h)))
    tr = ffi.GmTranscript()
    ffi.check(L.gm_merlin_transcript(h, C.byref(tr)))
    pair = np.zeros(24, dtype=np.uint64)
    used, rounds = C.c_uint64(), C.c_uint64()
    ffi.check(L.gm_pippenger_prove_tr(s["wg"].h, cp.ctypes.data, ce.ctypes.data, C.c_void_p(s["d_inv"].data_ptr()), kk.ctypes.data,
                                      C.byref(tr), pair.ctypes.data, C.byref(used), C.byref(rounds)))
    pp, pn = C.c_void_p(), C.c_uint64()
    ffi.check(L.gm_merlin_proof(h, C.byref(pp), C.byref(pn)))
    proof = C.string_at(pp, pn.value)
    L.gm_merlin_destroy(h)
    return proof, tuple(codec.g1_aff_from_limbs(pair))


def test_gpu_proof_bytes_verify_through_the_merlin_transcript():
    s = _setup(4, 2, 8, 1, 41)
    label = b"pippenger-gpu"
    proof, pair = _prove_merlin(s, label)
    got = VF.pippenger_verify_merlin(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, label, proof)
    assert got == pair
    assert VF.kzg_verify_pair(got, PR.G2_GEN, PR.g2_mul(PR.G2_GEN, s["tau"]))
    with pytest.raises(VF.Rejected):   # another domain separator: other challenges
        VF.pippenger_verify_merlin(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, b"pippenger-cpu", proof)
    for pos in (0, 48 * 3 + 5, len(proof) // 2, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        try:
            alt = VF.pippenger_verify_merlin(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, label, bytes(bad))
        except VF.Rejected:
            continue
        # a flipped commitment byte can still decode to a curve point the algebraic checks never open: the pairing catches it
        assert not VF.kzg_verify_pair(alt, PR.G2_GEN, PR.g2_mul(PR.G2_GEN, s["tau"]))
    with pytest.raises(VF.Rejected):
        VF.pippenger_verify_merlin(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, label, proof[:-32])
    with pytest.raises(VF.Rejected):
        VF.pippenger_verify_merlin(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, label, proof + b"\0" * 32)


def test_mid_size_gpu_proof_verifies():
    """x_logsize = 12, 64-bit scalars, device-generated inputs and SRS: beyond what the Python oracle prover reaches"""
    s = _setup(12, 4, 64, 1, 9, device_srs=True)
    proof, pair = _prove_merlin(s, b"mid")
    got = VF.pippenger_verify_merlin(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, b"mid", proof)
    assert got == pair
    assert VF.kzg_verify_pair(got, PR.G2_GEN, PR.g2_mul(PR.G2_GEN, s["tau"]))


def test_c_example_proves_and_verifies():
    """examples/pippenger.c: the reference's example flow (build data, run_pippenger, verify_pippenger) written in plain C against
    include/gkrmsm.h -- real merlin Fiat-Shamir, proof bytes, host verifier, pairing; exit code 0 iff the proof verifies and a
    tampered one does not"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "examples", "pippenger")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", root, "examples"])
    r = subprocess.run([exe, "--x-logsize", "9", "--d-logsize", "3", "--nbits", "24", "--commitment-log-multiplicity", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "proof verified" in r.stdout and "tampered proof rejected" in r.stdout


def test_gpu_prover_reproduces_the_committed_proof_fixture():
    """from the fixture's inputs, SRS (tau) and challenges the GPU prover writes exactly the recorded transcript"""
    from test_verifier_cpu import _load_proof_fixture
    p = _load_proof_fixture()
    x_log, d_log, y_size, y_log, clm = p["shape"]
    nv = x_log + clm
    basis, cur = [], G.GEN
    for _ in range((2 << nv) - 1):
        basis.append(cur)
        cur = G.mul(cur, p["tau"])
    d_pts = H.to_dev(codec.points_to_mont(p["pts"]))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(p["sc"])))
    wg = H.PippengerWG(plan, d_pts, y_log, clm, H.g1_aff_dev(basis))
    res = wg.prove(p["claims"][0], p["claims"][1], H.knuckles_setup(p["k"], nv), p["k"], p["tape"] + [0] * 8)
    assert res["msgs"] == p["scalars"] and res["points"] == p["points"] and res["pair"] == p["pair"]
    assert res["tape_used"] == len(p["tape"])
