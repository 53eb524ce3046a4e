"""The reference's identity tests replayed on the HIP path (through the C ABI), at the reference's literal shapes:

  check_univars  sumchecks/dense_eq.rs:258-344   the device's DenseDeg2 object against the NAIVE object (oracle/pyref/naive.py =
                                                  ExampleSumcheckObjectSO, sumcheck.rs:132-235): num_vars 6, 100 iterations
  check_univars  sumchecks/vecvec_eq.rs:511-600  the device's VecVecDeg2 object against the naive one: num_vars 6,
                                                  num_vertical_vars in {0, 1, 3} x {Full, Rows, Nothing}
  test_pippenger pippenger.rs:621-645            d_logsize 6, num_bits 128, x_logsize 10, commitment_log_multiplicity 0, transcript
                                                  label b"fgstglsp": run_pippenger on the device, verify_pippenger by the library
                                                  verifier + pairing, and the claimed output IS the MSM (expected_msm)
(pushforward_works, pushforward.rs:1050-1189, at its literal x_logsize 10 / y_size 5 / d_logsize 8 is a case of
tests/test_pushforward_gpu.py; logup_maincycle_works, logup_mainphase.rs:278-338, has no entry point of its own in the ABI -- the
main phase runs inside gm_pushforward_prove and is compared message by message there.)"""
import pytest

from gkr_msm_amd import ffi, harness as H
from pyref import algfn as A
from pyref import field as F
from pyref import naive as N

from ref_identity_common import dense_rand_points, vecvec_rand_points, vecvec_py
from test_ref_identity_cpu import _vec_claims

pytestmark = pytest.mark.gpu
P = F.P
PROJ_L1 = ffi.make_fn((4, 1))


def _rounds(gpu, naive, num_vars, rng):
    for _ in range(num_vars):
        a, b = gpu.unipoly(), naive.unipoly()
        for x in (0, 1, 2, 3):
            assert N.evaluate(a, x) == N.evaluate(b, x)
        assert list(a) == list(b)
        t = rng.next_fr()
        gpu.bind(t)
        naive.bind(t)
        assert naive.claim() == gpu.claim()


def test_dense_check_univars_on_the_device():
    rng = F.SplitMix64(0xD15F)
    num_vars, gamma, f = 6, 2, A.PROJ_L1
    for _ in range(100):
        cols = dense_rand_points(rng, num_vars)
        data = cols + [list(c) for c in cols]
        point = [rng.next_fr() for _ in range(num_vars)]
        eq = N.eq_table(point)
        naive = N.ExampleSumcheckObjectSO(data + [eq], N.GammaEq(f, gamma), num_vars)
        gpu = H.Sumcheckable.dense_deg2(PROJ_L1, num_vars, H.cols_to_dev(data), point, gamma, _vec_claims(f, data, eq))
        assert gpu.claim() == naive.claim()
        _rounds(gpu, naive, num_vars, rng)
        assert gpu.final_evals() == naive.final_evals()[:-1]


@pytest.mark.parametrize("num_vertical_vars", [0, 1, 3])
@pytest.mark.parametrize("denseness,iters", [("full", 30), ("rows", 30), ("nothing", 100)])
def test_vecvec_check_univars_on_the_device(num_vertical_vars, denseness, iters):
    rng = F.SplitMix64(0xBEC1 + 16 * num_vertical_vars + len(denseness))
    num_vars, gamma, f = 6, 2, A.PROJ_L1
    row_log = num_vars - num_vertical_vars
    for _ in range(iters):
        data3, pads = vecvec_rand_points(rng, row_log, num_vertical_vars, denseness)
        if max(len(r) for r in data3[0]) < 2:
            continue                                   # a lone 1-cell row: log_2(0) in the reference too (vecvec.rs:86)
        py = vecvec_py(data3, pads, row_log, num_vertical_vars)
        dense = [p.to_dense() for p in py] * 2
        point = [rng.next_fr() for _ in range(num_vars)]
        eq = N.eq_table(point)
        naive = N.ExampleSumcheckObjectSO(dense + [eq], N.GammaEq(f, gamma), num_vars)
        vv = H.VV.from_host(data3 * 2, [p[0] for p in pads] * 2, [p[1] for p in pads] * 2, row_log, num_vertical_vars)
        gpu = H.Sumcheckable.vecvec_deg2(PROJ_L1, vv, point, gamma, _vec_claims(f, dense, eq))
        assert gpu.claim() == naive.claim()
        _rounds(gpu, naive, num_vars, rng)
        assert gpu.final_evals() == naive.final_evals()


def test_pippenger_at_the_reference_tests_literal_shape():
    """pippenger.rs:621-645"""
    import numpy as np
    from gkr_msm_amd import codec, verifier as VF
    from pyref import g1 as G
    from pyref import pairing as PR
    from test_verifier_gpu import _prove_merlin, _setup
    x_log, d_log, nbits, clm = 10, 6, 128, 0
    s = _setup(x_log, d_log, nbits, clm, 61, device_srs=True)
    proof, pair = _prove_merlin(s, b"fgstglsp")
    got = VF.pippenger_verify_merlin(*s["shape"], s["claims"][0], s["claims"][1], G.GEN, 2, b"fgstglsp", proof)
    assert got == pair and VF.kzg_verify_pair(got, PR.G2_GEN, PR.g2_mul(PR.G2_GEN, s["tau"]))
    # expected_msm: sum_i coef_i * point_i by plain double-and-add on the host, against the group element the proof is about
    plan, d_pts, d_sc, _ = s["keep"]
    pts = codec.from_mont_limbs(H.to_host(d_pts).reshape(-1, 4))
    sc = codec.limbs_to_ints(H.to_host(d_sc).reshape(-1, 4)) if hasattr(codec, "limbs_to_ints") else [
        int(v[0]) | int(v[1]) << 64 | int(v[2]) << 128 | int(v[3]) << 192 for v in H.to_host(d_sc).reshape(-1, 4)]
    acc = (0, 1)
    for i, k in enumerate(sc):
        acc = F.te_add_affine(acc, F.te_mul_affine((pts[2 * i], pts[2 * i + 1]), k))
    assert H.combine_host(plan.window_points_raw(), d_log) == acc
