"""GPU parity for the polynomial layer and the sumcheck objects, through the C ABI, against the Python oracle.
Mirrors the reference's Pattern-B tests (optimised vs naive round polynomials, dense_eq.rs:258-344,
vecvec_eq.rs:511-600) and Pattern-A (prover claims == direct evaluation).  Bit-exact."""
import pytest

from gkr_msm_amd import ffi, harness as H
from pyref import algfn as A
from pyref import field as F
from pyref import polys as PL
from pyref import sumcheck as SC

pytestmark = pytest.mark.gpu

FN = {
    "aff_l1": (ffi.make_fn((1, 1)), A.AFF_L1), "aff_l2": (ffi.make_fn((2, 1)), A.AFF_L2),
    "aff_l3": (ffi.make_fn((3, 1)), A.AFF_L3), "proj_l1": (ffi.make_fn((4, 1)), A.PROJ_L1),
    "proj_l2": (ffi.make_fn((5, 1)), A.PROJ_L2), "proj_l3": (ffi.make_fn((6, 1)), A.PROJ_L3),
    "aff_l1_bc": (ffi.make_fn((1, 1), (9, 2)), A.StackedAlgFn(A.AFF_L1, A.RepeatedAlgFn(A.BitCheckFn(), 2))),
    "tri_l1_r2": (ffi.make_fn((7, 1), (4, 2)), A.StackedAlgFn(A.TRI_L1, A.RepeatedAlgFn(A.PROJ_L1, 2))),
    "l2_r5": (ffi.make_fn((5, 5)), A.RepeatedAlgFn(A.PROJ_L2, 5)),
    "l3_r5": (ffi.make_fn((6, 5)), A.RepeatedAlgFn(A.PROJ_L3, 5)),
    "id3": (ffi.make_fn((8, 3)), A.IdAlgFn(3)),
    "id3x2": (ffi.make_fn((8, 6)), A.RepeatedAlgFn(A.IdAlgFn(3), 2)),
}


def rand_cols(rng, k, n):
    return [[rng.next_fr() for _ in range(n)] for _ in range(k)]


@pytest.mark.parametrize("name", ["aff_l1", "proj_l2", "tri_l1_r2", "l3_r5", "aff_l1_bc"])
def test_dense_map(name):
    fn, pyf = FN[name]
    rng = F.SplitMix64(len(name))
    cols = rand_cols(rng, pyf.n_ins, 70)   # not a power of two on purpose
    out = H.cols_to_host(H.dense_map(fn, H.cols_to_dev(cols), pyf.n_outs))
    assert out == PL.dense_algfn_map(cols, pyf)


@pytest.mark.parametrize("name,lo_bit,bundle", [("id3", 0, 3), ("id3", 3, 3), ("id3x2", 4, 3), ("l3_r5", 2, 3),
                                                ("proj_l3", 0, 3), ("l3_r5", 5, 3)])
def test_dense_map_split(name, lo_bit, bundle):
    fn, pyf = FN[name]
    rng = F.SplitMix64(17 + lo_bit)
    cols = rand_cols(rng, pyf.n_ins, 64)
    out = H.cols_to_host(H.dense_map_split(fn, H.cols_to_dev(cols), pyf.n_outs, lo_bit, bundle))
    assert out == PL.dense_algfn_map_split(cols, pyf, PL.LO(lo_bit), bundle)


def test_dense_bind_and_eq_table():
    rng = F.SplitMix64(5)
    cols = rand_cols(rng, 5, 128)
    t = rng.next_fr()
    assert H.cols_to_host(H.dense_bind(H.cols_to_dev(cols), t)) == [PL.bind_dense(c, t) for c in cols]
    for nv in (0, 1, 5, 9):
        pt = [rng.next_fr() for _ in range(nv)]
        m = rng.next_fr()
        assert H.cols_to_host([H.eq_table(m, pt)])[0] == PL.eq_poly_sequence_from_multiplier(m, pt)[-1]


def run_rounds(gpu, ref, num_rounds, rng):
    """drive both objects with the same challenges; compare every round polynomial and the final evaluations"""
    for rnd in range(num_rounds):
        assert gpu.unipoly() == ref.unipoly(), "round %d polynomial" % rnd
        t = rng.next_bits(128)
        gpu.bind(t)
        ref.bind(t)
    assert gpu.final_evals() == ref.final_evals()


@pytest.mark.parametrize("name,nv", [("proj_l1", 6), ("proj_l2", 5), ("proj_l3", 7), ("tri_l1_r2", 4), ("l2_r5", 3),
                                     ("l3_r5", 5), ("proj_l1", 1), ("aff_l1_bc", 4),
                                     # 9 variables: the whole object runs in the persistent tail kernel; 10 and 12: pre-enqueued
                                     # (gated) folds first, then the hand-over to the tail; tri_l1_r2 has 9 segments
                                     ("proj_l2", 9), ("aff_l1_bc", 10), ("proj_l3", 12), ("tri_l1_r2", 11)])
def test_dense_deg2_sumcheck_object(name, nv):
    fn, pyf = FN[name]
    rng = F.SplitMix64(100 + nv)
    cols = rand_cols(rng, pyf.n_ins, 1 << nv)
    point = [rng.next_fr() for _ in range(nv)]
    gamma = rng.next_bits(128)
    outs = PL.dense_algfn_map(cols, pyf)
    claims = [PL.evaluate_poly(o, point) for o in outs]
    ref = SC.DenseDeg2SumcheckObjectSO.rlc(cols, pyf, claims, point, gamma)
    gpu = H.Sumcheckable.dense_deg2(fn, nv, H.cols_to_dev(cols), point, gamma, claims)
    assert gpu.claim() == ref.claim
    run_rounds(gpu, ref, nv, rng)


def rand_vecvec(rng, k, row_log, col_log, mode):
    if mode == "full":
        nrows, lens = 1 << col_log, [1 << row_log] * (1 << col_log)
    elif mode == "rows":
        nrows = 1 << col_log
        lens = [rng.next() % ((1 << row_log) + 1) for _ in range(nrows)]
    else:
        nrows = 1 + rng.next() % (1 << col_log)
        lens = [rng.next() % ((1 << row_log) + 1) for _ in range(nrows)]
    if max(lens) < 2:
        lens[0] = 2
    data = [[[rng.next_fr() for _ in range(l)] for l in lens] for _ in range(k)]
    rpad = [rng.next_fr() for _ in range(k)]
    cpad = [rng.next_fr() for _ in range(k)]
    py = [PL.VecVec(data[c], rpad[c], cpad[c], row_log, col_log) for c in range(k)]
    gpu = H.VV.from_host(data, rpad, cpad, row_log, col_log)
    return py, gpu


@pytest.mark.parametrize("mode", ["full", "rows", "nothing"])
def test_vecvec_maps(mode):
    rng = F.SplitMix64(len(mode))
    py, gpu = rand_vecvec(rng, 6, 4, 3, mode)
    assert gpu.to_dense() == [p.to_dense() for p in py]
    fn, pyf = FN["proj_l1"]
    m_py, m_gpu = PL.vecvec_map(py, pyf), gpu.map(fn)
    assert m_gpu.to_dense() == [p.to_dense() for p in m_py]
    rows, rp, cp = m_gpu.rows()
    assert rows == [p.data for p in m_py] and rp == [p.row_pad for p in m_py] and cp == [p.col_pad for p in m_py]
    fn3, pyf3 = FN["id3x2"]
    s_py, s_gpu = PL.vecvec_map_split(py, pyf3, PL.LO(0), 3), gpu.map_split(fn3, 3)
    rows, rp, cp = s_gpu.rows()
    assert rows == [p.data for p in s_py] and rp == [p.row_pad for p in s_py] and cp == [p.col_pad for p in s_py]
    assert s_gpu.to_dense() == [p.to_dense() for p in s_py]
    # GlueSplit::witness shape: slices, different bundles, concat (splits.rs:172-176)
    g_py = PL.vecvec_map_split(py[0:2], A.IdAlgFn(2), PL.LO(0), 2) + PL.vecvec_map_split(py[2:3], A.IdAlgFn(1), PL.LO(0), 1)
    g_gpu = gpu.slice(0, 2).map_split(ffi.make_fn((8, 2)), 2).concat(gpu.slice(2, 1).map_split(ffi.make_fn((8, 1)), 1))
    assert g_gpu.to_dense() == [p.to_dense() for p in g_py]


@pytest.mark.parametrize("mode", ["full", "rows", "nothing"])
def test_vecvec_map_split_to_dense(mode):
    rng = F.SplitMix64(3 + len(mode))
    py, gpu = rand_vecvec(rng, 4, 1, 4, mode)
    fn, pyf = FN["proj_l3"]
    exp = PL.vecvec_map_split_to_dense(py, pyf, PL.LO(0), 3)
    assert H.cols_to_host(gpu.map_split_to_dense(fn, 3, pyf.n_outs)) == exp


@pytest.mark.parametrize("name,row_log,col_log,mode", [
    ("proj_l1", 4, 2, "full"), ("proj_l1", 4, 3, "rows"), ("proj_l1", 5, 2, "nothing"), ("aff_l1_bc", 3, 3, "rows"),
    ("aff_l2", 4, 2, "nothing"), ("aff_l3", 1, 3, "rows"), ("proj_l2", 2, 4, "nothing"), ("proj_l3", 6, 1, "rows"),
    ("proj_l1", 3, 0, "rows")])
def test_vecvec_deg2_sumcheck_object(name, row_log, col_log, mode):
    fn, pyf = FN[name]
    rng = F.SplitMix64(1000 + row_log * 10 + col_log)
    py, gpu_vv = rand_vecvec(rng, pyf.n_ins, row_log, col_log, mode)
    nv = row_log + col_log
    point = [rng.next_fr() for _ in range(nv)]
    gamma = rng.next_bits(128)
    outs = [p.to_dense() for p in PL.vecvec_map(py, pyf)]
    claims = [PL.evaluate_poly(o, point) for o in outs]
    ref = SC.VecVecDeg2SumcheckObjectSO.rlc(py, pyf, claims, point, col_log, gamma)
    gpu = H.Sumcheckable.vecvec_deg2(fn, gpu_vv, point, gamma, claims)
    assert gpu.claim() == ref.claim()
    chal = []
    for rnd in range(nv):
        assert gpu.unipoly() == ref.unipoly(), "round %d polynomial" % rnd
        t = rng.next_bits(128)
        chal.append(t)
        gpu.bind(t)
        ref.bind(t)
    fe = gpu.final_evals()
    assert fe == ref.final_evals()
    # Pattern A: final evaluations are the input polynomials at the challenge point (+ eq)
    r = list(reversed(chal))
    assert fe[:-1] == [PL.evaluate_poly(p.to_dense(), r) for p in py]


def test_generic_dense_sumcheck_object():
    rng = F.SplitMix64(9)
    nv = 6
    # Prod3 (pushforward.rs:27-49)
    cols = rand_cols(rng, 3, 1 << nv)
    claim = sum(a * b % F.P * c for a, b, c in zip(*cols)) % F.P
    ref = SC.DenseSumcheckObjectSO(cols, SC.Prod3Fn(), nv, claim)
    gpu = H.Sumcheckable.dense(1, None, nv, H.cols_to_dev(cols), 0, claim)
    run_rounds(gpu, ref, nv, rng)
    # EqWrapper(GammaWrapper(f))  (DenseEqSumcheckObject::rlc, sumcheck.rs:394-417)
    fn, pyf = FN["proj_l1"]
    cols = rand_cols(rng, pyf.n_ins, 1 << nv)
    point = [rng.next_fr() for _ in range(nv)]
    gamma = rng.next_bits(128)
    claims = [PL.evaluate_poly(o, point) for o in PL.dense_algfn_map(cols, pyf)]
    ref = SC.dense_eq_sumcheck_object(cols, pyf, point, claims, gamma)
    eq = PL.eq_poly_sequence_last(point)
    gpu = H.Sumcheckable.dense(0, fn, nv, H.cols_to_dev(cols + [eq]), gamma, ref.claim)
    run_rounds(gpu, ref, nv, rng)


def test_state_errors_mirror_reference_panics():
    fn, pyf = FN["proj_l1"]
    rng = F.SplitMix64(1)
    cols = rand_cols(rng, 6, 4)
    so = H.Sumcheckable.dense_deg2(fn, 2, H.cols_to_dev(cols), [1, 2], 3, [0, 0, 0, 0])
    with pytest.raises(ffi.GmError):
        so.bind(5)                      # bind before unipoly (dense_eq.rs:105 unwrap on None)
    so.unipoly()
    with pytest.raises(ffi.GmError):
        so.unipoly()                    # dense_eq.rs:109-111 panic!()
