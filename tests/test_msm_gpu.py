"""GPU parity: Pippenger MSM (digits, bucket scatter, bucket sums, bucket reduction, final point)
through the C ABI vs the Python big-int oracle on the same seeded inputs.  Bit-exact."""
import numpy as np
import pytest

from gkr_msm_amd import codec, harness
from pyref import field as F
from pyref import gkr as G
from pyref.polys import log2_exact

pytestmark = pytest.mark.gpu

SHAPES = [  # (x_logsize, d_logsize, nbits)
    (2, 2, 12), (4, 2, 12), (5, 3, 16), (6, 2, 10), (3, 3, 24), (8, 4, 32), (7, 6, 128), (9, 8, 64),
    (11, 5, 40), (10, 9, 27), (10, 10, 30),
]


@pytest.mark.parametrize("x_log,d_log,nbits", SHAPES)
def test_msm_matches_oracle(x_log, d_log, nbits):
    y_size = (nbits + d_log - 1) // d_log
    y_log = log2_exact(y_size)
    n = 1 << x_log
    pts = F.random_points(n, 100 + x_log)
    sc = F.random_scalars(n, nbits, 200 + d_log)
    if n >= 4:
        sc[1] = 0           # a zero scalar lands in bucket 0 of every window
        sc[2] = sc[3]       # collisions
    image, digits, counter, wg = G.pippenger_witness(pts, sc, y_size, y_log, d_log, x_log)
    out = G.pippenger_dense_output(wg, y_log, d_log)
    expect_pt = G.pippenger_final_point(out, d_log)

    d_pts = harness.to_dev(codec.points_to_mont(pts))
    d_sc = harness.to_dev(codec.ints_to_limbs(sc))
    plan = harness.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    dg, ct, rl = plan.digits_counter_rowlen()
    assert dg.tolist() == digits
    assert ct.tolist() == counter
    # bucket populations (un-padded): oracle rows are padded to even, recompute from digits
    exp_len = np.zeros(y_size << d_log, dtype=np.int64)
    for y in range(y_size):
        for x in range(n):
            exp_len[(y << d_log) + digits[y][x]] += 1
    assert rl.tolist() == exp_len.tolist()
    bs = plan.bucket_sums()
    nrows = y_size << d_log
    for c in range(3):
        assert bs[c] == wg.bucket_sums[c][:nrows], "bucket sums col %d" % c
    wp = plan.window_points()
    for c in range(3 * (d_log + 1)):
        assert wp[c] == out[c][:y_size], "window points col %d" % c
    # padded windows (y >= y_size) hold identities and do not change the Horner sum
    got = harness.combine_host(plan.window_points_raw(), d_log)
    assert got == expect_pt
    # independent check: naive double-and-add
    if n <= 64:
        acc = (0, 1)
        for p, s in zip(pts, sc):
            acc = F.te_add_affine(acc, F.te_mul_affine(p, s))
        assert got == acc
    plan.close()


@pytest.mark.parametrize("kind", ["all_equal", "negated_pairs", "identities", "max_scalars", "zero_scalars"])
def test_msm_on_degenerate_inputs(kind):
    """inputs that put P + P, P + (-P) and the identity into the bucket additions, and scalars whose every digit is 2^d - 1 / 0 (one
    bucket per window takes every point): the twisted-Edwards bucket sums are the reference's unified projective formulas in its
    association order, so the stored (X, Y, Z) must still equal the oracle's limb for limb, and the final point the naive sum"""
    x_log, d_log, nbits = 6, 3, 24
    y_size = nbits // d_log
    y_log = log2_exact(y_size)
    n = 1 << x_log
    pts = F.random_points(n, 555)
    sc = F.random_scalars(n, nbits, 556)
    if kind == "all_equal":
        pts = [pts[0]] * n
    elif kind == "negated_pairs":
        for i in range(0, n, 2):
            pts[i + 1] = F.te_neg(pts[i])
            sc[i + 1] = sc[i]               # same buckets: every bucket sum passes through P + (-P)
    elif kind == "identities":
        for i in range(0, n, 3):
            pts[i] = (0, 1)
    elif kind == "max_scalars":
        sc = [(1 << nbits) - 1] * n
    else:
        sc = [0] * n
    image, digits, counter, wg = G.pippenger_witness(pts, sc, y_size, y_log, d_log, x_log)
    out = G.pippenger_dense_output(wg, y_log, d_log)
    d_pts = harness.to_dev(codec.points_to_mont(pts))
    plan = harness.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, harness.to_dev(codec.ints_to_limbs(sc)))
    dg, ct, _ = plan.digits_counter_rowlen()
    assert dg.tolist() == digits and ct.tolist() == counter
    bs = plan.bucket_sums()
    for c in range(3):
        assert bs[c] == wg.bucket_sums[c][: y_size << d_log], "bucket sums col %d" % c
    wp = plan.window_points()
    for c in range(3 * (d_log + 1)):
        assert wp[c] == out[c][:y_size], "window points col %d" % c
    got = harness.combine_host(plan.window_points_raw(), d_log)
    assert got == G.pippenger_final_point(out, d_log)
    acc = (0, 1)
    for p_, s_ in zip(pts, sc):
        acc = F.te_add_affine(acc, F.te_mul_affine(p_, s_))
    assert got == acc
    # the image-part prover on the same witness (its layer functions see the same degenerate cells)
    from gkr_msm_amd.harness import PipWitness
    rng = F.SplitMix64(9)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = G.pippenger_claims(out, r)
    tape = [rng.next_bits(128) for _ in range(4000)]
    from pyref.sumcheck import TapeTranscript
    tr = TapeTranscript(tape)
    G.prove_image_part(tr, y_log, d_log, x_log, claims, wg)
    w = PipWitness(plan, d_pts, y_log)
    res = w.prove_image_part(claims[0], claims[1], tape)
    assert res["msgs"] == [v for m in tr.msgs for v in m]
    w.close()
    plan.close()


def test_window_sharding_matches_full():
    x_log, d_log, nbits = 8, 4, 32
    y_size = 8
    n = 1 << x_log
    pts = F.random_points(n, 1)
    sc = F.random_scalars(n, nbits, 2)
    d_pts = harness.to_dev(codec.points_to_mont(pts))
    d_sc = harness.to_dev(codec.ints_to_limbs(sc))
    full = harness.MsmPlan(x_log, d_log, y_size)
    full.run(d_pts, d_sc)
    ref = full.window_points_raw()
    parts = []
    for (a, b) in [(0, 3), (3, 4), (4, 8)]:
        p = harness.MsmPlan(x_log, d_log, y_size, a, b)
        p.run(d_pts, d_sc)
        parts.append(p.window_points_raw())
    cat = np.concatenate(parts, axis=1)
    assert np.array_equal(cat, ref)


def test_gen_points_and_one_shot():
    import ctypes as C
    from gkr_msm_amd import ffi
    L = ffi.lib()
    n, seed = 64, 0x474b524d534d
    d_pts = harness.dev_empty(n * 8)
    ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, seed, harness.cur_stream()))
    got = codec.from_mont_limbs(harness.to_host(d_pts).reshape(-1, 4))
    rng = F.SplitMix64(seed)
    g = (18886178867200960497001835917649091219057080094937609519140440539760939937304,
         19188667384257783945677642223292697773471335439753913231509108946878080696678)
    pts = []
    for i in range(n):
        k = rng.next() | 1
        e = F.te_mul_affine(g, k)
        assert (got[2 * i], got[2 * i + 1]) == e
        pts.append(e)
    sc = F.random_scalars(n, 256, 9)
    d_sc = harness.to_dev(codec.ints_to_limbs(sc))
    out = np.zeros(8, dtype=np.uint64)
    ffi.check(L.gm_msm_te(C.c_void_p(d_pts.data_ptr()), C.c_void_p(d_sc.data_ptr()), 6, 8, 256, out.ctypes.data,
                          harness.cur_stream()))
    acc = (0, 1)
    for p, s in zip(pts, sc):
        acc = F.te_add_affine(acc, F.te_mul_affine(p, s))
    assert tuple(codec.from_mont_limbs(out.reshape(2, 4))) == acc


def test_bs_scalars_into_bigint():
    import ctypes as C
    from gkr_msm_amd import ffi
    L = ffi.lib()
    sc = F.random_scalars(100, 256, 3) + [0, 1, F.BS_ORDER - 1]
    mont = [s * F.BS_R % F.BS_ORDER for s in sc]
    d_in = harness.to_dev(codec.ints_to_limbs(mont))
    d_out = harness.dev_empty(len(sc) * 4)
    ffi.check(L.gm_bs_scalars_into_bigint(C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), len(sc),
                                          harness.cur_stream()))
    assert codec.limbs_to_ints(harness.to_host(d_out)) == sc


def test_fr_batch_ops():
    import ctypes as C
    from gkr_msm_amd import ffi
    L = ffi.lib()
    rng = F.SplitMix64(77)
    n = 1000
    a = [rng.next_fr() for _ in range(n)]
    b = [rng.next_fr() for _ in range(n)]
    a[0], b[0], a[1], b[1] = 0, 0, F.P - 1, F.P - 1
    da, db = harness.to_dev(codec.to_mont_limbs(a)), harness.to_dev(codec.to_mont_limbs(b))
    do = harness.dev_empty(n * 4)

    def run(op):
        ffi.check(L.gm_fr_batch(op, C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(do.data_ptr()),
                                n, harness.cur_stream()))
        return codec.from_mont_limbs(harness.to_host(do))
    assert run(0) == [(x + y) % F.P for x, y in zip(a, b)]
    assert run(1) == [(x - y) % F.P for x, y in zip(a, b)]
    assert run(2) == [(x * y) % F.P for x, y in zip(a, b)]
    assert run(3) == [(-x) % F.P for x in a]
    inv = run(4)
    assert all(x * y % F.P == 1 for x, y in zip(a[1:], inv[1:]))
    assert run(7) == [F.mul_by_a(x) for x in a]
    assert run(8) == [F.mul_by_d(x) for x in a]


@pytest.mark.parametrize("name", ["msm_x4_d2_n12.json", "msm_x5_d3_n24.json"])
def test_golden_fixture_msm_and_prover(name):
    """the committed fixtures (tests/golden) through the HIP path: MSM outputs and the image-part prover"""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)))
    ints = lambda xs: [int(x, 16) for x in xs]  # noqa: E731
    x_log, d_log, y_size, y_log = fx["x_logsize"], fx["d_logsize"], fx["y_size"], fx["y_logsize"]
    d_pts = harness.to_dev(codec.points_to_mont([(int(p[0], 16), int(p[1], 16)) for p in fx["points"]]))
    d_sc = harness.to_dev(codec.ints_to_limbs(ints(fx["scalars"])))
    plan = harness.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    dg, ct, _ = plan.digits_counter_rowlen()
    assert dg.tolist() == fx["digits"] and ct.tolist() == fx["counter"]
    bs = plan.bucket_sums()
    for c in range(3):
        assert bs[c] == ints(fx["bucket_sums"][c])
    wp = plan.window_points()
    for c in range(3 * (d_log + 1)):
        assert wp[c] == ints(fx["window_points"][c])[:y_size]
    assert list(harness.combine_host(plan.window_points_raw(), d_log)) == ints(fx["msm_result"])
    w = harness.PipWitness(plan, d_pts, y_log)
    res = w.prove_image_part(ints(fx["claim_point"]), ints(fx["claim_evs"]), ints(fx["tape"]))
    assert res["tape_used"] == len(fx["tape"])
    assert res["msgs"] == ints(fx["prover_messages"])
    assert res["point"] == ints(fx["final_point"]) and res["evs"] == ints(fx["final_evs"])


def test_phase1_polys_and_second_phase():
    """a12/a13: c, d, ac_c, ac_d (pushforward.rs:489-510) and c_pull / d_pull (second_phase, pushforward.rs:572-596)"""
    import ctypes as C
    from gkr_msm_amd import ffi
    x_log, d_log, nbits = 7, 4, 24
    y_size = 6
    y_log = 3
    n = 1 << x_log
    pts = F.random_points(n, 31)
    sc = F.random_scalars(n, nbits, 32)
    _, digits, counter = G.bucketing_image(pts, sc, y_size, y_log, d_log, x_log)
    d_pts = harness.to_dev(codec.points_to_mont(pts))
    d_sc = harness.to_dev(codec.ints_to_limbs(sc))
    plan = harness.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    L = ffi.lib()
    tot = y_size * n
    dc, dd = harness.dev_empty(tot * 4), harness.dev_empty(tot * 4)
    dac, dad = harness.dev_empty(n * 4), harness.dev_empty((1 << d_log) * 4)
    ffi.check(L.gm_msm_phase1_polys(plan.h, C.c_void_p(dc.data_ptr()), C.c_void_p(dd.data_ptr()), C.c_void_p(dac.data_ptr()),
                                    C.c_void_p(dad.data_ptr()), harness.cur_stream()))
    c, d, ac_c, ac_d = G.pushforward_phase1_polys(digits, counter, x_log, d_log)
    assert harness.cols_to_host([dc, dd, dac, dad]) == [c, d, ac_c, ac_d]
    rng = F.SplitMix64(5)
    r = [rng.next_fr() for _ in range(y_log + d_log + x_log)]
    cp, dp = harness.dev_empty(tot * 4), harness.dev_empty(tot * 4)
    ra = harness.fr_arg(r)
    ffi.check(L.gm_msm_second_phase(plan.h, ra.ctypes.data, y_log, C.c_void_p(cp.data_ptr()), C.c_void_p(dp.data_ptr()),
                                    harness.cur_stream()))
    e_c, e_d = G.pushforward_second_phase(digits, counter, r, y_log, d_log, x_log)
    assert harness.cols_to_host([cp, dp]) == [e_c, e_d]
