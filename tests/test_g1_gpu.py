"""GPU parity of the BLS12-381 G1 side through the C ABI vs the Python oracle (oracle/pyref/g1.py): device point formulas,
the sum-by-key engine (bucket sums), MSM over affine and over projective bases (msm_nonaffine.rs), weighted sums,
binary_msm / prepare_bases, Pullback::bucketed_msm, and the G1 part of PushForwardState::new.  All results are compared as
group elements (affine coordinates) -- the equality `Projective` has in ark-ec."""
import numpy as np
import pytest

from gkr_msm_amd import codec, harness as H
from pyref import field as F
from pyref import g1 as G
from pyref import polys as PL

pytestmark = pytest.mark.gpu


def _scalars(n, seed, nbits=255):
    rng = F.SplitMix64(seed)
    return [rng.next_fr() & ((1 << nbits) - 1) for _ in range(n)]


def test_device_point_formulas():
    n = 70
    rng = F.SplitMix64(1)
    pts, qts = G.random_points(n, 2), G.random_points(n, 3)
    qts[0] = pts[0]
    qts[1] = G.neg(pts[1])
    pts[2] = None
    qts[3] = None
    pts[4] = qts[4] = None
    zs = [rng.next_fr() | 1 for _ in range(n)]
    zt = [rng.next_fr() | 1 for _ in range(n)]
    want = [G.add(p, q) for p, q in zip(pts, qts)]
    ja, jb = H.g1_jac_dev(pts, zs), H.g1_jac_dev(qts, zt)
    aa, ab = H.g1_aff_dev(pts), H.g1_aff_dev(qts)
    assert H.g1_read_jac(H.g1_batch(0, ja, jb, n, 18)) == want
    assert H.g1_read_jac(H.g1_batch(2, ja, ab, n, 18)) == want
    assert H.g1_read_jac(H.g1_batch(4, aa, ab, n, 18)) == want
    assert H.g1_read_jac(H.g1_batch(1, ja, None, n, 18)) == [G.double(p) for p in pts]
    assert H.g1_read_aff(H.g1_batch(3, ja, None, n, 12)) == pts


def test_gen_points_are_multiples_of_the_generator():
    n, seed = 9, 77
    got = H.g1_read_aff(H.g1_gen_points(n, seed))
    M = (1 << 64) - 1
    for i, p in enumerate(got):
        z = (seed + 0x9E3779B97F4A7C15 * (i + 1)) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z = (z ^ (z >> 31)) | 1
        assert p == G.mul(G.GEN, z)


@pytest.mark.parametrize("n,nb,seed", [(1, 1, 5), (50, 7, 6), (300, 16, 7), (257, 3, 8), (64, 64, 9)])
def test_bucket_sums_vs_oracle(n, nb, seed):
    rng = F.SplitMix64(seed)
    bases = G.random_points(n, seed + 100)
    if n > 10:
        bases[3] = bases[2]           # equal neighbours: doubling inside the tree
        bases[5] = G.neg(bases[4])    # cancel to infinity
        bases[6] = None               # infinity as an input
    mapping = [rng.next() % nb for _ in range(n)]
    if n > 10:
        mapping[2] = mapping[3] = mapping[4] = mapping[5] = 0
        if nb > 2:
            mapping = [m if m != nb - 1 else 0 for m in mapping]   # the last bucket stays empty
    want = [None] * nb
    for b, m in zip(bases, mapping):
        want[m] = G.add(want[m], b)
    got = H.g1_read_jac(H.g1_bucket_sums(H.g1_aff_dev(bases), mapping, nb))
    assert got == want


def test_bucket_sums_rejects_out_of_range_mapping():
    bases = G.random_points(4, 1)
    with pytest.raises(Exception):
        H.g1_bucket_sums(H.g1_aff_dev(bases), [0, 1, 9, 1], 3)


@pytest.mark.parametrize("n,nbits,seed", [(1, 255, 1), (3, 255, 2), (33, 255, 3), (150, 64, 4), (40, 17, 5)])
def test_msm_affine_and_nonaffine_vs_reference_algorithm(n, nbits, seed):
    bases = G.random_points(n, 50 + seed)
    sc = _scalars(n, 60 + seed, nbits)
    sc[0] = 0 if n > 1 else sc[0]
    if n > 2:
        sc[1], sc[2] = 1, (F.P - 1) & ((1 << nbits) - 1)
    want = G.msm_bigint_wnaf_nonaff(bases, sc)        # the branch the reference takes for G1 (msm_nonaffine.rs:45-46)
    assert want == G.msm_bigint_nonaff(bases, sc)
    d_sc = H.to_dev(codec.ints_to_limbs(sc))
    assert H.g1_msm(H.g1_aff_dev(bases), d_sc, n, nbits=nbits if nbits < 255 else 255) == want
    rng = F.SplitMix64(seed)
    zs = [rng.next_fr() | 1 for _ in range(n)]
    jb = list(bases)
    if n > 4:
        jb[4] = None                                   # projective bases may be the identity (empty outer buckets)
        want = G.msm_bigint_wnaf_nonaff(jb, sc)
    assert H.g1_msm_nonaff(H.g1_jac_dev(jb, zs), d_sc, n) == want
    # Montgomery scalars: `into_bigint()` on the device
    assert H.g1_msm_nonaff(H.g1_jac_dev(jb, zs), H.to_dev(codec.to_mont_limbs(sc)), n, mont=True) == want


def test_msm_linearity_and_constant_scalar_at_size():
    """size-independent properties at 2^14 points: msm(s1) + msm(s2) == msm(s1 + s2); msm(k, k, ...) == k * sum(bases)"""
    n = 1 << 14
    d_b = H.g1_gen_points(n, 5)
    rng = np.random.default_rng(3)
    s1 = rng.integers(0, 2 ** 62, size=(n, 4), dtype=np.uint64)
    s2 = rng.integers(0, 2 ** 62, size=(n, 4), dtype=np.uint64)   # limbs < 2^62: the limb-wise sum is the integer sum
    a = H.g1_msm(d_b, H.to_dev(s1), n)
    b = H.g1_msm(d_b, H.to_dev(s2), n)
    c = H.g1_msm(d_b, H.to_dev(s1 + s2), n)
    assert G.add(a, b) == c and c is not None
    total = H.g1_read_jac(H.g1_bucket_sums(d_b, [0] * n, 1))[0]
    k = 0x1234567_89ABCDEF_0FEDCBA9_87654321
    ks = codec.ints_to_limbs([k] * n)
    assert H.g1_msm(d_b, H.to_dev(ks), n) == G.mul(total, k)


def test_weighted_sum_is_the_running_sum_loop():
    groups, glen = 3, 37
    pts = G.random_points(groups * glen, 8)
    pts[5] = None
    got = H.g1_weighted_sum(H.g1_jac_dev(pts), groups, glen)
    assert got == [G.running_sum_reduce(pts[g * glen:(g + 1) * glen]) for g in range(groups)]


@pytest.mark.parametrize("gamma", [8, 3, 5])
def test_binary_msm_like_the_reference_tests(gamma):
    """binary_msm.rs:62-95"""
    num = 100
    rng = F.SplitMix64(20 + gamma)
    bits = [bool(rng.next() & 1) for _ in range(num)]
    bases = G.random_points(num, 30 + gamma)
    tables = H.g1_prepare_bases(H.g1_aff_dev(bases), num, gamma)
    want_tables = [e for t in G.prepare_bases(bases, gamma) for e in (t + [None] * ((1 << gamma) - 1 - len(t)))]
    assert H.g1_read_aff(tables) == want_tables
    got = H.g1_binary_msm(G.prepare_coefs(bits, gamma), tables, gamma)
    assert got == G.naive_msm(bases, [1 if b else 0 for b in bits])
    assert got == G.binary_msm(G.prepare_coefs(bits, gamma), G.prepare_bases(bases, gamma))


def test_pullback_bucketed_msm():
    """pullback.rs:83-106 at reduced size"""
    rng = F.SplitMix64(31)
    mapping = [rng.next() % 16 for _ in range(96)]
    image = [rng.next_fr() for _ in range(16)]
    bases = G.random_points(96, 32)
    assert H.g1_pullback_msm(H.g1_aff_dev(bases), mapping, image) == G.pullback_bucketed_msm(mapping, image, bases)


@pytest.mark.parametrize("x_log,d_log,nbits,clm", [(4, 2, 8, 0), (5, 3, 12, 1), (4, 2, 10, 2)])
def test_pushforward_outer_buckets_and_commitments(x_log, d_log, nbits, clm):
    y_size = (nbits + d_log - 1) // d_log
    y_log = PL.log2_exact(y_size) if (y_size & (y_size - 1)) == 0 else (y_size - 1).bit_length()
    n = 1 << x_log
    pts = F.random_points(n, 7)
    sc = F.random_scalars(n, nbits, 8)
    sc[0] = 0
    d_pts = H.to_dev(codec.points_to_mont(pts))
    d_sc = H.to_dev(codec.ints_to_limbs(sc))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    dg, ct, rl = plan.digits_counter_rowlen()
    basis = G.random_points(n << clm, 9)
    d_out, c_out, d_comm, c_comm = G.pushforward_outer(dg.tolist(), ct.tolist(), basis, x_log, d_log, clm)
    g_d, g_c, stride, g_dc, g_cc = H.msm_g1_outer(plan, H.g1_aff_dev(basis), clm, n)
    assert stride == int(rl.max())
    n_mat = len(d_out)
    got_d = H.g1_read_jac(g_d)
    got_c = H.g1_read_jac(g_c)
    for m in range(n_mat):
        assert got_d[m * (1 << d_log):(m + 1) * (1 << d_log)] == d_out[m]
        row = got_c[m * stride:(m + 1) * stride]
        assert row[:len(c_out[m])] == c_out[m] and all(p is None for p in row[len(c_out[m]):])
    assert g_dc == d_comm and g_cc == c_comm
    # the identity the reference keeps as a commented assert (pushforward.rs:526-529): d_comm == commit(d), c_comm == commit(c)
    cm = 1 << clm
    for m in range(n_mat):
        ys = range(m * cm, min((m + 1) * cm, y_size))
        assert d_comm[m] == G.naive_msm([basis[x + n * (y % cm)] for y in ys for x in range(n)],
                                        [int(dg[y][x]) for y in ys for x in range(n)])
        assert c_comm[m] == G.naive_msm([basis[x + n * (y % cm)] for y in ys for x in range(n)],
                                        [int(ct[y][x]) for y in ys for x in range(n)])
    # phase 2 (pushforward.rs:596-605): msm_nonaff over the outer buckets with the eq tables
    rng = F.SplitMix64(10)
    eq_d = PL.eq_poly_sequence_last([rng.next_fr() for _ in range(d_log)])
    d_eq = H.to_dev(codec.to_mont_limbs(eq_d))
    for m in range(n_mat):
        sub = g_d[m * 18 * (1 << d_log):(m + 1) * 18 * (1 << d_log)]
        assert H.g1_msm_nonaff(sub, d_eq, 1 << d_log, mont=True) == G.msm_nonaff(d_out[m], eq_d)


def test_grouped_msm_matches_separate_msms():
    """gm_g1_msm_nonaff_grouped: several projective base arrays (different lengths, one stride) against one scalar array"""
    stride, ns = 40, [40, 0, 7, 33, 1]
    rng = F.SplitMix64(61)
    sc = [rng.next_fr() for _ in range(stride)]
    sc[0] = 0
    groups = [G.random_points(stride, 70 + g) for g in range(len(ns))]
    groups[3][5] = None
    zs = [rng.next_fr() | 1 for _ in range(stride * len(ns))]
    flat = [p for grp in groups for p in grp]
    got = H.g1_msm_nonaff_grouped(H.g1_jac_dev(flat, zs), stride, ns, H.to_dev(codec.to_mont_limbs(sc)), mont=True)
    assert got == [G.msm_nonaff(grp[:n], sc[:n]) if n else None for grp, n in zip(groups, ns)]


@pytest.mark.parametrize("lp,lb,lcols,gamma", [(2, 2, 1, 3), (3, 3, 2, 5), (2, 4, 0, 8)])
def test_gen1_column_commitments(lp, lb, lcols, gamma):
    """gm_gkr_msm_commit = the commitments gkr_msm_prove writes first (gkr_msm_simple.rs:117-151)"""
    import ctypes as C
    import torch
    from gkr_msm_amd import ffi
    npts, nbits_total = 1 << lp, 1 << (lp + lb)
    col_size = nbits_total >> lcols
    pts = F.random_points(npts, 5)
    rng = F.SplitMix64(7)
    bits = [rng.next() & 1 for _ in range(nbits_total)]
    bases = G.random_points(col_size, 9)
    tables = G.prepare_bases(bases, gamma)
    want_bits = [G.binary_msm(G.prepare_coefs([bool(b) for b in bits[col_size * i:col_size * (i + 1)]], gamma), tables)
                 for i in range(1 << lcols)]
    prep = [p[0] for p in pts] + [p[1] for p in pts] + [0] * (col_size - 2 * npts)
    want_pts = G.naive_msm(bases, prep)
    d_bases = H.g1_aff_dev(bases)
    d_tables = H.g1_prepare_bases(d_bases, col_size, gamma)
    d_pts = H.to_dev(codec.points_to_mont(pts))
    d_bits = torch.from_numpy(np.array(bits, dtype=np.uint8)).cuda()
    hb = np.zeros((1 << lcols, 12), dtype=np.uint64)
    hp = np.zeros(12, dtype=np.uint64)
    ffi.check(ffi.lib().gm_gkr_msm_commit(C.c_void_p(d_pts.data_ptr()), C.c_void_p(d_bits.data_ptr()), lp, lb, lcols,
                                          C.c_void_p(d_bases.data_ptr()), C.c_void_p(d_tables.data_ptr()), gamma, hb.ctypes.data,
                                          hp.ctypes.data, H.cur_stream()))
    assert codec.g1_aff_from_limbs(hb) == want_bits
    assert codec.g1_aff_from_limbs(hp)[0] == want_pts


def test_mock_srs_is_powers_of_tau():
    rng = F.SplitMix64(88)
    tau = rng.next_fr()
    got = H.g1_read_aff(H.g1_mock_srs(tau, 9, G.GEN))
    cur = G.GEN
    for p in got:
        assert p == cur
        cur = G.mul(cur, tau)


def test_fixed_base_msm_is_the_same_group_element():
    """gm_g1_fixed_base_register: the MSMs over a registered proving key (16 x 16-bit windows of precomputed multiples, one bucket
    set) return the element the per-window path returns -- oracle-checked at a small size, path against path at 2^13, for full,
    prefix and narrow (access-count style) scalars, an infinity among the bases and zero scalars"""
    n = 40
    bases = G.random_points(n, 21)
    bases[7] = None
    rng = F.SplitMix64(77)
    sc = [rng.next_fr() for _ in range(n)]
    sc[3] = 0
    sc[4] = F.P - 1
    d_b = H.g1_aff_dev(bases)
    d_sc = H.to_dev(codec.ints_to_limbs(sc))
    want = G.naive_msm(bases, sc)
    assert H.g1_msm(d_b, d_sc, n) == want
    H.g1_fixed_base_register(d_b, n)
    try:
        assert H.g1_msm(d_b, d_sc, n) == want
        assert H.g1_msm(d_b, d_sc, 17) == G.naive_msm(bases[:17], sc[:17])            # a prefix of the key
        assert H.g1_msm(d_b, H.to_dev(codec.to_mont_limbs(sc)), n, mont=True) == want
        small = [rng.next() & 0xFFFFFFFF for _ in range(n)]
        assert H.g1_msm(d_b, H.to_dev(codec.ints_to_limbs(small)), n, nbits=32) == G.naive_msm(bases, small)
    finally:
        H.g1_fixed_base_release(d_b)
    n = 1 << 13
    d_b = H.g1_gen_points(n, 9)
    s = np.random.default_rng(5).integers(0, 2 ** 64, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 62) - 1)
    d_s = H.to_dev(s)
    plain = H.g1_msm(d_b, d_s, n)
    H.g1_fixed_base_register(d_b, n)
    try:
        assert H.g1_msm(d_b, d_s, n) == plain and plain is not None
        assert H.g1_msm(d_b, d_s, n - 5) == H.g1_msm(H.g1_gen_points(n, 9), d_s, n - 5)
    finally:
        H.g1_fixed_base_release(d_b)
