"""CPU tests of the G1 side: the Python oracle (oracle/pyref/g1.py) against public BLS12-381 facts and against the
identities the reference's own tests assert (binary_msm.rs:62-95: binary_msm == sum of the selected bases;
pullback.rs:83-106: bucketed_msm == msm of the pulled-back values), and the library's host-side Fq / G1 code (same
source as the device code) against that oracle.  No GPU compute calls."""
import ctypes as C

import numpy as np

from gkr_msm_amd import codec, ffi
from pyref import field as F
from pyref import g1 as G


def test_curve_constants():
    assert G.Q.bit_length() == 381 and G.Q % 4 == 3
    assert G.on_curve(G.GEN)
    assert G.mul(G.GEN, G.R_ORDER - 1) == G.neg(G.GEN)          # r * G = O with r = the Fr modulus
    assert G.add(G.mul(G.GEN, G.R_ORDER - 1), G.GEN) is None
    # 2G, public value (e.g. the zkcrypto / IETF BLS12-381 vectors)
    assert G.double(G.GEN) == (
        0x0572CBEA904D67468808C8EB50A9450C9721DB309128012543902D0AC358A62AE28F75BB8F1C7C42C39A8C5529BF0F4E,
        0x166A9D8CABC673A322FDA673779D8E3822BA3ECB8670E461F73BB9021D5FD76A4C56D9D4CD16BD1BBA86881979749D28)
    assert codec.Q == G.Q


def test_msm_nonaffine_both_branches_vs_naive():
    n = 40
    pts = G.random_points(n, 11)
    rng = F.SplitMix64(12)
    sc = [rng.next_fr() for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, F.P - 1
    pts[5] = None                                  # projective bases may be the identity (outer buckets often are)
    pts[7] = pts[6]
    want = G.naive_msm(pts, sc)
    assert G.msm_bigint_wnaf_nonaff(pts, sc) == want
    assert G.msm_bigint_nonaff(pts, sc) == want
    small = [rng.next() & 0xFFFFF for _ in range(n)]   # <= 60 bits: the early-exit num_bits path (msm_nonaffine.rs:100-103)
    assert G.max_num_bits(small) <= 20
    assert G.msm_bigint_wnaf_nonaff(pts, small) == G.naive_msm(pts, small)
    assert G.msm_bigint_nonaff(pts, small) == G.naive_msm(pts, small)


def test_make_digits_recompose():
    rng = F.SplitMix64(3)
    for w in (3, 7, 13, 16):
        for _ in range(20):
            a = rng.next_fr()
            d = G.make_digits(a, w, 255)
            assert sum(x << (w * i) for i, x in enumerate(d)) == a
            assert all(-(1 << (w - 1)) <= x < (1 << (w - 1)) for x in d[:-1])


def test_binary_msm_pattern_of_reference_tests():
    """binary_msm.rs:62-95 (bin_msm, bin_msm_gamma_3): 100 random bits / bases, gamma 8 and 3"""
    num = 100
    rng = F.SplitMix64(21)
    bits = [bool(rng.next() & 1) for _ in range(num)]
    bases = G.random_points(num, 22)
    want = G.naive_msm(bases, [1 if b else 0 for b in bits])
    for gamma in (8, 3):
        assert G.binary_msm(G.prepare_coefs(bits, gamma), G.prepare_bases(bases, gamma)) == want


def test_pullback_pattern_of_reference_test():
    """pullback.rs:83-106 at reduced size"""
    rng = F.SplitMix64(31)
    mapping = [rng.next() % 16 for _ in range(96)]
    image = [rng.next_fr() for _ in range(16)]
    bases = G.random_points(96, 32)
    assert G.pullback_bucketed_msm(mapping, image, bases) == G.naive_msm(bases, G.pullback_values(mapping, image))


def test_host_fq_and_point_ops_vs_oracle():
    L = ffi.lib()
    rng = F.SplitMix64(41)
    n = 64

    def rq():
        return (rng.next_fr() * rng.next_fr() + rng.next()) % G.Q
    a = [rq() for _ in range(n)]
    b = [rq() for _ in range(n)]
    a[:3] = [0, G.Q - 1, 1]
    b[:3] = [5, G.Q - 1, G.Q - 1]
    A_, B_ = codec.fq_to_mont_limbs(a), codec.fq_to_mont_limbs(b)
    o = np.zeros_like(A_)
    for op in (6, 7):   # 64-bit host multiplier and the 32-bit-limb device formulation
        ffi.check(L.gm_g1_host(op, A_.ctypes.data, B_.ctypes.data, o.ctypes.data, n))
        assert codec.fq_from_mont_limbs(o) == [x * y % G.Q for x, y in zip(a, b)]
    pts = G.random_points(n, 42)
    qts = G.random_points(n, 43)
    qts[0] = pts[0]              # doubling through the addition formulas
    qts[1] = G.neg(pts[1])       # P + (-P)
    pts[2] = None
    qts[3] = None
    pts[4] = qts[4] = None
    zs = [rng.next_fr() | 1 for _ in range(n)]
    zt = [rng.next_fr() | 1 for _ in range(n)]
    want = [G.add(p, q) for p, q in zip(pts, qts)]
    ja, jb = codec.g1_jac_to_limbs(pts, zs), codec.g1_jac_to_limbs(qts, zt)
    aa, ab = codec.g1_aff_to_limbs(pts), codec.g1_aff_to_limbs(qts)
    oj = np.zeros((n, 18), dtype=np.uint64)
    oa = np.zeros((n, 12), dtype=np.uint64)
    ffi.check(L.gm_g1_host(0, ja.ctypes.data, jb.ctypes.data, oj.ctypes.data, n))
    assert codec.g1_jac_from_limbs(oj) == want
    ffi.check(L.gm_g1_host(2, ja.ctypes.data, ab.ctypes.data, oj.ctypes.data, n))
    assert codec.g1_jac_from_limbs(oj) == want
    ffi.check(L.gm_g1_host(4, aa.ctypes.data, ab.ctypes.data, oj.ctypes.data, n))
    assert codec.g1_jac_from_limbs(oj) == want
    ffi.check(L.gm_g1_host(1, ja.ctypes.data, None, oj.ctypes.data, n))
    assert codec.g1_jac_from_limbs(oj) == [G.double(p) for p in pts]
    ffi.check(L.gm_g1_host(3, ja.ctypes.data, None, oa.ctypes.data, n))
    assert codec.g1_aff_from_limbs(oa) == pts
    oc = np.zeros(n, dtype=np.uint64)
    ffi.check(L.gm_g1_host(5, aa.ctypes.data, None, oc.ctypes.data, n))
    assert oc.tolist() == [1] * n
    bad = codec.g1_aff_to_limbs([(3, 5)])
    ffi.check(L.gm_g1_host(5, bad.ctypes.data, None, oc.ctypes.data, 1))
    assert oc[0] == 0


def test_c_oracle_g1_vs_python_oracle():
    """the C restatement (CPU baseline of bench.py) against the Python big-int one"""
    import oracle_ffi as O
    n = 45
    rng = F.SplitMix64(51)
    bases = G.random_points(n, 52)
    bases[4] = None
    bases[6] = bases[5]
    sc = [rng.next_fr() for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, F.P - 1
    zs = [rng.next_fr() | 1 for _ in range(n)]
    want = G.msm_bigint_wnaf_nonaff(bases, sc)
    got = O.g1_msm_wnaf_nonaff(codec.g1_jac_to_limbs(bases, zs), codec.ints_to_limbs(sc), threads=2)
    assert codec.g1_aff_from_limbs(got)[0] == want
    got = O.g1_msm_affine(codec.g1_aff_to_limbs(bases), codec.ints_to_limbs(sc))
    assert codec.g1_aff_from_limbs(got)[0] == want
    small = [rng.next() & 0xFFFF for _ in range(n)]    # the <= 60-bit early exit
    got = O.g1_msm_affine(codec.g1_aff_to_limbs(bases), codec.ints_to_limbs(small))
    assert codec.g1_aff_from_limbs(got)[0] == G.msm_bigint_wnaf_nonaff(bases, small)
    # binary_msm
    bits = [bool(rng.next() & 1) for _ in range(n)]
    tabs = G.prepare_bases(bases, 4)
    flat = [e for t in tabs for e in (t + [None] * (15 - len(t)))]
    got = O.g1_binary_msm(G.prepare_coefs(bits, 4), codec.g1_aff_to_limbs(flat), 4)
    assert codec.g1_aff_from_limbs(got)[0] == G.binary_msm(G.prepare_coefs(bits, 4), tabs)
    # pushforward outer buckets: synthetic digits / counters consistent with a bucketing
    x_log, d_log, y_size, clm = 4, 2, 3, 1
    N = 1 << x_log
    digits = [[rng.next() % (1 << d_log) for _ in range(N)] for _ in range(y_size)]
    counter = []
    for y in range(y_size):
        seen, row = {}, []
        for x in range(N):
            row.append(seen.get(digits[y][x], 0))
            seen[digits[y][x]] = row[-1] + 1
        counter.append(row)
    basis = G.random_points(N << clm, 53)
    d_out, c_out, d_comm, c_comm = G.pushforward_outer(digits, counter, basis, x_log, d_log, clm)
    r = O.g1_pushforward_outer(np.array(digits), np.array(counter), codec.g1_aff_to_limbs(basis), x_log, d_log, y_size, clm, 2)
    gd, gc = codec.g1_jac_from_limbs(r["d_outer"]), codec.g1_jac_from_limbs(r["c_outer"])
    for m in range(len(d_out)):
        assert gd[m << d_log:(m + 1) << d_log] == d_out[m]
        row = gc[m * r["c_stride"]:(m + 1) * r["c_stride"]]
        assert row[:len(c_out[m])] == c_out[m] and all(p is None for p in row[len(c_out[m]):])
    assert codec.g1_aff_from_limbs(r["d_comm"]) == d_comm and codec.g1_aff_from_limbs(r["c_comm"]) == c_comm
