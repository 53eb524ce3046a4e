"""Shared by tests/test_ref_identity_cpu.py and tests/test_ref_identity_gpu.py: the inputs of the reference's `check_univars`
tests (sumchecks/dense_eq.rs:258-344, sumchecks/vecvec_eq.rs:511-600), re-drawn from this repo's SplitMix64 (the reference
draws from ark_std::test_rng(), an un-vendored dependency): random PROJECTIVE Bandersnatch points (x z, y z, z), the dense
generator `Vec::rand_points` (polys/dense.rs:194-209) and the three VecVec generators (polys/vecvec.rs:226-345)."""
from pyref import field as F
from pyref import polys as PL

P = F.P


def proj_points(rng, n):
    """n random projective points: an affine point of the prime-order subgroup scaled by a random z"""
    aff = F.random_points(n, rng.next())
    out = []
    for (x, y) in aff:
        z = rng.next_fr() or 1
        out.append((x * z % P, y * z % P, z))
    return out


def dense_rand_points(rng, num_vars):
    """[x, y, z] columns of 2^num_vars points (polys/dense.rs:194-209)"""
    pts = proj_points(rng, 1 << num_vars)
    return [[p[c] for p in pts] for c in range(3)]


def vecvec_rand_points(rng, row_logsize, col_logsize, denseness):
    """rows of [x, y, z] with pads (0, 1, 1): Full / Rows = every row full (rand_points_dense, rand_points_dense_rows are the same
    generator in the reference, vecvec.rs:226-305); Nothing = 1 .. 2^col rows of 1 .. 2^row cells (rand_points, :307-345)"""
    if denseness in ("full", "rows"):
        lens = [1 << row_logsize] * (1 << col_logsize)
    else:
        lens = [1 + rng.next() % (1 << row_logsize) for _ in range(1 + rng.next() % (1 << col_logsize))]
    rows = [proj_points(rng, l) for l in lens]
    data = [[[p[c] for p in r] for r in rows] for c in range(3)]
    pads = [(0, 0), (1, 1), (1, 1)]
    return data, pads


def vecvec_py(data, pads, row_logsize, col_logsize):
    return [PL.VecVec([list(r) for r in data[c]], pads[c][0], pads[c][1], row_logsize, col_logsize) for c in range(3)]
