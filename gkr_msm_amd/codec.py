"""Harness helpers: Python ints <-> the 4 x u64 little-endian Montgomery limbs used at the C ABI."""
import numpy as np

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P
R_INV = pow(R, -1, P)
_M64 = (1 << 64) - 1


def ints_to_limbs(vals):
    """list of ints (< 2^256) -> (n, 4) uint64, no Montgomery conversion"""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i, 0] = v & _M64
        out[i, 1] = (v >> 64) & _M64
        out[i, 2] = (v >> 128) & _M64
        out[i, 3] = (v >> 192) & _M64
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in arr]


def to_mont_limbs(vals):
    """canonical ints mod P -> (n, 4) uint64 Montgomery limbs"""
    return ints_to_limbs([(v % P) * R % P for v in vals])


def from_mont_limbs(arr):
    """(n, 4) uint64 Montgomery limbs -> canonical ints"""
    return [v * R_INV % P for v in limbs_to_ints(arr)]


def points_to_mont(points):
    """[(x, y), ...] canonical -> (n, 8) uint64 (x limbs then y limbs), the TE Affine{x,y} layout"""
    flat = []
    for (x, y) in points:
        flat.append(x)
        flat.append(y)
    return to_mont_limbs(flat).reshape(-1, 8)
