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


# ---- BLS12-381 Fq / G1 wire forms (include/gkrmsm.h, G1 section)
Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
QR = (1 << 384) % Q
QR_INV = pow(QR, -1, Q)


def fq_to_mont_limbs(vals):
    """canonical ints mod Q -> (n, 6) uint64 Montgomery limbs"""
    out = np.empty((len(vals), 6), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = (v % Q) * QR % Q
        for k in range(6):
            out[i, k] = (m >> (64 * k)) & _M64
    return out


def fq_from_mont_limbs(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 6)
    return [sum(int(r[k]) << (64 * k) for k in range(6)) * QR_INV % Q for r in arr]


def g1_aff_to_limbs(points):
    """[(x, y) or None, ...] -> (n, 12) uint64; infinity = (0, 0)"""
    flat = []
    for p in points:
        flat.extend((0, 0) if p is None else p)
    return fq_to_mont_limbs(flat).reshape(-1, 12)


def g1_aff_from_limbs(arr):
    v = fq_from_mont_limbs(np.asarray(arr, dtype=np.uint64).reshape(-1, 6))
    return [None if (v[2 * i] == 0 and v[2 * i + 1] == 0) else (v[2 * i], v[2 * i + 1]) for i in range(len(v) // 2)]


def g1_jac_to_limbs(points, zs=None):
    """affine points (or None) -> (n, 18) uint64 Jacobian (x z^2, y z^3, z) with the given z values (default 1)"""
    flat = []
    for i, p in enumerate(points):
        z = 1 if zs is None else zs[i] % Q
        if p is None:
            flat.extend((0, 1, 0))
        else:
            flat.extend((p[0] * z * z % Q, p[1] * z * z * z % Q, z))
    return fq_to_mont_limbs(flat).reshape(-1, 18)


def g1_jac_from_limbs(arr):
    """(n, 18) uint64 Jacobian -> affine points (or None)"""
    v = fq_from_mont_limbs(np.asarray(arr, dtype=np.uint64).reshape(-1, 6))
    out = []
    for i in range(len(v) // 3):
        x, y, z = v[3 * i:3 * i + 3]
        if z == 0:
            out.append(None)
        else:
            zi = pow(z, -1, Q)
            out.append((x * zi * zi % Q, y * zi * zi * zi % Q))
    return out


# the standard BLS12-381 G1 generator (affine, canonical)
G1_GEN = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
          0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
