"""Integer model of the 9 x 29-bit field form (gkr_msm_amd/csrc/fr9.hip.h) and of every formula the kernels evaluate in it
(msm.hip: aff_add9 / proj_add9; sumcheck.hip: lean_gamma_eval9 + the accumulate / finish steps of k_round_deg2_lean9).

The device code relies on static bounds: a column of the product never exceeds 64 bits, a limb-wise difference never goes below
zero, a limb-wise sum never wraps 32 bits, a value handed to fr9_store is below 33 p.  The model restates the routines limb for
limb and ASSERTS those conditions while it runs them on adversarial canonical inputs (0, 1, p - 1, p - 2, 2^254, values whose
29-bit limbs are all ones, values just above / below multiples of 2^29) and on random ones, and checks every result against plain
arithmetic mod p.  It is the CPU-side companion of scripts/ubench/fr9_mul_test.hip (tests/test_fr9_gpu.py), which compares the
device code itself with the 8 x 32 field."""
import itertools
import random

P = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
M29 = (1 << 29) - 1
U32 = 1 << 32
U64 = 1 << 64
R256 = pow(2, 256, P)
D_TE = 45022363124591815672509500913686876175488063829319466900776701791074614335719  # Bandersnatch d (utils.rs:35)


def limbs(v):
    return [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]


def value(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


P9 = limbs(P)
assert P9[0] == 1


def bias(k):
    n = limbs(k * P)
    c = [n[0] + (1 << 30)] + [n[i] + (1 << 30) - 2 for i in range(1, 8)] + [n[8] - 2]
    assert value(c) == k * P
    return c


BIAS8, BIAS32, BIAS64 = bias(8), bias(32), bias(64)
assert BIAS64 == [0x40000040, 0x5ffffdfe, 0x45bfeffd, 0x52017ffd, 0x40154ef4, 0x41013439, 0x483339d6, 0x594cebe8, 0x1cfb69d2]
# the constants of fr9.hip.h
assert BIAS8 == [0x40000008, 0x5fffffbe, 0x5cb7fdfd, 0x5a402ffd, 0x4c02a9dc, 0x40202685, 0x49066739, 0x53299d7b, 0x039f6d38]
assert BIAS32 == [0x40000020, 0x5ffffefe, 0x52dff7fd, 0x4900bffd, 0x500aa779, 0x40809a1b, 0x44199cea, 0x4ca675f3, 0x0e7db4e8]
assert limbs(pow(2, 261, P)) == [0x1fffffba, 0x22f, 0x1cb61180, 0x0a4e5c00, 0x0ee8b1a2, 0x16e6aedf, 0x1907f8bb, 0x0853ddf7, 0x4d043f]
assert limbs(R256) == [0x1ffffffe, 0xf, 0x00d20080, 0x096ff400, 0x04ff5588, 0x07f7f65e, 0x15be6631, 0x0b3598a0, 0x1824b1]
assert limbs(pow(2, 271, P))[0] == 0x1ffee558 and limbs(pow(2, 276, P))[0] == 0x1fdcaaf7
assert limbs(D_TE * pow(2, 261, P) % P) == [0x1458e5f2, 0x1ced1bb7, 0x0c2440c6, 0x03a6574f, 0x06ebc6f2, 0x05d944c8, 0x185ecb02, 0x1cdb6c09,
                                            0x6ed285]


def mul(a, b):
    """fr9_mul: product scanning; every column sum must fit 64 bits"""
    m, r, acc = [0] * 9, [0] * 9, 0
    for k in range(17):
        for i in range(max(0, k - 8), min(k, 8) + 1):
            acc += a[i] * b[k - i]
        if k < 9:
            for j in range(k):
                acc += m[j] * P9[k - j]
            assert acc < U64, "column %d overflows" % k
            m[k] = (-acc) & M29
            acc += m[k]
            assert acc < U64 and acc & M29 == 0
        else:
            for j in range(k - 8, 9):
                acc += m[j] * P9[k - j]
            assert acc < U64, "column %d overflows" % k
            r[k - 9] = acc & M29
        acc >>= 29
    assert acc < U32
    r[8] = acc
    return r


def sqr(a):
    assert all(2 * x < U32 for x in a)
    return mul(a, a)      # the device routine sums the same terms (doubled limbs for the symmetric ones)


def add(a, b):
    r = [x + y for x, y in zip(a, b)]
    assert all(x < U32 for x in r), "limb-wise sum wraps"
    return r


def mul5(a):
    r = [5 * x for x in a]
    assert all(x < U32 for x in r)
    return r


def sub_bias(a, bs, c):
    r = list(a)
    for i in range(9):
        r[i] += c[i]
        for b in bs:
            r[i] -= b[i]
        assert 0 <= r[i] < U32, "limb %d of a difference leaves [0, 2^32)" % i
    return r


def norm(a):
    r, c = [0] * 9, 0
    for i in range(8):
        t = a[i] + c
        assert t < U32
        r[i], c = t & M29, t >> 29
    r[8] = a[8] + c
    assert r[8] < U32
    return r


def load_shifted(x):     # fr9_load: the limbs of 32 X
    return limbs(x << 5)


def load_raw(x):         # fr9_load_raw
    return limbs(x)


def store_shifted(y):    # fr9_store: (Y + m p) / 32, one conditional subtraction
    assert value(y) < 33 * P, "fr9_store needs a value below 33 p"
    m = (-y[0]) & 31
    z = value(y) + m * P
    assert z % 32 == 0
    w = z >> 5
    assert w < 2 * P and w < 1 << 256
    return w - P if w >= P else w


def store_raw(y):        # fr9_to_raw: normalised limbs, value below 2 p
    assert all(x <= M29 for x in y[:8])
    w = value(y)
    assert w < 2 * P and w < 1 << 256
    return w - P if w >= P else w


ONE261 = limbs(pow(2, 261, P))
ONE256 = limbs(R256)
D261 = limbs(D_TE * pow(2, 261, P) % P)
ZERO = [0] * 9


def mont(v):   # canonical Montgomery form as stored in memory
    return v * R256 % P


def unmont(x):
    return x * pow(R256, -1, P) % P


# ------------------------------------------------------------------------------------------------ MSM formulas (shifted form)
def aff_add9(x1, y1, x2, y2):
    A, B = mul(x1, x2), mul(y1, y2)
    C = mul(add(x1, y1), add(x2, y2))
    s = sub_bias(C, [A, B], BIAS32)
    t = norm(add(B, mul5(A)))
    dxy = mul(mul(A, B), D261)
    m = norm(sub_bias(ONE261, [dxy], BIAS8))
    q = add(ONE261, dxy)
    return mul(m, s), mul(q, t), mul(m, q)


def proj_add9(p, g):
    A, B, zz = mul(p[0], g[0]), mul(p[1], g[1]), mul(p[2], g[2])
    C = mul(add(p[0], p[1]), add(g[0], g[1]))
    s = sub_bias(C, [A, B], BIAS32)
    t = add(B, mul5(A))
    X, Y, z2 = mul(s, zz), mul(t, zz), sqr(zz)
    dxy = mul(mul(A, B), D261)
    m = norm(sub_bias(z2, [dxy], BIAS8))
    q = add(z2, dxy)
    return mul(m, X), mul(q, Y), mul(m, q)


def ref_aff(x1, y1, x2, y2):
    A, B = x1 * x2 % P, y1 * y2 % P
    s, t = (x1 * y2 + x2 * y1) % P, (B + 5 * A) % P
    dxy = D_TE * A * B % P
    m, q = (1 - dxy) % P, (1 + dxy) % P
    return m * s % P, q * t % P, m * q % P


def ref_proj(p, g):
    A, B, zz = p[0] * g[0] % P, p[1] * g[1] % P, p[2] * g[2] % P
    s, t = (p[0] * g[1] + g[0] * p[1]) % P, (B + 5 * A) % P
    X, Y, z2 = s * zz % P, t * zz % P, zz * zz % P
    dxy = D_TE * A * B % P
    m, q = (z2 - dxy) % P, (z2 + dxy) % P
    return m * X % P, q * Y % P, m * q % P


def adversarial_values():
    vals = [0, 1, 2, P - 1, P - 2, 1 << 254, (1 << 254) - 1, (1 << 232) - 1, 1 << 232, (1 << 29) - 1, 1 << 29]
    ones = sum(M29 << (29 * i) for i in range(8)) + ((P >> 232) - 1 << 232)   # every low limb all ones, top limb just below p's
    vals += [ones % P, (ones - 1) % P, P >> 1, (P >> 1) + 1]
    return vals


def stored_values(n_random, seed):
    rng = random.Random(seed)
    # what memory holds is the Montgomery image; adversarial patterns are applied to the STORED integers (that is what the
    # limb bounds see), random field elements cover the rest
    return adversarial_values() + [rng.randrange(P) for _ in range(n_random)]


def test_msm_formulas_stay_inside_their_bounds_and_give_the_field_values():
    vals = stored_values(6, 1)
    rng = random.Random(2)
    quads = [tuple(rng.choice(vals) for _ in range(4)) for _ in range(300)] + list(itertools.product([0, P - 1, vals[11]], repeat=4))
    for x1, y1, x2, y2 in quads:
        got = [store_shifted(c) for c in aff_add9(*(load_shifted(v) for v in (x1, y1, x2, y2)))]
        exp = [mont(c) for c in ref_aff(*(unmont(v) for v in (x1, y1, x2, y2)))]
        assert got == exp
    hexes = [tuple(rng.choice(vals) for _ in range(6)) for _ in range(300)] + list(itertools.product([0, P - 1], repeat=6))
    for h in hexes:
        p, g = h[:3], h[3:]
        got = [store_shifted(c) for c in proj_add9([load_shifted(v) for v in p], [load_shifted(v) for v in g])]
        exp = [mont(c) for c in ref_proj([unmont(v) for v in p], [unmont(v) for v in g])]
        assert got == exp
    # the identity's constants as operands (k_add_level0's pad cell: (0, 1))
    for x1, y1 in [(vals[3], vals[4]), (0, R256)]:
        got = [store_shifted(c) for c in aff_add9(load_shifted(x1), load_shifted(y1), ZERO, ONE261)]
        assert got == [mont(c) for c in ref_aff(unmont(x1), unmont(y1), 0, 1)]


# ------------------------------------------------------------------------------------------------ large-round kernels (raw form)
ONE251 = limbs(1 << 251)


def eval9(prim, ld, g, lds=None):
    """lean_gamma_eval9: ld(q) = input q (domain 256, normalised, S <= 10); lds(q) = the same input from shifted loads (domain 261,
    S <= 128); g[o] = gamma^o loaded shifted (domain 261)"""
    if prim == "ADD_INVERSES":
        v0 = ld(0)
        t = mul(mul(g[1], v0), lds(1))
        return add(add(v0, ld(1)), t)
    if prim == "AFF_L2":
        v0 = ld(0)
        A = mul(mul(g[2], v0), lds(1))
        A = add(A, mul(g[1], ld(2)))
        return add(add(A, v0), ld(1))
    if prim == "PT_BIT_CHOICE":
        b = ld(0)
        by = add(mul(b, norm(sub_bias(ld(2), [ONE256], BIAS8))), ONE251)
        return add(mul(b, ld(1)), mul(g[1], by))
    if prim == "LOGUP_LAYER":
        v1, v3 = ld(1), ld(3)
        A = mul(ld(0), v3)
        A = add(A, mul(v1, ld(2)))
        return add(A, mul(g[1], mul(v1, v3)))
    if prim in ("AFF_L1", "AFF_L1_BC"):
        v3, v2, v0 = ld(3), ld(2), ld(0)
        A = mul(v0, v3)
        t = mul5(mul(v0, v2))
        v1 = ld(1)
        t = add(t, mul(v1, v3))
        A = add(A, mul(g[1], mul(v2, v1)))
        A = add(A, mul(g[2], t))
        if prim == "AFF_L1_BC":
            for k in range(2):
                b = ld(4 + k)
                A = add(A, mul(g[3 + k], mul(b, norm(sub_bias(b, [ONE256], BIAS8)))))
        return A
    if prim in ("AFF_L3", "PROJ_L3"):
        dxy = mul(ld(2 if prim == "AFF_L3" else 3), D261)
        base = ONE256 if prim == "AFF_L3" else ld(2)
        m = norm(sub_bias(base, [dxy], BIAS8))
        q = add(base, dxy)
        A = mul(m, ld(0))
        A = add(A, mul(g[1], mul(q, ld(1))))
        return add(A, mul(g[2], mul(m, q)))
    if prim == "PROJ_L1":
        v3, v4, v0 = ld(3), ld(4), ld(0)
        A = mul(v0, v4)
        t = mul5(mul(v0, v3))
        v1 = ld(1)
        t = add(t, mul(v1, v4))
        A = add(A, mul(g[1], mul(v3, v1)))
        A = add(A, mul(g[2], t))
        return add(A, mul(g[3], mul(ld(2), ld(5))))
    assert prim == "PROJ_L2"
    v0, v1, v3 = ld(0), ld(1), ld(3)
    A = mul(add(v0, v1), v3)
    A = add(A, mul(g[1], mul(ld(2), v3)))
    A = add(A, mul(g[2], sqr(v3)))
    return add(A, mul(g[3], mul(v0, v1)))


def ref_eval(prim, v, g):
    if prim == "ADD_INVERSES":
        return (v[0] + v[1] + g[1] * v[0] * v[1]) % P
    if prim == "AFF_L2":
        return (v[0] + v[1] + g[1] * v[2] + g[2] * v[0] * v[1]) % P
    if prim == "PT_BIT_CHOICE":
        return (v[0] * v[1] + g[1] * (v[0] * (v[2] - 1) + 1)) % P
    if prim == "LOGUP_LAYER":
        return (v[0] * v[3] + v[1] * v[2] + g[1] * v[1] * v[3]) % P
    if prim in ("AFF_L1", "AFF_L1_BC"):
        A = v[0] * v[3] + g[1] * v[2] * v[1] + g[2] * (v[1] * v[3] + 5 * v[0] * v[2])
        if prim == "AFF_L1_BC":
            A += g[3] * (v[4] * v[4] - v[4]) + g[4] * (v[5] * v[5] - v[5])
        return A % P
    if prim in ("AFF_L3", "PROJ_L3"):
        dxy = D_TE * v[2 if prim == "AFF_L3" else 3]
        base = 1 if prim == "AFF_L3" else v[2]
        m, q = base - dxy, base + dxy
        return (m * v[0] + g[1] * q * v[1] + g[2] * m * q) % P
    if prim == "PROJ_L1":
        return (v[0] * v[4] + g[1] * v[3] * v[1] + g[2] * (v[1] * v[4] + 5 * v[0] * v[3]) + g[3] * v[2] * v[5]) % P
    return ((v[0] + v[1]) * v[3] + g[1] * v[2] * v[3] + g[2] * v[3] * v[3] + g[3] * v[0] * v[1]) % P


N_IN = {"AFF_L1": 4, "AFF_L1_BC": 6, "AFF_L3": 3, "PROJ_L1": 6, "PROJ_L2": 4, "PROJ_L3": 4, "AFF_L2": 3, "PT_BIT_CHOICE": 3, "ADD_INVERSES": 2,
        "LOGUP_LAYER": 4}
TERMS_256 = ("AFF_L2", "ADD_INVERSES")   # their terms stay in domain 256: the accumulators sit five places higher


def test_large_round_kernels_in_the_raw_form():
    """one thread of k_round_deg2_lean9: 40 pairs (so the every-16-pairs scale reduction runs twice), both evaluation points, the
    VecVec weight (eq x coef) and the dense one (eq), finish by 2^276 / 2^271 and the canonical store"""
    vals = stored_values(4, 3)
    rng = random.Random(4)
    for prim, ni in N_IN.items():
        for vecvec in (True, False):
            gam = [rng.choice(vals) for _ in range(5)]                      # stored gamma powers (index 0 unused)
            g9 = [load_shifted(x) for x in gam]
            a = [ZERO, ZERO]
            exp = [0, 0]
            for it in range(40):
                extreme = it < 12
                pick = (lambda: rng.choice(vals[:15])) if extreme else (lambda: rng.randrange(P))
                p0, p1 = [pick() for _ in range(ni)], [pick() for _ in range(ni)]
                eqv, coef = pick(), pick()
                if vecvec:
                    w = mul(load_raw(eqv), load_raw(coef))
                    w_ref = unmont(eqv) * unmont(coef) % P
                else:
                    w = load_raw(eqv)
                    w_ref = unmont(eqv)
                for h in range(2):
                    def ld(q, h=h):
                        x1 = load_raw(p1[q])
                        if not h:
                            return x1
                        return norm(sub_bias(add(x1, x1), [load_raw(p0[q])], BIAS8))
                    def lds(q, h=h):
                        x1 = load_shifted(p1[q])
                        if not h:
                            return x1
                        return norm(sub_bias(add(x1, x1), [load_shifted(p0[q])], BIAS64))
                    t = mul(eval9(prim, ld, g9, lds), w)
                    a[h] = norm(add(a[h], t))
                    v = [unmont(p1[q]) if not h else (2 * unmont(p1[q]) - unmont(p0[q])) % P for q in range(ni)]
                    exp[h] = (exp[h] + ref_eval(prim, v, [unmont(x) for x in gam]) * w_ref) % P
                if it & 15 == 15:
                    a = [mul(x, ONE261) for x in a]
            K = limbs(pow(2, (271 if vecvec else 266) if prim in TERMS_256 else (276 if vecvec else 271), P))
            got = [store_raw(mul(x, K)) for x in a]
            assert got == [mont(e) for e in exp], (prim, vecvec)


def test_conversions_and_single_products():
    vals = stored_values(50, 5)
    for x in vals:
        assert store_shifted(load_shifted(x)) == x
        assert store_raw(load_raw(x)) == x
    rng = random.Random(6)
    for _ in range(400):
        x, y = rng.choice(vals), rng.choice(vals)
        assert store_shifted(mul(load_shifted(x), load_shifted(y))) == mont(unmont(x) * unmont(y) % P)
        assert store_shifted(sqr(load_shifted(x))) == mont(unmont(x) ** 2 % P)
