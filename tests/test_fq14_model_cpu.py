"""Integer model of the 14 x 28-bit form of the BLS12-381 base field (gkr_msm_amd/csrc/fq14.hip.h) and of the three G1 addition
formulas the device evaluates in it (g1_add14 / g1_add_mixed14 / g1_add_aff14 in g1.hip.h).

As for the 9 x 29 scalar field (test_fr9_model_cpu.py) the device code relies on static bounds: a product column never exceeds
64 bits, a limb-wise difference never goes negative, a limb never wraps 32 bits, a value handed to the store is below 256 q.  The
model restates the routines limb for limb, ASSERTS those conditions on adversarial and random inputs, and checks the results
against plain arithmetic mod q: the (X, Y, Z) a formula stores must be the canonical values the standard formulas (EFD
add-2007-bl / madd-2007-bl / mmadd-2007-bl, what the 12 x 32 path computes) give.  The device code itself is compared with the
12 x 32 path on the GPU (tests/test_g1_gpu.py: scripts/ubench/fq14_test.hip)."""
import random

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
M28 = (1 << 28) - 1
U32 = 1 << 32
U64 = 1 << 64
R384 = pow(2, 384, Q)
R392 = pow(2, 392, Q)
QINV28 = (-pow(Q, -1, 1 << 28)) % (1 << 28)
Q13 = Q >> 364
KC = (1 << 32) // (Q13 + 1)


def limbs(v):
    return [(v >> (28 * i)) & M28 for i in range(13)] + [v >> 364]


def value(l):
    return sum(x << (28 * i) for i, x in enumerate(l))


Q14 = limbs(Q)
# the constants of fq14.hip.h
assert Q14 == [0x0fffaaab, 0x0fefffff, 0x03ffffb9, 0x0fffeb15, 0x06241eab, 0x0a0f6b0f, 0x0f6730d2, 0x0f38512b, 0x04774b84, 0x04bacd76,
               0x0ba7b643, 0x0e69a4b1, 0x01ea397f, 0x0001a011]
assert QINV28 == 0x0ffcfffd and KC == 40323 and (-pow(Q, -1, 256)) % 256 == 253


def bias(k):
    """k q with every limb below the top one raised by 2^28 (paid for by the next one): dominates a normalised subtrahend"""
    n = limbs(k * Q)
    c = [n[0] + (1 << 28)] + [n[i] + (1 << 28) - 1 for i in range(1, 13)] + [n[13] - 1]
    assert value(c) == k * Q and all(x >= M28 for x in c[:13])
    return c


BIAS4, BIAS16 = bias(4), bias(16)
assert BIAS4 == [0x1ffeaaac, 0x1fbffffe, 0x1ffffee6, 0x1fffac53, 0x18907aae, 0x183dac3c, 0x1d9cc349, 0x1ce144ae, 0x11dd2e12, 0x12eb35d8,
                 0x1e9ed90c, 0x19a692c5, 0x17a8e5fe, 0x00068043]
assert BIAS16 == [0x1ffaaab0, 0x1efffffe, 0x1ffffb9e, 0x1ffeb152, 0x1241eabe, 0x10f6b0f5, 0x16730d29, 0x138512be, 0x1774b84e, 0x1bacd763,
                  0x1a7b6433, 0x169a4b1a, 0x1ea397fd, 0x001a0110]


def load(x):
    """fq14_load: canonical X = x 2^384 (12 x 32 in memory) -> the limbs of X 2^8 - k q, k from the top limb: below 1.1 q"""
    assert 0 <= x < (1 << 384)
    v = limbs(x << 8)
    assert v[13] < (1 << 28)
    k = (v[13] * KC) >> 32
    r, acc = [0] * 14, 0
    for i in range(14):
        acc += v[i] - k * Q14[i]          # signed 64-bit accumulator
        assert -(1 << 63) <= acc < (1 << 63)
        r[i] = acc & M28
        acc >>= 28                         # arithmetic shift
    assert acc == 0, "V - k q must be non-negative and fit"
    assert value(r) == (x << 8) - k * Q
    if x < Q:
        assert value(r) * 10 < 11 * Q
    return r


def mul(a, b):
    m, r, acc = [0] * 14, [0] * 14, 0
    for k in range(27):
        for i in range(max(0, k - 13), min(k, 13) + 1):
            acc += a[i] * b[k - i]
        if k < 14:
            for j in range(k):
                acc += m[j] * Q14[k - j]
            m[k] = ((acc & (U32 - 1)) * QINV28) & M28
            acc += m[k] * Q14[0]
            assert acc < U64 and acc & M28 == 0
        else:
            for j in range(k - 13, 14):
                acc += m[j] * Q14[k - j]
            assert acc < U64
            r[k - 14] = acc & M28
        acc >>= 28
    assert acc < U32
    r[13] = acc
    assert all(x < U32 for x in a) and all(x < U32 for x in b)
    assert value(r) * (1 << 392) == value(a) * value(b) + value(m) * Q
    return r


def sqr(a):
    """fq14_sqr: the symmetric terms once against the doubled limbs -- the same column sums as mul(a, a)"""
    assert all(2 * x < U32 for x in a)
    return mul(a, a)


def add(a, b):
    r = [x + y for x, y in zip(a, b)]
    assert all(x < U32 for x in r)
    return r


def shl(a, n):
    r = [x << n for x in a]
    assert all(x < U32 for x in r)
    return r


def sub(a, b, bias_c, k):
    assert all(y <= M28 for y in b[:13]) and b[13] <= bias_c[13], "subtrahend must be normalised and below k q"
    r = [x + c - y for x, c, y in zip(a, bias_c, b)]
    assert all(0 <= x < U32 for x in r)
    assert value(r) == value(a) + k * Q - value(b)
    return r


def norm(a):
    r, c = [0] * 14, 0
    for i in range(13):
        t = a[i] + c
        assert t < U32
        r[i] = t & M28
        c = t >> 28
    r[13] = a[13] + c
    assert r[13] < U32 and value(r) == value(a)
    return r


def store(y):
    """fq14_to: Y (< 256 q, limbs < 2^32) -> canonical (Y + m q) / 256 with one conditional subtraction"""
    assert value(y) < 256 * Q and all(x < U32 for x in y)
    m = (y[0] * 253) & 255
    z, acc = [0] * 14, 0
    for i in range(14):
        acc += m * Q14[i] + y[i]
        assert acc < U64
        z[i] = acc & M28 if i < 13 else acc
        acc >>= 28
    zz = value(z)
    assert zz % 256 == 0 and zz == value(y) + m * Q
    w = zz >> 8
    assert w < 2 * Q and w < (1 << 384)
    return w - Q if w >= Q else w


def maybe_zero(h):
    """the device's pre-filter for H = 0 (mod q): H = a + 4 q - b below 6 q, so only j q, j <= 5, can be it"""
    return any(((h[0] - j * Q14[0]) & M28) == 0 for j in range(6))


def tail(u1, s1, h, d, zf):
    """shared tail: H = U2 - U1 + 4 q, d = S2 - S1 + 4 q (unnormalised, < 5.2 q); returns stored X3, Y3, Z3"""
    hh = sqr(h)
    i4 = shl(hh, 2)
    j = mul(h, i4)
    v = mul(u1, i4)
    dd = sqr(d)
    t = norm(add(j, shl(v, 1)))
    x3 = norm(sub(shl(dd, 2), t, BIAS4, 4))
    w = sub(v, x3, BIAS16, 16)
    dw = mul(d, w)
    yj = mul(s1, j)
    y3 = shl(sub(dw, yj, BIAS4, 4), 1)
    z3 = shl(h if zf is None else mul(zf, h), 1)
    return store(x3), store(y3), store(z3)


def tail_cell(u1, s1, h, d, zf):
    """g1_tail14 as the tree keeps its cells (G1P14): X3, Y3, Z3 as normalised limbs, not stored"""
    hh = sqr(h)
    i4 = shl(hh, 2)
    j = mul(h, i4)
    v = mul(u1, i4)
    dd = sqr(d)
    t = norm(add(j, shl(v, 1)))
    x3 = norm(sub(shl(dd, 2), t, BIAS4, 4))
    w = sub(v, x3, BIAS16, 16)
    dw = mul(d, w)
    yj = mul(s1, j)
    y3 = norm(shl(sub(dw, yj, BIAS4, 4), 1))
    z3 = norm(shl(h if zf is None else mul(zf, h), 1))
    for c in (x3, y3, z3):
        assert all(l <= M28 for l in c[:13]) and value(c) * 10 < 103 * Q   # limbs < 2^28, S <= 10.3
    return x3, y3, z3


def add_cells(p, q):
    """g1_add14p on two cells (limbs < 2^28, S <= 10.3)"""
    (a, b, z1), (c, e, z2) = p, q
    z1z1, z2z2 = sqr(z1), sqr(z2)
    u1, u2 = mul(a, z2z2), mul(c, z1z1)
    s1, s2 = mul(mul(b, z2), z2z2), mul(mul(e, z1), z1z1)
    return tail_cell(u1, s1, sub(u2, u1, BIAS4, 4), sub(s2, s1, BIAS4, 4), mul(z1, z2))


def add_aff(x1, y1, x2, y2):
    a, b, c, e = load(x1), load(y1), load(x2), load(y2)
    return tail(a, b, sub(c, a, BIAS4, 4), sub(e, b, BIAS4, 4), None)


def add_mixed(X1, Y1, Z1, x2, y2):
    a, b, z, c, e = load(X1), load(Y1), load(Z1), load(x2), load(y2)
    zz = sqr(z)
    u2 = mul(c, zz)
    s2 = mul(mul(e, z), zz)
    h = sub(u2, a, BIAS4, 4)
    return tail(a, b, h, sub(s2, b, BIAS4, 4), z), h


def add_jac(X1, Y1, Z1, X2, Y2, Z2):
    a, b, z1, c, e, z2 = load(X1), load(Y1), load(Z1), load(X2), load(Y2), load(Z2)
    z1z1, z2z2 = sqr(z1), sqr(z2)
    u1, u2 = mul(a, z2z2), mul(c, z1z1)
    s1, s2 = mul(mul(b, z2), z2z2), mul(mul(e, z1), z1z1)
    h = sub(u2, u1, BIAS4, 4)
    return tail(u1, s1, h, sub(s2, s1, BIAS4, 4), mul(z1, z2)), h


# ---------------------------------------------------------------- plain arithmetic mod q on Montgomery (R = 2^384) values
RI = pow(R384, -1, Q)


def mm(a, b):
    return a * b * RI % Q


def ref_tail(U1, S1, H, rr, zfac):
    I = mm(2 * H % Q, 2 * H % Q)
    J = mm(H, I)
    V = mm(U1, I)
    X3 = (mm(rr, rr) - J - 2 * V) % Q
    Y3 = (mm(rr, (V - X3) % Q) - 2 * mm(S1, J)) % Q
    return X3, Y3, mm(zfac, H)


def ref_add_aff(x1, y1, x2, y2):
    return ref_tail(x1, y1, (x2 - x1) % Q, 2 * (y2 - y1) % Q, 2 * R384 % Q)


def ref_add_mixed(X1, Y1, Z1, x2, y2):
    zz = mm(Z1, Z1)
    U2, S2 = mm(x2, zz), mm(mm(y2, Z1), zz)
    return ref_tail(X1, Y1, (U2 - X1) % Q, 2 * (S2 - Y1) % Q, 2 * Z1 % Q)


def ref_add_jac(X1, Y1, Z1, X2, Y2, Z2):
    a, b = mm(Z1, Z1), mm(Z2, Z2)
    U1, U2 = mm(X1, b), mm(X2, a)
    S1, S2 = mm(mm(Y1, Z2), b), mm(mm(Y2, Z1), a)
    zfac = (mm((Z1 + Z2) % Q, (Z1 + Z2) % Q) - a - b) % Q
    return ref_tail(U1, S1, (U2 - U1) % Q, 2 * (S2 - S1) % Q, zfac)


def adversarial():
    out = [0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2, R384, Q - R384, (1 << 380), (1 << 380) - 1]
    ones = value([M28] * 13 + [0])
    out += [ones % Q, (ones >> 1) % Q, value([M28 if i % 2 else 0 for i in range(13)] + [0x1a010])]
    for i in (1, 5, 13):
        out += [((1 << (28 * i)) - 1) % Q, (1 << (28 * i)) % Q, ((1 << (28 * i)) + 1) % Q]
    return out


def test_load_store_round_trip_and_bounds():
    rng = random.Random(14)
    for x in adversarial() + [rng.randrange(Q) for _ in range(400)]:
        l = load(x)
        assert store(l) == x
        # a product of two loads and a constant one
        assert store(mul(l, limbs(R392))) == x
    # non-canonical memory values (anything below 2^384) still load to the right residue
    for x in (Q, Q + 1, (1 << 384) - 1, 2 * Q + 5):
        assert value(load(x)) % Q == (x << 8) % Q


def test_mul_sqr_sub_norm_match_plain_arithmetic():
    rng = random.Random(15)
    xs = adversarial() + [rng.randrange(Q) for _ in range(60)]
    for x in xs[:24]:
        for y in xs[:24]:
            a, b = load(x), load(y)
            assert store(mul(a, b)) == mm(x, y)
            assert store(sub(a, b, BIAS4, 4)) == (x - y) % Q
            assert store(norm(sub(shl(a, 2), norm(add(b, shl(b, 1))), BIAS4, 4))) == (4 * x - 3 * y) % Q
    for x in xs:
        assert store(sqr(load(x))) == mm(x, x)
    # the widest operands the formulas feed to a product: two unnormalised differences (limbs < 2^28 + 2^29), and one against 4 x a product
    fat = [M28 + b for b in BIAS4[:13]] + [BIAS4[13] + Q13 + 1]
    mul(fat, fat)
    mul(fat, [4 * M28] * 13 + [4 * (Q13 + 2)])


def test_the_three_addition_formulas_store_the_standard_coordinates():
    rng = random.Random(16)
    pool = adversarial()
    for it in range(120):
        pick = (lambda: rng.choice(pool)) if it < 40 else (lambda: rng.randrange(Q))
        x1, y1, z1, x2, y2, z2 = (pick() for _ in range(6))
        assert add_aff(x1, y1, x2, y2) == ref_add_aff(x1, y1, x2, y2)
        got, h = add_mixed(x1, y1, z1, x2, y2)
        assert got == ref_add_mixed(x1, y1, z1, x2, y2)
        got, h = add_jac(x1, y1, z1, x2, y2, z2)
        assert got == ref_add_jac(x1, y1, z1, x2, y2, z2)


def test_zero_prefilter_never_misses_h_equal_zero():
    rng = random.Random(17)
    for _ in range(60):
        x1, y1, z1, y2 = (rng.randrange(Q) for _ in range(4))
        # q = the affine form of p's x: U2 = x2 Z1^2 = X1  ->  H = 0 (mod q)
        zz = mm(z1, z1)
        x2 = mm(x1, mm(pow(zz * RI % Q, -1, Q) * R384 % Q, R384))
        assert mm(x2, zz) == x1
        _, h = add_mixed(x1, y1, z1, x2, y2)
        assert value(h) % Q == 0 and maybe_zero(h)
        # and for a generic pair it (almost surely) does not fire
        _, h = add_mixed(x1, y1, z1, rng.randrange(Q), y2)
        assert value(h) % Q != 0


def test_cells_stay_within_bounds_through_the_levels_of_a_tree():
    """eight affine inputs, three levels: level 0 affine + affine into cells, the cells added pairwise twice, one store at the end"""
    rng = random.Random(18)
    for it in range(12):
        pick = (lambda: rng.choice(adversarial())) if it < 4 else (lambda: rng.randrange(Q))
        pts = [(pick(), pick()) for _ in range(8)]
        cells, refs = [], []
        for k in range(0, 8, 2):
            (x1, y1), (x2, y2) = pts[k], pts[k + 1]
            a, b, c, e = load(x1), load(y1), load(x2), load(y2)
            cells.append(tail_cell(a, b, sub(c, a, BIAS4, 4), sub(e, b, BIAS4, 4), None))
            refs.append(ref_add_aff(x1, y1, x2, y2))
        while len(cells) > 1:
            cells = [add_cells(cells[k], cells[k + 1]) for k in range(0, len(cells), 2)]
            refs = [ref_add_jac(*refs[k], *refs[k + 1]) for k in range(0, len(refs), 2)]
        assert tuple(store(c) for c in cells[0]) == refs[0]


# ---------------------------------------------------------------- XYZZ cells (g1.hip.h: G1X14, the form the sum-by-key tree keeps)
def xyzz_tail(u1, s1, pp_, r_, zz12, zzz12):
    """g1x_tail: P = U2 - U1 + 4 q, R = S2 - S1 + 4 q (unnormalised); add-2008-s / mmadd-2008-s.  Returns the cell (X3, Y3, ZZ3, ZZZ3)"""
    PP, RR = sqr(pp_), sqr(r_)
    PPP, Qv = mul(pp_, PP), mul(u1, PP)
    t = norm(add(PPP, shl(Qv, 1)))
    x3 = norm(sub(RR, t, BIAS4, 4))
    w = sub(Qv, x3, BIAS16, 16)
    rw, sp = mul(r_, w), mul(s1, PPP)
    y3 = norm(sub(rw, sp, BIAS4, 4))
    zz3, zzz3 = (PP, PPP) if zz12 is None else (mul(zz12, PP), mul(zzz12, PPP))
    for c, bound in ((x3, 52), (y3, 52), (zz3, 11), (zzz3, 11)):
        assert all(l <= M28 for l in c[:13]) and value(c) * 10 < bound * Q   # limbs < 2^28; S <= 5.2 (X, Y), 1.1 (ZZ, ZZZ)
    return x3, y3, zz3, zzz3


def xyzz_add_aff(x1, y1, x2, y2):
    a, b, c, e = load(x1), load(y1), load(x2), load(y2)
    return xyzz_tail(a, b, sub(c, a, BIAS4, 4), sub(e, b, BIAS4, 4), None, None)


def xyzz_add(p, q):
    (x1, y1, zz1, zzz1), (x2, y2, zz2, zzz2) = p, q
    u1, u2 = mul(x1, zz2), mul(x2, zz1)
    s1, s2 = mul(y1, zzz2), mul(y2, zzz1)
    return xyzz_tail(u1, s1, sub(u2, u1, BIAS4, 4), sub(s2, s1, BIAS4, 4), mul(zz1, zz2), mul(zzz1, zzz2))


def xyzz_from_jac(X, Y, Z):
    z = load(Z)
    zz = sqr(z)
    return load(X), load(Y), zz, mul(z, zz)


def xyzz_to_jac(c):
    """(X ZZ, Y ZZZ, ZZ): a Jacobian representative with Z' = ZZ (x = X / ZZ = X' / Z'^2, y = Y / ZZZ = Y' / Z'^3 as ZZ^3 = ZZZ^2)"""
    x, y, zz, zzz = c
    return store(mul(x, zz)), store(mul(y, zzz)), store(zz)


def jac_affine(X, Y, Z):
    """Montgomery-domain Jacobian -> plain affine integers"""
    x, y, z = (v * RI % Q for v in (X, Y, Z))
    zi = pow(z, -1, Q)
    return x * zi * zi % Q, y * zi * zi * zi % Q


def test_xyzz_cells_give_the_same_point_and_stay_within_bounds():
    """eight affine inputs, three levels in the XYZZ form against the Jacobian reference formulas: the same affine point; a Jacobian
    source converted into a cell and added; the widest operands of every product asserted inside mul / sqr / sub"""
    rng = random.Random(19)
    for it in range(14):
        pick = (lambda: rng.choice(adversarial())) if it < 4 else (lambda: rng.randrange(Q))
        pts = [(pick(), pick()) for _ in range(8)]
        cells, refs = [], []
        for k in range(0, 8, 2):
            (x1, y1), (x2, y2) = pts[k], pts[k + 1]
            cells.append(xyzz_add_aff(x1, y1, x2, y2))
            refs.append(ref_add_aff(x1, y1, x2, y2))
        while len(cells) > 1:
            cells = [xyzz_add(cells[k], cells[k + 1]) for k in range(0, len(cells), 2)]
            refs = [ref_add_jac(*refs[k], *refs[k + 1]) for k in range(0, len(refs), 2)]
        if any(v == 0 for v in refs[0][2:]) or it < 4:
            continue      # adversarial field elements are not curve points: a zero denominator may occur; the bounds were still checked
        assert jac_affine(*xyzz_to_jac(cells[0])) == jac_affine(*refs[0])
        # a Jacobian source (wire form) as a cell, added to the tree's result
        X, Y, Z = (rng.randrange(Q) for _ in range(3))
        got = xyzz_add(cells[0], xyzz_from_jac(X, Y, Z))
        assert jac_affine(*xyzz_to_jac(got)) == jac_affine(*ref_add_jac(*refs[0], X, Y, Z))
    # infinity: Z = 0 gives ZZ = ZZZ = 0 limb for limb (the test the device uses)
    c = xyzz_from_jac(5, 7, 0)
    assert all(l == 0 for l in c[2]) and all(l == 0 for l in c[3])

