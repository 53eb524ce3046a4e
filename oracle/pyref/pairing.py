"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

BLS12-381 pairing in Python big-int arithmetic, the check behind KzgVerifyingKey::verify_pair
(/root/reference/src/commitments/kzg.rs:61-67).  The reference takes it from ark-bls12-381 0.4.0 / ark-ec 0.4.2 (not
vendored); this is an independent restatement of the published construction in a different representation from the library's
(gkr_msm_amd/csrc/pairing.hpp): Fq12 = Fq[w] / (w^12 - 2 w^6 + 2) as plain polynomials, u = w^6 - 1, and the whole final
exponent (q^12 - 1) / r applied by one square-and-multiply.  tower_to_poly converts the library's tower coordinates, so the two
implementations are compared value for value.
"""
from .g1 import Q

R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
X_ABS = 0xd201000000010000

G2_GEN = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
           0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
          (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
           0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))


# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2 + 1), elements (a, b)
def f2_add(x, y): return ((x[0] + y[0]) % Q, (x[1] + y[1]) % Q)
def f2_sub(x, y): return ((x[0] - y[0]) % Q, (x[1] - y[1]) % Q)
def f2_mul(x, y): return ((x[0] * y[0] - x[1] * y[1]) % Q, (x[0] * y[1] + x[1] * y[0]) % Q)
def f2_inv(x):
    n = pow(x[0] * x[0] + x[1] * x[1], -1, Q)
    return (x[0] * n % Q, -x[1] * n % Q)


B2 = (4, 4)   # 4 (1 + u)


def g2_on_curve(p):
    if p is None:
        return True
    x, y = p
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), B2)


def g2_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if p[1] != q[1] or p[1] == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(p[0], p[0])), f2_inv(f2_mul((2, 0), p[1])))
    else:
        lam = f2_mul(f2_sub(q[1], p[1]), f2_inv(f2_sub(q[0], p[0])))
    x = f2_sub(f2_sub(f2_mul(lam, lam), p[0]), q[0])
    return (x, f2_sub(f2_mul(lam, f2_sub(p[0], x)), p[1]))


def g2_mul(p, k):
    acc = None
    for bit in bin(k)[2:]:
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, p)
    return acc


def g2_neg(p):
    return None if p is None else (p[0], ((-p[1][0]) % Q, (-p[1][1]) % Q))


# ---------------------------------------------------------------- Fq12 = Fq[w]/(w^12 - 2 w^6 + 2), 12 coefficients
def f12(c0=0):
    return [c0 % Q] + [0] * 11


ONE12 = f12(1)


def f12_mul(x, y):
    t = [0] * 23
    for i, a in enumerate(x):
        if a:
            for j, b in enumerate(y):
                t[i + j] += a * b
    for d in range(22, 11, -1):   # w^12 = 2 w^6 - 2
        c = t[d]
        if c:
            t[d - 6] += 2 * c
            t[d - 12] -= 2 * c
    return [v % Q for v in t[:12]]


def f12_sub(x, y):
    return [(a - b) % Q for a, b in zip(x, y)]


def _poly_divmod(a, b):
    a = list(a)
    db = len(b) - 1
    inv = pow(b[-1], -1, Q)
    out = [0] * max(len(a) - db, 1)
    for i in range(len(a) - 1, db - 1, -1):
        c = a[i] * inv % Q
        out[i - db] = c
        if c:
            for j in range(db + 1):
                a[i - db + j] = (a[i - db + j] - c * b[j]) % Q
    r = a[:db]
    while r and r[-1] == 0:
        r.pop()
    return out, r


def f12_inv(x):
    """extended Euclid on polynomials over Fq"""
    mod = [2, 0, 0, 0, 0, 0, Q - 2, 0, 0, 0, 0, 0, 1]
    lm, hm = [1], [0]
    low = list(x)
    while low and low[-1] == 0:
        low.pop()
    high = mod
    while len(low) > 1:
        qt, rem = _poly_divmod(high, low)
        # nm = hm - qt * lm
        prod = [0] * (len(qt) + len(lm))
        for i, a in enumerate(qt):
            for j, b in enumerate(lm):
                prod[i + j] = (prod[i + j] + a * b) % Q
        n = max(len(prod), len(hm))
        nm = [((hm[i] if i < len(hm) else 0) - (prod[i] if i < len(prod) else 0)) % Q for i in range(n)]
        high, hm, low, lm = low, lm, rem, nm
    c = pow(low[0], -1, Q)
    out = [v * c % Q for v in lm] + [0] * 12
    return out[:12]


def f12_pow(x, e):
    acc = ONE12
    for bit in bin(e)[2:]:
        acc = f12_mul(acc, acc)
        if bit == "1":
            acc = f12_mul(acc, x)
    return acc


def _embed_fq2(c, shift):
    """(a + b u) w^shift with u = w^6 - 1, shift < 6"""
    out = [0] * 12
    out[shift] = (c[0] - c[1]) % Q
    out[shift + 6] = c[1] % Q
    return out


def tower_to_poly(coeffs):
    """12 Fq coordinates in the library's tower order -- Fq12 = (Fq6 a, Fq6 b) with a + b w, Fq6 = (Fq2 a, b, c) with a + b v + c v^2,
    v = w^2, Fq2 = (a, b) with a + b u -- to the polynomial basis"""
    out = [0] * 12
    for wi in range(2):
        for vj in range(3):
            base = 6 * wi + 2 * vj
            c = (coeffs[base], coeffs[base + 1])
            e = _embed_fq2(c, 2 * vj + wi)
            out = [(a + b) % Q for a, b in zip(out, e)]
    return out


def untwist(q):
    """(x', y') on the twist -> (x' / w^2, y' / w^3) on E(Fq12)"""
    w2_inv = f12_inv([0, 0, 1] + [0] * 9)
    w3_inv = f12_inv([0, 0, 0, 1] + [0] * 8)
    return f12_mul(_embed_fq2(q[0], 0), w2_inv), f12_mul(_embed_fq2(q[1], 0), w3_inv)


def miller_loop(p, q):
    qx, qy = untwist(q)
    px, py = f12(p[0]), f12(p[1])
    tx, ty = qx, qy
    f = ONE12

    def line(ax, ay, lam):
        return f12_sub(f12_sub(py, ay), f12_mul(lam, f12_sub(px, ax)))

    def step(ax, ay, bx, lam):
        x = f12_sub(f12_sub(f12_mul(lam, lam), ax), bx)
        return x, f12_sub(f12_mul(lam, f12_sub(ax, x)), ay)
    for bit in bin(X_ABS)[3:]:
        lam = f12_mul(f12_mul(f12(3), f12_mul(tx, tx)), f12_inv(f12_mul(f12(2), ty)))
        f = f12_mul(f12_mul(f, f), line(tx, ty, lam))
        tx, ty = step(tx, ty, tx, lam)
        if bit == "1":
            lam = f12_mul(f12_sub(qy, ty), f12_inv(f12_sub(qx, tx)))
            f = f12_mul(f, line(tx, ty, lam))
            tx, ty = step(tx, ty, qx, lam)
    return f12_pow(f, Q ** 6)   # conjugation (x < 0), the slow honest way


def pairing(p, q):
    """e(P, Q) for affine P in G1 (x, y) and Q in G2 ((x0, x1), (y0, y1)); None = infinity"""
    if p is None or q is None:
        return ONE12
    return f12_pow(miller_loop(p, q), (Q ** 12 - 1) // R_ORDER)
