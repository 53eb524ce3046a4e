"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Python big-int restatement of the field / curve layer of morgana-proofs/GKR-MSM.
Everything here works on canonical integers mod p (NOT Montgomery form); the
helpers at the bottom convert to/from the in-memory Montgomery limbs that the
Rust reference (ark-ff 0.4.2 `Fp<MontBackend<_,4>>`) and our C ABI use.

Reference citations (relative to /root/reference):
  * Fr = BLS12-381 scalar field, used as Bandersnatch base field
    (src/utils.rs:22-49, `TwistedEdwardsConfig for Fr`).
  * COEFF_D Montgomery limbs KAT: src/utils.rs:35.
  * mul_by_a(x) = -(4x + x) = -5x : src/utils.rs:40-43.
"""

# BLS12-381 scalar field modulus (ark-bls12-381 0.4.0 Fr).
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P          # Montgomery radix
R_INV = pow(R, -1, P)

# Bandersnatch (ark-ed-on-bls12-381-bandersnatch 0.4.0): a*x^2 + y^2 = 1 + d*x^2*y^2
TE_A = P - 5
TE_D = 0x6389C12633C267CBC66E3BF86BE3B6D8CB66677177E54F92B369F2F5188D58E7
# prime-order subgroup size (= Bandersnatch ScalarField modulus) and cofactor
BS_ORDER = 0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1
BS_COFACTOR = 4
BS_R = (1 << 256) % BS_ORDER

# KAT from the reference: src/utils.rs:35  (d * 2^256 mod p as 4 x u64 LE limbs)
COEFF_D_MONT_LIMBS = [12167860994669987632, 4043113551995129031,
                      6052647550941614584, 3904213385886034240]


def limbs_to_int(limbs):
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


def int_to_limbs(x, n=4):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def to_mont(x):
    return (x * R) % P


def from_mont(x):
    return (x * R_INV) % P


def inv(x):
    return pow(x, -1, P)


def mul_by_a(x):
    # src/utils.rs:40-43 : t = x.double().double(); -(t + x)
    return (-(5 * x)) % P


def mul_by_d(x):
    return (x * TE_D) % P


# ---------------------------------------------------------------- curve group law
# Used only to cross-check the layer formulas (reference test Pattern C,
# src/cleanup/protocols/gkrs/bintree_add.rs:507-638) and to build inputs.

def te_on_curve(x, y):
    return (TE_A * x * x + y * y - 1 - TE_D * x * x % P * y * y) % P == 0


def te_add_affine(p, q):
    """Unified affine twisted-Edwards addition (textbook law)."""
    x1, y1 = p
    x2, y2 = q
    k = TE_D * x1 % P * x2 % P * y1 % P * y2 % P
    x3 = (x1 * y2 + x2 * y1) % P * inv((1 + k) % P) % P
    y3 = (y1 * y2 - TE_A * x1 % P * x2) % P * inv((1 - k) % P) % P
    return (x3, y3)


def te_double_affine(p):
    return te_add_affine(p, p)


def te_neg(p):
    return ((-p[0]) % P, p[1])


def te_mul_affine(p, k):
    acc = (0, 1)
    base = p
    while k:
        if k & 1:
            acc = te_add_affine(acc, base)
        base = te_add_affine(base, base)
        k >>= 1
    return acc


def proj_to_affine(X, Y, Z):
    zi = inv(Z)
    return (X * zi % P, Y * zi % P)


def sqrt_mod_p(a):
    """Tonelli-Shanks in Fr (p = 1 mod 2^32)."""
    a %= P
    if a == 0:
        return 0
    if pow(a, (P - 1) // 2, P) != 1:
        return None
    s, q = 0, P - 1
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 5
    while pow(z, (P - 1) // 2, P) != P - 1:
        z += 1
    m, c, t, r = s, pow(z, q, P), pow(a, q, P), pow(a, (q + 1) // 2, P)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % P
            i += 1
        b = pow(c, 1 << (m - i - 1), P)
        m, c = i, b * b % P
        t, r = t * c % P, r * b % P
    return r


def te_point_from_x(x):
    """Solve a x^2 + y^2 = 1 + d x^2 y^2 for y; None if no solution."""
    x %= P
    num = (1 - TE_A * x * x) % P
    den = (1 - TE_D * x * x) % P
    if den == 0:
        return None
    y = sqrt_mod_p(num * inv(den) % P)
    if y is None:
        return None
    return (x, y)


class SplitMix64:
    """Deterministic stream shared by oracle, tests and bench (seed 'GKRMSM')."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def next_bits(self, nbits):
        v = 0
        for i in range((nbits + 63) // 64):
            v |= self.next() << (64 * i)
        return v & ((1 << nbits) - 1)

    def next_fr(self):
        return self.next_bits(512) % P


def find_subgroup_generator():
    """Smallest-x point cleared of the cofactor; order check included."""
    x = 1
    while True:
        pt = te_point_from_x(x)
        if pt is not None:
            g = te_mul_affine(pt, BS_COFACTOR)
            if g != (0, 1) and te_mul_affine(g, BS_ORDER) == (0, 1):
                return g
        x += 1


_GEN = None


def generator():
    global _GEN
    if _GEN is None:
        _GEN = find_subgroup_generator()
    return _GEN


def random_points(n, seed):
    """n points of the prime-order subgroup: P_i = k_i * G (k_i 64-bit from SplitMix64)."""
    rng = SplitMix64(seed)
    g = generator()
    # incremental: P_0 = k0*G, then P_{i+1} = P_i + (k_i mod 2^16 + 1) * G via small table
    table = [(0, 1)]
    for _ in range(256):
        table.append(te_add_affine(table[-1], g))
    cur = te_mul_affine(g, rng.next() | 1)
    out = []
    for _ in range(n):
        out.append(cur)
        cur = te_add_affine(cur, table[1 + (rng.next() & 0xFF)])
    return out


def random_scalars(n, nbits, seed):
    """mirrors build_pippenger_data: from_le_bytes_mod_order(first nbits/8 bytes)
    (src/cleanup/protocols/pippenger.rs:463-466); reduced mod the Bandersnatch order."""
    rng = SplitMix64(seed ^ 0x5CA1A125)
    return [rng.next_bits(256) % (1 << (8 * (nbits // 8))) % BS_ORDER for _ in range(n)]
