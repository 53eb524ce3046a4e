"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Python big-int restatement of the BLS12-381 G1 side of the hot path:

  * msm_bigint_wnaf_nonaff / msm_bigint_nonaff / make_digits / ln_without_floats   src/msm_nonaffine.rs:89-323
    (ark-ec 0.4.2 `Projective<g1::Config>` has NEGATION_IS_CHEAP = true, so the reference takes the wNAF branch
    msm_nonaffine.rs:45-46; both branches are restated)
  * binary_msm, prepare_chunk, prepare_bases, prepare_coefs, into_u8               src/binary_msm.rs:13-53
  * the G1 part of PushForwardState::new (d_outer / c_outer buckets, c_comm, d_comm, KZG commits of p_0, p_1,
    ac_c, ac_d) and of second_phase (c_pull / d_pull commitments)                  src/cleanup/protocols/pushforward/pushforward.rs:395-456, 504-533, 596-605
  * KzgProvingKey::commit                                                          src/commitments/kzg.rs:123-126
  * Pullback::{values, bucketed_msm}                                               src/pullback.rs:15-59

The arithmetic lives in un-vendored dependencies (ark-ff / ark-ec 0.4.2, ark-bls12-381 0.4.0, Cargo.lock); the curve is
the public BLS12-381 G1: y^2 = x^3 + 4 over Fq, prime-order subgroup of order r = Fr modulus, cofactor
0x396c8c005555e1568c00aaab0000aaab.  Points here are affine (x, y) integer pairs or None (infinity): `Projective`
equality in ark-ec is equality of the represented group element, which is what every output below is compared by.
"""
from . import field as F

Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
B = 4
R_ORDER = F.P  # G1 scalar field = BLS12-381 Fr
COFACTOR = 0x396C8C005555E1568C00AAAB0000AAAB
MONT_R = (1 << 384) % Q
MONT_R_INV = pow(MONT_R, -1, Q)

# the standard generator of the prime-order subgroup (ark-bls12-381 g1::G1_GENERATOR_X / _Y; also the IETF / zkcrypto one)
GEN = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
       0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)


def on_curve(p):
    if p is None:
        return True
    x, y = p
    return (y * y - x * x * x - B) % Q == 0


def neg(p):
    if p is None:
        return None
    return (p[0], (-p[1]) % Q)


def add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


def double(p):
    return add(p, p)


def mul(p, k):
    k %= R_ORDER
    acc = None
    while k:
        if k & 1:
            acc = add(acc, p)
        p = double(p)
        k >>= 1
    return acc


def random_points(n, seed):
    """n points of the prime-order subgroup, k_i * GEN with seeded k_i (distinct, non-trivial)"""
    rng = F.SplitMix64(seed)
    cur = mul(GEN, rng.next_fr() | 1)
    step = mul(GEN, rng.next_fr() | 1)
    out = []
    for _ in range(n):
        out.append(cur)
        cur = add(cur, step)
        if rng.next() & 3 == 0:
            step = double(step)
    return out


def naive_msm(bases, scalars):
    acc = None
    for b, s in zip(bases, scalars):
        acc = add(acc, mul(b, s))
    return acc


# ------------------------------------------------------------------ msm_nonaffine.rs
def ln_without_floats(a):
    """msm_nonaffine.rs:319-322; ark_std::log2(a) = ceil(log2(a)) (0 for a <= 1)"""
    lg = 0 if a <= 1 else (a - 1).bit_length()
    return lg * 69 // 100


def window_size(size):
    return 3 if size < 32 else ln_without_floats(size) + 2


def max_num_bits(bigints):
    """msm_nonaffine.rs:93-104 / :190-201: exact bit length if every scalar is <= 60 bits, else MODULUS_BIT_SIZE"""
    m = 1
    for b in bigints:
        if b.bit_length() > m:
            m = b.bit_length()
        if m > 60:
            return 255
    return m


def make_digits(a, w, num_bits):
    """msm_nonaffine.rs:275-314 (a: canonical integer < 2^256, read as 4 u64 limbs)"""
    limbs = F.int_to_limbs(a, 4)
    radix = 1 << w
    mask = radix - 1
    carry = 0
    if num_bits == 0:
        num_bits = a.bit_length()
    count = (num_bits + w - 1) // w
    digits = [0] * count
    for i in range(count):
        off = i * w
        ui, bi = off // 64, off % 64
        if bi < 64 - w or ui == len(limbs) - 1:
            buf = limbs[ui] >> bi
        else:
            buf = ((limbs[ui] >> bi) | (limbs[ui + 1] << (64 - bi))) & 0xFFFFFFFFFFFFFFFF
        coef = carry + (buf & mask)
        carry = (coef + radix // 2) >> w
        digits[i] = coef - (carry << w)
    digits[count - 1] += carry << w
    return digits


def _combine_windows(window_sums, c):
    lowest = window_sums[0]
    total = None
    for s in reversed(window_sums[1:]):
        total = add(total, s)
        for _ in range(c):
            total = double(total)
    return add(lowest, total)


def msm_bigint_wnaf_nonaff(bases, bigints):
    """msm_nonaffine.rs:89-161"""
    size = min(len(bases), len(bigints))
    nb = max_num_bits(bigints)
    bases, scalars = bases[:size], bigints[:size]
    c = window_size(size)
    count = (nb + c - 1) // c
    digs = [make_digits(s, c, nb) for s in scalars]
    sums = []
    for i in range(count):
        buckets = [None] * (1 << c)
        for d, base in zip(digs, bases):
            s = d[i]
            if s > 0:
                buckets[s - 1] = add(buckets[s - 1], base)
            elif s < 0:
                buckets[-s - 1] = add(buckets[-s - 1], neg(base))
        run, res = None, None
        for b in reversed(buckets):
            run = add(run, b)
            res = add(res, run)
        sums.append(res)
    return _combine_windows(sums, c)


def msm_bigint_nonaff(bases, bigints):
    """msm_nonaffine.rs:164-272 (the unsigned-window branch)"""
    size = min(len(bases), len(bigints))
    nb = max_num_bits(bigints)
    bases, scalars = bases[:size], bigints[:size]
    c = window_size(size)
    sums = []
    for w_start in range(0, nb, c):
        res = None
        buckets = [None] * ((1 << c) - 1)
        for s, base in zip(scalars, bases):
            if s == 0:
                continue
            if s == 1:
                if w_start == 0:
                    res = add(res, base)
                continue
            d = ((s >> w_start) & 0xFFFFFFFFFFFFFFFF) % (1 << c)
            if d:
                buckets[d - 1] = add(buckets[d - 1], base)
        run = None
        for b in reversed(buckets):
            run = add(run, b)
            res = add(res, run)
        sums.append(res)
    return _combine_windows(sums, c)


def msm_nonaff(bases, scalars):
    """VariableBaseMsmNonaffine::msm_nonaff (msm_nonaffine.rs:34-50): length check, into_bigint, wNAF branch"""
    assert len(bases) == len(scalars)
    return msm_bigint_wnaf_nonaff(bases, [s % R_ORDER for s in scalars])


# ------------------------------------------------------------------ binary_msm.rs
def into_u8(bits):
    s = 0
    for b in bits[:8]:
        s = (s << 1) + (1 if b else 0)
    return s


def prepare_coefs(bits, gamma):
    return [into_u8(bits[i:i + gamma]) for i in range(0, len(bits), gamma)]


def prepare_chunk(chunk, gamma):
    out = []
    rev = list(reversed(chunk))
    for i in range(1, 1 << gamma):
        acc = None
        for idx in range(min(gamma, len(rev))):
            if (1 << idx) & i:
                acc = add(acc, rev[idx])
        out.append(acc)
    return out


def prepare_bases(bases, gamma):
    return [prepare_chunk(bases[i:i + gamma], gamma) for i in range(0, len(bases), gamma)]


def binary_msm(coefs, tables):
    assert len(coefs) == len(tables)
    acc = None
    for c, t in zip(coefs, tables):
        if c:
            acc = add(acc, t[c - 1])
    return acc


# ------------------------------------------------------------------ pushforward.rs (G1 part) / kzg.rs / pullback.rs
def kzg_commit(basis, poly):
    """KzgProvingKey::commit (kzg.rs:123-126): <G1 as VariableBaseMSM>::msm(&ptau_1[..len], poly)"""
    assert len(poly) <= len(basis)
    return naive_msm(basis[:len(poly)], poly)


def running_sum_reduce(buckets):
    """pushforward.rs:504-524: acc = sum_{i<len-1} running(len-1-i) = sum_i i * bucket_i"""
    acc, run = None, None
    n = len(buckets)
    for i in range(n - 1):
        run = add(run, buckets[n - i - 1])
        acc = add(acc, run)
    return acc


def pushforward_outer(digits, counter, basis, x_logsize, d_logsize, clm):
    """pushforward.rs:395-456 + 504-524.  digits/counter: [y][x]; basis: kzg_basis (>= 2^(x_logsize + clm) points).
    Returns (d_outer_buckets, c_outer_buckets, d_comm, c_comm), one entry per matrix commitment."""
    y_size = len(digits)
    x_size = 1 << x_logsize
    comm_mul = 1 << clm
    d_rows, c_rows, ub = [], [], []
    for y in range(y_size):
        db = [None] * (1 << d_logsize)
        cb = {}
        max_c = 0
        for x in range(x_size):
            d, c = digits[y][x], counter[y][x]
            max_c = max(max_c, c)
            pt = basis[x + x_size * (y % comm_mul)]
            db[d] = add(db[d], pt)
            cb[c] = add(cb.get(c), pt)
        d_rows.append(db)
        c_rows.append(cb)
        ub.append(max_c + 1)
    d_outer, c_outer = [], []
    for m in range(0, y_size, comm_mul):
        chunk = range(m, min(m + comm_mul, y_size))
        d_outer.append([_sum(d_rows[y][i] for y in chunk) for i in range(1 << d_logsize)])
        mc = max(ub[y] for y in chunk)
        c_outer.append([_sum(c_rows[y].get(i) for y in chunk) for i in range(mc)])
    d_comm = [running_sum_reduce(b) for b in d_outer]
    c_comm = [running_sum_reduce(b) for b in c_outer]
    return d_outer, c_outer, d_comm, c_comm


def _sum(it):
    acc = None
    for p in it:
        acc = add(acc, p)
    return acc


def second_phase_comms(d_outer, c_outer, eq_d, eq_c):
    """pushforward.rs:596-605: msm_nonaff(outer buckets, eq tables)"""
    d_pull = [msm_nonaff(b, eq_d) for b in d_outer]
    c_pull = [msm_nonaff(b, eq_c[:len(b)]) for b in c_outer]
    return d_pull, c_pull


def pullback_values(mapping, image):
    return [image[i] for i in mapping]


def pullback_bucketed_msm(mapping, image, bases):
    """pullback.rs:27-59"""
    assert len(mapping) == len(bases)
    buckets = [None] * len(image)
    for b, m in zip(bases, mapping):
        buckets[m] = add(buckets[m], b)
    return msm_nonaff(buckets, image)
