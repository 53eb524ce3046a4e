"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Python big-int restatement of the Knuckles opening, the last step of Pippenger::prove's "open" span:

  * KnucklesProvingKey::new (inverses), compute_t                 src/commitments/knuckles.rs:64-82, 111-154
  * KzgProvingKey::{commit, open}, div_by_linear, ev              src/commitments/kzg.rs:73-81, 123-132, 142-150
  * KzgVerifyingKey::verify_reduce_to_pair                        src/commitments/kzg.rs:46-59
  * KnucklesOpeningProtocol::prove, and the verifier's algebraic check   src/cleanup/protocols/opening.rs:39-98, 100-148
Pairings are not restated: tests build the SRS from a known tau and check the KZG equations in the exponent.
"""
from . import g1 as G
from .field import P


def setup_inverses(k, num_vars):
    """knuckles.rs:64-82: inverses of k^s - k^(N-1), with 1 at s = N-1"""
    n = 1 << num_vars
    pows = [pow(k, i, P) for i in range(2 * n - 1)]
    kn = pows[n - 1]
    v = [(x - kn) % P for x in pows]
    v[n - 1] = (v[n - 1] + 1) % P
    return [pow(x, -1, P) for x in v]


def compute_t(poly, point, num_vars, inverses):
    """knuckles.rs:111-154"""
    assert len(point) == num_vars
    pt = list(reversed(point))
    n = 1 << num_vars
    assert len(poly) <= n
    pt_rev = [(1 - x) % P for x in pt]
    t = (list(poly) + [0] * (2 * n - 1))[:2 * n - 1]
    t_scaled = [0] * (2 * n - 1)
    curr = n
    for i in range(num_vars):
        for idx in range(curr):
            t_scaled[idx] = t[idx] * pt_rev[i] % P
        off = 1 << i
        curr += off
        for idx in range(curr):
            t[idx] = (t[idx] - t_scaled[idx] + (t_scaled[idx - off] if idx >= off else 0)) % P
    opening = t[n - 1]
    t[n - 1] = 0
    return [a * b % P for a, b in zip(t, inverses)], opening


def ev(poly, x):
    acc, power = 0, 1
    for c in poly:
        acc = (acc + c * power) % P
        power = power * x % P
    return acc


def div_by_linear(poly, pt):
    """kzg.rs:73-81"""
    q = [0] * (len(poly) - 1)
    rem = poly[-1]
    for i in range(len(q) - 1, -1, -1):
        q[i] = rem
        rem = (poly[i] + rem * pt) % P
    return q, rem


def kzg_commit(basis, poly):
    return G.naive_msm(basis[:len(poly)], poly)


def kzg_open(basis, poly, pt):
    q, r = div_by_linear(poly, pt)
    return kzg_commit(basis, q), r


def verify_reduce_to_pair(g0, poly_comm, quot_comm, at, opening):
    """kzg.rs:46-59: ([Q] * a - g0 * b + [P], [Q])"""
    a = G.add(G.add(G.mul(quot_comm, at), G.neg(G.mul(g0, opening))), poly_comm)
    return a, quot_comm


def knuckles_open(tr, points_out, basis, inverses, k, num_vars, commitment, point, claimed_ev, poly):
    """opening.rs:39-98.  tr: TapeTranscript (scalars); points_out: list collecting the G1 points written, in order.
    Returns ((A, B), proof dict)."""
    t, opening = compute_t(poly, point, num_vars, inverses)
    assert opening == claimed_ev
    def wp(pt_):   # write_points::<G1> in transcript order when the transcript records points
        points_out.append(pt_)
        if hasattr(tr, "write_points"):
            tr.write_points([pt_])
    t_comm = kzg_commit(basis, t)
    wp(t_comm)
    x = tr.challenge(128)
    kx = x * k % P
    t_x, p_x = ev(t, x), ev(poly, x)
    tr.write_scalars([t_x, p_x])
    lam = tr.challenge(128)
    padded = list(poly) + [0] * (len(t) - len(poly))
    p_lt = [(lam * b + a) % P for a, b in zip(padded, t)]
    p_lt_x_proof, _ = kzg_open(basis, p_lt, x)
    wp(p_lt_x_proof)
    t_kx_proof, t_kx = kzg_open(basis, t, kx)
    tr.write_scalars([t_kx])
    wp(t_kx_proof)
    fin = tr.challenge(128)
    p_lt_comm = G.add(G.mul(t_comm, lam), commitment)
    p_lt_open = (t_x * lam + p_x) % P
    a0, b0 = verify_reduce_to_pair(basis[0], p_lt_comm, p_lt_x_proof, x, p_lt_open)
    a1, b1 = verify_reduce_to_pair(basis[0], t_comm, t_kx_proof, kx, t_kx)
    pair = (G.add(a0, G.mul(a1, fin)), G.add(b0, G.mul(b1, fin)))
    return pair, dict(t_comm=t_comm, t_x=t_x, p_x=p_x, p_lt_x_proof=p_lt_x_proof, t_kx=t_kx, t_kx_proof=t_kx_proof, x=x, lam=lam,
                      fin=fin)


def verifier_identity(k, num_vars, point, claimed_ev, proof):
    """opening.rs:126-145: x (T(kx) - k^(N-1) T(x)) + x^N claim == x P(x) Eq_point(x)"""
    x = proof["x"]
    k_pow = pow(k, (1 << num_vars) - 1, P)
    xpow, eq_ev = x, 1
    for i in range(num_vars):
        r = point[num_vars - i - 1]
        eq_ev = eq_ev * ((r + (1 - r) * xpow) % P) % P
        xpow = xpow * xpow % P
    lhs = (x * (proof["t_kx"] - k_pow * proof["t_x"]) + xpow * claimed_ev) % P
    return lhs == x * proof["p_x"] % P * eq_ev % P
