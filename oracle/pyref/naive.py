"""TEST INFRASTRUCTURE (oracle): the reference's NAIVE sumcheck object, restated on its own.

`ExampleSumcheckObjectSO` (/root/reference/src/cleanup/protocols/sumcheck.rs:132-235, "Not parallelized, not optimized, dumb as
rock.  Use to test agreement with other protocols.") is what the reference's own tests hold the optimised round polynomials
against (`check_univars`, sumchecks/dense_eq.rs:258-344 and sumchecks/vecvec_eq.rs:511-600).  It shares nothing with the
optimised objects of pyref/sumcheck.py: plain lists, the whole function (eq column included) evaluated at 0, 1, ..., deg on every
pair, interpolation by solving the Vandermonde system with Gaussian elimination (liblasso's UniPoly::from_evals does the same,
un-vendored dependency liblasso @ 925a7a74) -- not the Lagrange construction pyref/sumcheck.py uses."""
from .field import P


def interpolate_gauss(evals):
    """coefficients c_0..c_d of the polynomial with p(i) = evals[i], i = 0..d: Gaussian elimination on the Vandermonde system"""
    n = len(evals)
    m = [[pow(i, j, P) for j in range(n)] + [evals[i] % P] for i in range(n)]
    for col in range(n):
        piv = next(r for r in range(col, n) if m[r][col])
        m[col], m[piv] = m[piv], m[col]
        inv = pow(m[col][col], P - 2, P)
        m[col] = [v * inv % P for v in m[col]]
        for r in range(n):
            if r != col and m[r][col]:
                f = m[r][col]
                m[r] = [(a - f * b) % P for a, b in zip(m[r], m[col])]
    return [m[i][n] for i in range(n)]


def evaluate(coeffs, x):
    acc, xp = 0, 1
    for c in coeffs:
        acc = (acc + c * xp) % P
        xp = xp * x % P
    return acc


class ExampleSumcheckObjectSO:
    """sumcheck.rs:132-235.  f: any object with exec(list) -> field element, deg, n_ins (single output)."""

    def __init__(self, polys, f, num_vars):
        assert len(polys) == f.n_ins                           # :153
        for p in polys:
            assert len(p) == 1 << num_vars                     # :155
        self.polys = [[v % P for v in p] for p in polys]
        self.f, self.num_vars, self.round_idx = f, num_vars, 0
        self.cached, self.challenges = None, []

    def claim(self):
        """:160-162"""
        n = 1 << (self.num_vars - self.round_idx)
        return sum(self.f.exec([p[i] for p in self.polys]) for i in range(n)) % P

    def unipoly(self):
        """:184-227: acc[0] at the even element, acc[1] at the odd one, acc[s] at odd + (s - 1) (odd - even)"""
        assert self.round_idx < self.num_vars, "the protocol has already ended"
        if self.cached is None:
            half = 1 << (self.num_vars - self.round_idx - 1)
            deg = self.f.deg
            acc = [0] * (deg + 1)
            for i in range(half):
                lo = [p[2 * i] for p in self.polys]
                hi = [p[2 * i + 1] for p in self.polys]
                acc[0] = (acc[0] + self.f.exec(lo)) % P
                acc[1] = (acc[1] + self.f.exec(hi)) % P
                dif = [(h - l) % P for h, l in zip(hi, lo)]
                args = list(hi)
                for s in range(2, deg + 1):
                    args = [(a + d) % P for a, d in zip(args, dif)]
                    acc[s] = (acc[s] + self.f.exec(args)) % P
            self.cached = interpolate_gauss(acc)
        return list(self.cached)

    def bind(self, t):
        """:166-181 (bind_dense_poly :165)"""
        assert self.round_idx < self.num_vars, "the protocol has already ended"
        assert self.cached is not None, "should evaluate unipoly before binding"
        self.challenges.append(t)
        self.polys = [[(p[2 * i] + t * (p[2 * i + 1] - p[2 * i])) % P for i in range(len(p) // 2)] for p in self.polys]
        self.round_idx += 1
        self.cached = None

    def final_evals(self):
        assert self.round_idx == self.num_vars, "can only call final evals after the last round"
        return [p[0] for p in self.polys]


class GammaEq:
    """EqWrapper::new(GammaWrapper::new(f, gamma)) (sumcheck.rs:706-740, 803-829) written out for the naive object:
    (sum_o gamma^o f_o(args[:-1])) * args[-1]"""

    def __init__(self, f, gamma):
        self.f, self.gamma = f, gamma % P
        self.deg, self.n_ins = f.deg + 1, f.n_ins + 1

    def exec(self, args):
        out = self.f.exec(args[:-1])
        acc, g = 0, 1
        for o in out:
            acc = (acc + g * o) % P
            g = g * self.gamma % P
        return acc * args[-1] % P


def eq_table(point):
    """eq_poly_sequence_last (utils.rs:222-250) by its definition: eq(point, x) for x = 0 .. 2^n - 1, point[0] <-> the MSB of x"""
    n = len(point)
    out = []
    for x in range(1 << n):
        v = 1
        for j in range(n):
            bit = (x >> (n - 1 - j)) & 1
            v = v * (point[j] if bit else (1 - point[j])) % P
        out.append(v)
    return out
