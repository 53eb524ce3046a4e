"""TEST INFRASTRUCTURE ONLY (oracle).

gen-2 sumcheck objects and round driver.  Restates
  /root/reference/src/cleanup/protocols/sumcheck.rs:14-44, 101-123, 237-347, 591-602, 706-889
  /root/reference/src/cleanup/protocols/sumchecks/dense_eq.rs:20-248
  /root/reference/src/cleanup/protocols/sumchecks/vecvec_eq.rs:27-475
liblasso `UniPoly::from_evals` (git 925a7a74, un-vendored) = the unique
interpolation through x = 0..D; restated here as exact Lagrange interpolation.
"""
from .field import P, inv
from .polys import (EQPolyData, bind_dense, dense_bind_21, dense_make_21, eq_eval,
                    eq_poly_sequence, eq_poly_sequence_from_multiplier, eq_poly_sequence_last,
                    make_gamma_pows, zip_with_gamma, gamma_rlc)


# ------------------------------------------------------------------ univariate helpers
def poly_mul_linear(c, a):
    """c(x) * (x - a)"""
    out = [0] * (len(c) + 1)
    for i, v in enumerate(c):
        out[i + 1] = (out[i + 1] + v) % P
        out[i] = (out[i] - a * v) % P
    return out


def unipoly_from_evals(evals):
    """coefficients (low -> high) of the unique deg<=D poly with p(i)=evals[i], i=0..D"""
    n = len(evals)
    coeffs = [0] * n
    for i in range(n):
        num = [1]
        den = 1
        for j in range(n):
            if j != i:
                num = poly_mul_linear(num, j)
                den = den * (i - j) % P
        s = evals[i] * inv(den) % P
        for k in range(n):
            coeffs[k] = (coeffs[k] + num[k] * s) % P
    return coeffs


def evaluate_univar(coeffs, x):
    """sumcheck.rs:33-44"""
    r = 0
    for c in reversed(coeffs):
        r = (r * x + c) % P
    return r


def compress_coefficients(coeffs):
    """sumcheck.rs:27-31"""
    return [coeffs[0]] + list(coeffs[2:])


def decompress_coefficients(cwl, s):
    """sumcheck.rs:14-25"""
    sm = 2 * cwl[0] % P
    for c in cwl[1:]:
        sm = (sm + c) % P
    return [cwl[0], (s - sm) % P] + list(cwl[1:])


def from12(p1, p2, eq1, prev_claim):
    """vecvec_eq.rs:197-216 ; returns coefficient vector"""
    eq0 = (1 - eq1) % P
    eq2 = (2 * eq1 - eq0) % P
    eq3 = (2 * eq2 - eq1) % P
    prod1 = p1 * eq1 % P
    prod0 = (prev_claim - prod1) % P
    p0 = prod0 * inv(eq0) % P
    p3 = (3 * p2 - 3 * p1 + p0) % P
    return unipoly_from_evals([prod0, prod1, p2 * eq2 % P, p3 * eq3 % P])


# ------------------------------------------------------------------ transcript stand-in
class TapeTranscript:
    """Deterministic stand-in for ProofTranscript2 (merlin is out of scope):
    challenges come from a fixed tape, prover messages are recorded."""

    def __init__(self, tape):
        self.tape = list(tape)
        self.pos = 0
        self.msgs = []

    def challenge(self, bits=128):
        v = self.tape[self.pos]
        self.pos += 1
        return v % P if bits >= 255 else v & ((1 << bits) - 1)

    def write_scalars(self, xs):
        self.msgs.append(list(xs))


# ------------------------------------------------------------------ generic wrappers
class GammaWrapper:
    """sumcheck.rs:706-740"""

    def __init__(self, f, gamma):
        assert f.n_outs > 1
        self.f = f
        self.gamma_pows = [gamma % P]
        for _ in range(f.n_outs - 2):
            self.gamma_pows.append(gamma * self.gamma_pows[-1] % P)
        self.deg, self.n_ins = f.deg, f.n_ins

    def exec(self, args):
        out = self.f.exec(args)
        ret = out[0]
        for a, b in zip(out[1:], self.gamma_pows):
            ret = (ret + a * b) % P
        return ret


class EqWrapper:
    """sumcheck.rs:803-829"""

    def __init__(self, f):
        self.f = f
        self.deg, self.n_ins = f.deg + 1, f.n_ins + 1

    def exec(self, args):
        return self.f.exec(args) * args[self.f.n_ins] % P


class Prod3Fn:
    deg, n_ins = 3, 3

    def exec(self, a):
        return a[0] * a[1] % P * a[2] % P


class DenseSumcheckObjectSO:
    """sumcheck.rs:237-347 ; generic degree-D dense round object"""

    def __init__(self, polys, f, num_vars, claim_hint):
        assert len(polys) == f.n_ins
        for p in polys:
            assert len(p) == 1 << num_vars
        self.polys = [list(p) for p in polys]
        self.f, self.num_vars, self.round_idx = f, num_vars, 0
        self.claim = claim_hint % P
        self.cached, self._challenges = None, []

    def unipoly(self):
        if self.cached is None:
            half = 1 << (self.num_vars - self.round_idx - 1)
            n = len(self.polys)
            D = self.f.deg
            acc = [0] * D
            for i in range(half):
                args = [self.polys[j][2 * i + 1] for j in range(n)]
                acc[0] = (acc[0] + self.f.exec(args)) % P
                difs = [(self.polys[j][2 * i + 1] - self.polys[j][2 * i]) % P for j in range(n)]
                for s in range(1, D):
                    args = [(a + d) % P for a, d in zip(args, difs)]
                    acc[s] = (acc[s] + self.f.exec(args)) % P
            total = [0] + acc
            total[0] = (self.claim - total[1]) % P
            self.cached = unipoly_from_evals(total)
        return self.cached

    def bind(self, t):
        assert self.cached is not None
        self._challenges.append(t)
        self.polys = [bind_dense(p, t) for p in self.polys]
        self.round_idx += 1
        self.claim = evaluate_univar(self.cached, t)
        self.cached = None

    def final_evals(self):
        assert self.round_idx == self.num_vars
        return [p[0] for p in self.polys]


def dense_eq_sumcheck_object(polys, f, point, claim_hint, gamma):
    """sumcheck.rs:378-417 : DenseEqSumcheckObject::rlc"""
    fw = EqWrapper(GammaWrapper(f, gamma))
    polys = [list(p) for p in polys] + [eq_poly_sequence_last(point)]
    return DenseSumcheckObjectSO(polys, fw, len(point), gamma_rlc(gamma, claim_hint))


class DenseDeg2SumcheckObjectSO:
    """dense_eq.rs:61-173"""

    def __init__(self, polys, func, gamma_pows, claim, point):
        self.eq_poly_data = eq_poly_sequence(point[0:len(point) - 1])
        self.polys = [list(p) for p in polys]
        self.func, self.gamma_pows, self.claim = func, gamma_pows, claim % P
        self.point = list(point)
        self.multiplier, self.cached, self.current_point = 1, None, []

    @staticmethod
    def rlc(polys, func, claims, point, gamma):
        """dense_eq.rs:42-59"""
        gp = make_gamma_pows(gamma, func.n_outs)
        claim = claims[0]
        for i in range(1, len(claims)):
            claim = (claim + gp[i] * claims[i]) % P
        return DenseDeg2SumcheckObjectSO(polys, func, gp, claim, point)

    def unipoly(self):
        assert self.cached is None
        for v in self.polys:
            dense_make_21(v)
        n_out = self.func.n_outs
        pad = self.func.exec([0] * len(self.polys))
        s2, s1, eq_sum_ = [0] * n_out, [0] * n_out, 0
        eq = self.eq_poly_data[-1]
        for idx in range(len(self.polys[0]) // 2):
            a2 = self.func.exec([p[2 * idx] for p in self.polys])
            a1 = self.func.exec([p[2 * idx + 1] for p in self.polys])
            for i in range(n_out):
                s2[i] = (s2[i] + a2[i] * eq[idx]) % P
                s1[i] = (s1[i] + a1[i] * eq[idx]) % P
            eq_sum_ = (eq_sum_ + eq[idx]) % P
        tr = (1 - eq_sum_) % P
        for i in range(n_out):
            s2[i] = (s2[i] + pad[i] * tr) % P
            s1[i] = (s1[i] + pad[i] * tr) % P
        t2, t1 = s2[0], s1[0]
        for i in range(1, n_out):
            t2 = (t2 + s2[i] * self.gamma_pows[i]) % P
            t1 = (t1 + s1[i] * self.gamma_pows[i]) % P
        t2 = t2 * self.multiplier % P
        t1 = t1 * self.multiplier % P
        self.cached = from12(t1, t2, self.point[-1], self.claim)
        return self.cached

    def bind(self, t):
        q = self.point[-1]
        self.multiplier = self.multiplier * ((1 - q - t + 2 * q * t) % P) % P
        self.polys = [dense_bind_21(v, t) for v in self.polys]
        self.current_point.append(t)
        self.eq_poly_data.pop()
        self.point.pop()
        self.claim = evaluate_univar(self.cached, t)
        self.cached = None

    def final_evals(self):
        return [p[0] for p in self.polys]


class VecVecDeg2SumcheckObjectSO:
    """vecvec_eq.rs:72-398 (Sparse stage + handover to the dense stage)"""

    def __init__(self, polys, func, gamma_pows, claim, point, col_logsize):
        self.polys = [p.clone() for p in polys]
        self.func, self.gamma_pows, self._claim = func, gamma_pows, claim % P
        self.eq = EQPolyData(point, col_logsize, max(len(r) for r in polys[0].data))
        self.cached, self.dense, self.current_point = None, None, []

    @staticmethod
    def rlc(polys, func, claims, point, num_vertical_vars, gamma):
        """vecvec_eq.rs:53-70"""
        gp = make_gamma_pows(gamma, func.n_outs)
        claim = claims[0]
        for i in range(1, len(claims)):
            claim = (claim + gp[i] * claims[i]) % P
        return VecVecDeg2SumcheckObjectSO(polys, func, gp, claim, point, num_vertical_vars)

    def claim(self):
        return self.dense.claim if self.dense is not None else self._claim

    def unipoly(self):
        if self.dense is not None:
            return self.dense.unipoly()
        assert self.cached is None
        for p in self.polys:
            p.make_21()
        n_out = self.func.n_outs
        pad = self.func.exec([p.row_pad for p in self.polys])
        cpad = self.func.exec([p.col_pad for p in self.polys])
        s2, s1 = [0] * n_out, [0] * n_out
        rows = len(self.polys[0].data)
        for ri in range(rows):
            l2, l1 = [0] * n_out, [0] * n_out
            seg = len(self.polys[0].data[ri]) // 2
            eq = self.eq.get_segment_evals(seg)
            for idx in range(seg):
                a2 = self.func.exec([p.data[ri][2 * idx] for p in self.polys])
                a1 = self.func.exec([p.data[ri][2 * idx + 1] for p in self.polys])
                for i in range(n_out):
                    l2[i] = (l2[i] + a2[i] * eq[idx]) % P
                    l1[i] = (l1[i] + a1[i] * eq[idx]) % P
            tr = self.eq.get_trailing_sum(seg)
            vm = self.eq.row_eq_coefs[ri]
            for i in range(n_out):
                s2[i] = (s2[i] + (l2[i] + pad[i] * tr) * vm) % P
                s1[i] = (s1[i] + (l1[i] + pad[i] * tr) * vm) % P
        if rows < (1 << self.eq.padded_vars_idx):
            for i in range(n_out):
                res = cpad[i] * self.eq.row_eq_coefs_tail_sums[rows] % P
                s2[i] = (s2[i] + res) % P
                s1[i] = (s1[i] + res) % P
        t2, t1 = s2[0], s1[0]
        for i in range(1, n_out):
            t2 = (t2 + s2[i] * self.gamma_pows[i]) % P
            t1 = (t1 + s1[i] * self.gamma_pows[i]) % P
        t2 = t2 * self.eq.multiplier % P
        t1 = t1 * self.eq.multiplier % P
        self.cached = from12(t1, t2, self.eq.point[self.eq.binding_var_idx], self._claim)
        return self.cached

    def bind(self, t):
        if self.dense is not None:
            self.dense.bind(t)
            return
        if self.eq.binding_var_idx > self.eq.padded_vars_idx:
            for p in self.polys:
                p.bind_21(t)
            self.current_point.append(t)
            self.eq.bind(t)
            self._claim = evaluate_univar(self.cached, t)
            self.cached = None
        else:
            self._bind_into_dense(t)

    def _bind_into_dense(self, t):
        """vecvec_eq.rs:157-190"""
        tm1 = (t - 1) % P
        n = 1 << self.eq.padded_vars_idx
        polys = []
        for p in self.polys:
            col = []
            for r in p.data:
                if len(r) == 0:
                    col.append(p.row_pad)
                else:
                    assert len(r) == 2
                    col.append((r[1] + tm1 * (r[0] - r[1])) % P)
            col += [p.col_pad] * (n - len(col))
            polys.append(col[:n])
        q = self.eq.point[self.eq.binding_var_idx]
        mult = self.eq.multiplier * ((1 - q - t + 2 * q * t) % P) % P
        polys.append(eq_poly_sequence_from_multiplier(mult, self.eq.point[0:self.eq.padded_vars_idx])[-1])
        fw = EqWrapper(GammaWrapper(self.func, self.gamma_pows[1]))
        self.dense = DenseSumcheckObjectSO(polys, fw, self.eq.padded_vars_idx, evaluate_univar(self.cached, t))
        self.cached = None

    def final_evals(self):
        assert self.dense is not None
        return self.dense.final_evals()


# ------------------------------------------------------------------ drivers
def generic_sumcheck_prove(transcript, degrees, claim, so, record=None):
    """sumcheck.rs:101-123 ; returns ((claim, r), final_evals)"""
    r = []
    for d in degrees:
        poly = so.unipoly()
        msg = compress_coefficients(poly)
        assert len(msg) == d, (len(msg), d)
        transcript.write_scalars(msg)
        if record is not None:
            record.append(list(poly))
        x = transcript.challenge(128)
        r.append(x)
        so.bind(x)
        claim = evaluate_univar(poly, x)
    r.reverse()
    return (claim, r), so.final_evals()


def dense_deg2_sumcheck_prove(transcript, f, num_vars, claims, advice, record=None):
    """dense_eq.rs:198-229 ; claims = (point, evs)"""
    assert f.deg == 2
    gamma = transcript.challenge(128)
    point, evs = claims
    so = DenseDeg2SumcheckObjectSO.rlc(advice, f, evs, point, gamma)
    (_, pt), poly_evs = generic_sumcheck_prove(transcript, [f.deg + 1] * num_vars, so.claim, so, record)
    transcript.write_scalars(poly_evs)
    return (pt, poly_evs)


def vecvec_deg2_sumcheck_prove(transcript, f, num_vars, num_vertical_vars, claims, advice, record=None):
    """vecvec_eq.rs:424-456"""
    assert f.deg == 2
    gamma = transcript.challenge(128)
    point, evs = claims
    so = VecVecDeg2SumcheckObjectSO.rlc(advice, f, evs, point, num_vertical_vars, gamma)
    (_, pt), poly_evs = generic_sumcheck_prove(transcript, [f.deg + 1] * num_vars, so.claim(), so, record)
    poly_evs = poly_evs[:-1]
    transcript.write_scalars(poly_evs)
    return (pt, poly_evs)


def dense_eq_sumcheck_prove(transcript, f, num_vars, claims, advice, record=None):
    """sumcheck.rs:844-872 : DenseEqSumcheck::prove"""
    gamma = transcript.challenge(128)
    point, evs = claims
    so = dense_eq_sumcheck_object(advice, f, point, evs, gamma)
    (_, pt), poly_evs = generic_sumcheck_prove(transcript, [f.deg + 1] * num_vars, so.claim, so, record)
    poly_evs = poly_evs[:-1]
    transcript.write_scalars(poly_evs)
    return (pt, poly_evs)


def verify_dense_like(transcript_msgs, tape, f, num_vars, claims, kind):
    """Replays the verifier of DenseDeg2Sumcheck / VecVecDeg2Sumcheck
    (dense_eq.rs:231-247, vecvec_eq.rs:458-475) against recorded prover messages."""
    t = TapeTranscript(tape)
    gamma = t.challenge(128)
    point, evs = claims
    claim = zip_with_gamma(gamma, evs)
    r = []
    msgs = list(transcript_msgs)
    for _ in range(num_vars):
        poly = decompress_coefficients(msgs.pop(0), claim)
        x = t.challenge(128)
        r.append(x)
        claim = evaluate_univar(poly, x)
    r.reverse()
    poly_evs = msgs.pop(0)
    assert zip_with_gamma(gamma, f.exec(poly_evs)) * eq_eval(point, r) % P == claim, "final combinator check"
    return (r, poly_evs)
