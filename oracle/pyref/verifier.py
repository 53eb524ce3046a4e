"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Python big-int restatement of the gen-2 verifier, Pippenger::verify (src/cleanup/protocols/pippenger.rs:296-406), written
from the reference's *verify* functions (not from this package's provers), so that the order and sizes of everything the
provers put on the transcript are checked against what the reference verifier reads:
  GenericSumcheckProtocol::verify / main_cycle_sumcheck_verifier   cleanup/protocols/sumcheck.rs:63-77,125-127
  DenseDeg2Sumcheck::verify                                         cleanup/protocols/sumchecks/dense_eq.rs:223-237
  VecVecDeg2Sumcheck::verify                                        cleanup/protocols/sumchecks/vecvec_eq.rs:452-467
  DenseEqSumcheck::verify                                           cleanup/protocols/sumcheck.rs:874-889
  SplitAt / GlueSplit / ZeroCheck ::verify (= prove)                splits.rs:145-147,199-201, zero_check.rs:31-33
  SimpleGKR::verify                                                 gkrs/gkr.rs:52-58
  PippengerBucketed::verify                                         pippenger_ending.rs:151-163
  LogupMainphaseProtocol::verify                                    pushforward/logup_mainphase.rs:202-240
  PushforwardProtocol::verify                                       pushforward/pushforward.rs:849-968
  MultiOpenReduction::verify                                        multiopen_reduction.rs:95-117
  KnucklesOpeningProtocol::verify                                   opening.rs:100-143
The transcript is a reader over the prover's messages (scalars and G1 points in write order) plus the challenge tape.
A failed check raises VerifyError.
"""
from . import g1 as G
from . import knuckles as KN
from . import pushforward as PF
from .field import P
from .gkr import bintree_protocol_layers, triangle_protocol_layers, split_at_prove, glue_split_prove
from .polys import HI, eq_eval, eq_poly_sequence_last, zip_with_gamma, make_gamma_pows
from .sumcheck import decompress_coefficients, evaluate_univar


class VerifyError(Exception):
    pass


def _check(cond, what):
    if not cond:
        raise VerifyError(what)


class ReadTranscript:
    """verifier-mode ProofTranscript2 over recorded messages: scalars (flat list), points, tape challenges"""

    def __init__(self, scalars, points, tape):
        self.scalars, self.points, self.tape = list(scalars), list(points), list(tape)
        self.si = self.pi = self.pos = 0

    def read_scalars(self, n):
        _check(self.si + n <= len(self.scalars), "proof ran out of scalars")
        out = self.scalars[self.si:self.si + n]
        self.si += n
        return out

    def read_points(self, n):
        _check(self.pi + n <= len(self.points), "proof ran out of points")
        out = self.points[self.pi:self.pi + n]
        self.pi += n
        return out

    def challenge(self, bits=128):
        v = self.tape[self.pos]
        self.pos += 1
        return v % P if bits >= 255 else v & ((1 << bits) - 1)

    def done(self):
        return self.si == len(self.scalars) and self.pi == len(self.points)


def sumcheck_verify(tr, degrees, claim):
    """main_cycle_sumcheck_verifier (sumcheck.rs:63-77)"""
    r = []
    for d in degrees:
        msg = tr.read_scalars(d)
        poly = decompress_coefficients(msg, claim)
        x = tr.challenge(128)
        r.append(x)
        claim = evaluate_univar(poly, x)
    r.reverse()
    return claim, r


def layer_sumcheck_verify(tr, f, num_vars, claims):
    """DenseDeg2Sumcheck / VecVecDeg2Sumcheck / DenseEqSumcheck ::verify -- the same text in all three"""
    gamma = tr.challenge(128)
    point, evs = claims
    folded = zip_with_gamma(gamma, list(evs))
    ev, out_pt = sumcheck_verify(tr, [f.deg + 1] * num_vars, folded)
    poly_evs = tr.read_scalars(f.n_ins)
    lhs = zip_with_gamma(gamma, [v % P for v in f.exec(poly_evs)]) * eq_eval(list(point), out_pt) % P
    _check(lhs == ev % P, "Final combinator check has failed (%s, %d variables)" % (f.name, num_vars))
    return (out_pt, poly_evs)


def simple_gkr_verify(tr, layers, claims):
    """gkr.rs:52-58 : layers in reverse"""
    for layer in reversed(layers):
        kind = layer[0]
        if kind in ("vecvec", "dense"):
            claims = layer_sumcheck_verify(tr, layer[1], layer[2], claims)
        elif kind == "split":
            claims = split_at_prove(tr, claims, layer[1], layer[2])
        elif kind == "zerocheck":
            claims = (claims[0], list(claims[1]) + [0, 0])
        else:
            raise ValueError(kind)
    return claims


def pippenger_bucketed_verify(tr, multirow_vars, bucket_vars, horizontal_vars, claims):
    """pippenger_ending.rs:151-163"""
    claims = simple_gkr_verify(tr, triangle_protocol_layers(multirow_vars + bucket_vars - 2, HI(multirow_vars)), claims)
    claims = split_at_prove(tr, claims, HI(multirow_vars), 3)
    claims = split_at_prove(tr, claims, HI(multirow_vars), 3)
    return simple_gkr_verify(tr, bintree_protocol_layers(multirow_vars + bucket_vars + horizontal_vars, horizontal_vars,
                                                          horizontal_vars, True), claims)


def logup_mainphase_verify(tr, logsizes, claim):
    """logup_mainphase.rs:202-240"""
    f = PF.LogupLayerFn
    num, denom = tr.read_scalars(2)
    _check(denom % P != 0, "logup: zero denominator")
    _check(num % P == denom * claim % P, "logup: num != denom * claim")
    logsizes = list(logsizes)
    curr = 0
    running = ([], [num, denom])
    acc = []
    while True:
        incoming = logsizes[-1]
        c4 = layer_sumcheck_verify(tr, f, curr, running)
        if incoming == curr:
            if len(logsizes) == 2:
                tmp = c4
                break
            running = (list(c4[0]), [c4[1][0], c4[1][1]])
            acc.append((list(c4[0]), [c4[1][2], c4[1][3]]))
            logsizes.pop()
        else:
            running = split_at_prove(tr, c4, HI(0), 2)
            curr += 1
    acc.append(tmp)
    acc.reverse()
    return acc


def pushforward_verify(tr, x_logsize, y_logsize, y_size, d_logsize, claims):
    """pushforward.rs:849-968"""
    point, evs = list(claims[0]), list(claims[1])
    evs[1] = (evs[1] - 1) % P
    r_y = point[:y_logsize]
    _check(len(point) == y_logsize + d_logsize + x_logsize, "pushforward: claim point length")
    matrix_logsize = x_logsize + y_logsize
    matrix_size = (1 << x_logsize) * y_size
    psi, tau_c, tau_d, tau_s = [tr.challenge(512) for _ in range(4)]
    gamma = tr.challenge(128)
    suppression_total = 2 * ((1 << matrix_logsize) - matrix_size) * pow(tau_s, P - 2, P) % P
    mp = logup_mainphase_verify(tr, [matrix_logsize - 1, matrix_logsize - 1, x_logsize, d_logsize], suppression_total)
    _check(len(mp) == 3, "logup: three claim groups")
    cd_claims, ac_c_claims, ac_d_claims = mp
    cd_point, cd_evs = split_at_prove(tr, cd_claims, HI(0), 2)
    gammas = make_gamma_pows(gamma, 5)
    _check(len(evs) == 3 and len(cd_evs) == 2, "pushforward: claim counts")
    ev_folded = (evs[0] + gammas[1] * evs[1] + gammas[2] * evs[2]) % P
    claim = (cd_evs[0] + gammas[1] * cd_evs[1] + gammas[2] * ev_folded) % P
    out_pt = []
    for _ in range(matrix_logsize):
        msg = tr.read_scalars(3)
        poly = decompress_coefficients(msg, claim)
        t = tr.challenge(128)
        claim = evaluate_univar(poly, t)
        out_pt.append(t)
    out_pt.reverse()
    p_folded_ev, c_pull_ev, d_pull_ev, c_ev, d_ev = tr.read_scalars(5)
    adj_p = (p_folded_ev - gamma) % P
    p_sel = adj_p * PF.eq_trunc_evaluate(y_logsize, y_size, r_y, out_pt[:y_logsize]) % P
    sel_ev = PF.selector_evaluate(y_logsize, y_size, out_pt[:y_logsize])
    tmp = tau_s * (1 - sel_ev) % P
    c_adj = (c_pull_ev + psi * c_ev - tau_c * sel_ev + tmp) % P
    d_adj = (d_pull_ev + psi * d_ev - tau_d * sel_ev + tmp) % P
    lhs = (eq_eval(list(cd_point), out_pt) * ((c_adj + d_adj) + gammas[1] * c_adj * d_adj)
           + gammas[2] * (c_pull_ev * d_pull_ev * p_sel)) % P
    _check(lhs == claim % P, "pushforward: combined sumcheck final check")
    return dict(gamma=gamma, matrix=(out_pt, [p_folded_ev, c_pull_ev, d_pull_ev, c_ev, d_ev]), ac_c=ac_c_claims, ac_d=ac_d_claims)


def multiopen_verify(tr, nvars, claims):
    """multiopen_reduction.rs:95-117; claims = [(point, ev)]"""
    nargs = len(claims)
    gamma = tr.challenge(128)
    fun = PF.FoldedProdAlgFn(gamma, nargs)
    folded = zip_with_gamma(gamma, [c[1] for c in claims])
    claim, out_pt = sumcheck_verify(tr, [fun.deg] * nvars, folded)
    evs = tr.read_scalars(nargs)
    ext = list(evs) + [eq_eval(list(c[0]), out_pt) for c in claims]
    _check(claim % P == fun.exec(ext) % P, "multiopen: final combinator check")
    return out_pt, evs


def knuckles_verify(tr, g0, k, num_vars, commitment, point, ev):
    """opening.rs:100-143; returns the pairing pair (a, b): the proof is valid iff e(a, [1]_2) = e(b, [tau]_2)"""
    t_comm = tr.read_points(1)[0]
    x = tr.challenge(128)
    kx = x * k % P
    t_x, p_x = tr.read_scalars(2)
    lam = tr.challenge(128)
    p_lt_comm = G.add(G.mul(t_comm, lam), commitment)
    p_lt_open = (t_x * lam + p_x) % P
    p_lt_x_proof = tr.read_points(1)[0]
    a0, b0 = KN.verify_reduce_to_pair(g0, p_lt_comm, p_lt_x_proof, x, p_lt_open)
    t_kx = tr.read_scalars(1)[0]
    t_kx_proof = tr.read_points(1)[0]
    a1, b1 = KN.verify_reduce_to_pair(g0, t_comm, t_kx_proof, kx, t_kx)
    k_pow = pow(k, (1 << num_vars) - 1, P)
    xpow, eq_ev = x, 1
    for i in range(num_vars):
        r = point[num_vars - i - 1]
        eq_ev = eq_ev * (r + (1 - r) * xpow) % P
        xpow = xpow * xpow % P
    lhs = (x * (t_kx - k_pow * t_x) + xpow * ev) % P
    rhs = x * p_x * eq_ev % P
    _check(lhs == rhs, "knuckles: T(kx) - k^(N-1) T(x) identity")
    fin = tr.challenge(128)
    return G.add(a0, G.mul(a1, fin)), G.add(b0, G.mul(b1, fin))


def pippenger_verify(tr, claims, y_size, y_logsize, d_logsize, x_logsize, clm, g0, k):
    """pippenger.rs:296-406; claims = (r_y, evs of the dense output).  Returns the deferred pairing pair."""
    n_mat = -(-y_size // (1 << clm))
    c = tr.read_points(n_mat)
    d = tr.read_points(n_mat)
    p_0 = tr.read_points(1)[0]
    p_1 = tr.read_points(1)[0]
    ac_c = tr.read_points(1)[0]
    ac_d = tr.read_points(1)[0]
    claims = pippenger_bucketed_verify(tr, y_logsize, d_logsize, x_logsize, claims)
    claims = glue_split_prove(tr, claims)
    c_pull = tr.read_points(n_mat)
    d_pull = tr.read_points(n_mat)
    fin = pushforward_verify(tr, x_logsize, y_logsize, y_size, d_logsize, claims)
    gamma = fin["gamma"]
    matrix_pt, (p_folded_ev, c_pull_ev, d_pull_ev, c_ev, d_ev) = fin["matrix"]
    ac_c_pt, ac_c_evs = fin["ac_c"]
    ac_d_pt, ac_d_evs = fin["ac_d"]
    p_folded_point = [0] * clm + list(matrix_pt[y_logsize:])
    ac_c_point = [0] * clm + list(ac_c_pt)
    ac_d_point = [0] * (x_logsize + clm - d_logsize) + list(ac_d_pt)
    combined_point = list(matrix_pt[y_logsize - clm:])
    multirow = eq_poly_sequence_last(list(matrix_pt[:y_logsize - clm]))

    def comb(cs):
        acc = None
        for coeff, cmt in zip(multirow, cs):
            acc = G.add(acc, G.mul(cmt, coeff))
        return acc
    c_comb, d_comb, cp_comb, dp_comb = comb(c), comb(d), comb(c_pull), comb(d_pull)
    u = tr.challenge(512)
    us = make_gamma_pows(u, 4)
    combined_comm = G.add(G.add(c_comb, G.mul(d_comb, us[1])), G.add(G.mul(cp_comb, us[2]), G.mul(dp_comb, us[3])))
    combined_ev = (c_ev + d_ev * us[1] + c_pull_ev * us[2] + d_pull_ev * us[3]) % P
    nv = x_logsize + clm
    mo_pt, mo_evs = multiopen_verify(tr, nv, [(p_folded_point, (p_folded_ev - gamma * gamma) % P), (ac_c_point, ac_c_evs[0]),
                                              (ac_d_point, ac_d_evs[0]), (combined_point, combined_ev)])
    q = tr.challenge(128)
    qs = make_gamma_pows(q, 4)
    parts = [G.add(p_0, G.mul(p_1, gamma)), ac_c, ac_d, combined_comm]
    folded_comm = None
    for a, b in zip(qs, parts):
        folded_comm = G.add(folded_comm, G.mul(b, a))
    pair = knuckles_verify(tr, g0, k, nv, folded_comm, mo_pt, zip_with_gamma(q, list(mo_evs)))
    _check(tr.done(), "proof has unread messages")
    return pair
