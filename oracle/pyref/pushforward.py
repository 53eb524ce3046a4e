"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Python big-int restatement of the pushforward argument of the gen-2 prover ("prove pushforward", pippenger.rs:147-160):

  * PushForwardState::new, Fr columns (c, d, ac_c, ac_d) and second_phase (c_pull, d_pull)   pushforward/pushforward.rs:489-510, 572-596
  * PushforwardProtocol::prove                                                                pushforward/pushforward.rs:640-846
  * LogupLayerFn, LogupMainphaseProtocol::{make_witness, prove}                               pushforward/logup_mainphase.rs:30-208
  * AddInversesFn, Prod3Fn                                                                    pushforward/pushforward.rs:27-49, 255-281
  * EqTruncPoly, SelectorPoly                                                                 verifier_polys.rs:43-149
  * AlgFnUtils::map_split_hi, pad_vector                                                      utils/algfn.rs:82-89, utils.rs:324-329
"""
from .algfn import AlgFn
from .field import P
from .polys import HI, eq_poly_sequence_last, eq_sum, evaluate_poly
from .sumcheck import (DenseSumcheckObjectSO, Prod3Fn, compress_coefficients, dense_eq_sumcheck_object,
                       dense_eq_sumcheck_prove, evaluate_univar)
from .gkr import split_at_prove

AddInversesFn = AlgFn("add_inverses", 2, 2, 2, lambda a: [(a[0] + a[1]) % P, a[0] * a[1] % P])
LogupLayerFn = AlgFn("logup_layer", 2, 4, 2, lambda a: [(a[0] * a[3] + a[1] * a[2]) % P, a[1] * a[3] % P])


def make_gamma_pows(gamma, count):
    g = [1, gamma % P]
    for i in range(2, count):
        g.append(g[i - 1] * gamma % P)
    return g


def fmap(f, args):
    n = len(args[0])
    outs = [[0] * n for _ in range(f.n_outs)]
    for i in range(n):
        r = f.exec([a[i] for a in args])
        for o in range(f.n_outs):
            outs[o][i] = r[o]
    return outs


def map_split_hi(f, args):
    """utils/algfn.rs:82-89: the two contiguous halves are mapped separately"""
    half = len(args[0]) // 2
    return [fmap(f, [a[:half] for a in args]), fmap(f, [a[half:] for a in args])]


def pad_vector(v, logsize, w):
    assert len(v) <= 1 << logsize
    return list(v) + [w % P] * ((1 << logsize) - len(v))


# ------------------------------------------------------------------ verifier polys
def eq_trunc_evals(num_vars, k, r):
    e = eq_poly_sequence_last(r)
    return [e[i] if i < k else 0 for i in range(1 << num_vars)]


def eq_trunc_evaluate(num_vars, k, r, pt):
    """verifier_polys.rs:108-147"""
    partial = [1]
    for i in range(num_vars):
        j = num_vars - i - 1
        partial.append(partial[-1] * ((1 - pt[j] - r[j] + 2 * r[j] * pt[j]) % P) % P)
    mult, acc = 1, 0
    if k >= (1 << num_vars):
        assert k == 1 << num_vars
        return partial[num_vars]
    for i in range(num_vars):
        left_bit = k >> (num_vars - i - 1)
        prev = mult
        if left_bit == 1:
            mult = mult * pt[i] % P * r[i] % P
            acc = (acc + prev * (1 - pt[i]) % P * (1 - r[i]) % P * partial[num_vars - i - 1]) % P
        else:
            mult = mult * (1 - pt[i]) % P * (1 - r[i]) % P
        k -= left_bit << (num_vars - i - 1)
    return acc


def selector_evaluate(num_vars, k, pt):
    return eq_sum(pt, k)


# ------------------------------------------------------------------ phase data
def phase1_data(points, digits, counter, x_logsize, d_logsize):
    """pushforward.rs:489-510: c, d flattened [y][x]; negated access counts"""
    d = [int(v) % P for row in digits for v in row]
    c = [int(v) % P for row in counter for v in row]
    ac_d = [0] * (1 << d_logsize)
    ac_c = [0] * (1 << x_logsize)
    for row in digits:
        for v in row:
            ac_d[int(v)] += 1
    for row in counter:
        for v in row:
            ac_c[int(v)] += 1
    return dict(c=c, d=d, p_0=[p[0] for p in points], p_1=[p[1] for p in points], ac_c=[(-v) % P for v in ac_c],
                ac_d=[(-v) % P for v in ac_d])


def phase2_data(digits, counter, r, y_logsize, d_logsize, x_logsize):
    """second_phase (pushforward.rs:572-596); r = [r_y | r_d | r_c]"""
    r_d = r[y_logsize:y_logsize + d_logsize]
    r_c = r[y_logsize + d_logsize:]
    assert len(r_c) == x_logsize
    eq_c, eq_d = eq_poly_sequence_last(r_c), eq_poly_sequence_last(r_d)
    return dict(c_pull=[eq_c[int(v)] for row in counter for v in row], d_pull=[eq_d[int(v)] for row in digits for v in row])


# ------------------------------------------------------------------ logup main phase
def logup_make_witness(logsizes, inputs):
    """logup_mainphase.rs:83-143"""
    for (n, d), lg in zip(inputs, logsizes):
        assert len(n) == 1 << lg and len(d) == 1 << lg
    inputs = list(reversed([[list(a), list(b)] for a, b in inputs]))
    layers = [inputs.pop(), inputs.pop()]
    i = 0
    while True:
        next_size = len(inputs[-1][0]) if inputs else 1
        curr = len(layers[i][0])
        a0, a1 = layers[i], layers[i + 1]
        if curr == next_size:
            layers.append(fmap(LogupLayerFn, [a0[0], a0[1], a1[0], a1[1]]))
            if inputs:
                layers.append(inputs.pop())
            else:
                break
            i += 2
        else:
            assert curr > next_size
            o0, o1 = map_split_hi(LogupLayerFn, [a0[0], a0[1], a1[0], a1[1]])
            layers.append(o0)
            layers.append(o1)
            i += 2
    tmp = layers.pop()
    assert len(tmp[0]) == 1 and len(tmp[1]) == 1
    return layers, (tmp[0][0], tmp[1][0])


def logup_mainphase_prove(tr, logsizes, claim, advice):
    """logup_mainphase.rs:156-208; returns the accumulated claims [(point, evs), ...]"""
    witness, (num, den) = logup_make_witness(logsizes, advice)
    assert den != 0 and num == den * claim % P
    tr.write_scalars([num, den])
    logsizes = list(logsizes)
    curr = 0
    running = ([], [num, den])
    acc = []
    while True:
        incoming = logsizes[-1]
        r0, r1 = witness.pop()
        l0, l1 = witness.pop()
        claim4 = dense_eq_sumcheck_prove(tr, LogupLayerFn, curr, running, [l0, l1, r0, r1])
        if incoming == curr:
            if len(logsizes) == 2:
                last = claim4
                break
            running = (list(claim4[0]), [claim4[1][0], claim4[1][1]])
            acc.append((list(claim4[0]), [claim4[1][2], claim4[1][3]]))
            logsizes.pop()
        else:
            running = split_at_prove(tr, claim4, HI(0), 2)
            curr += 1
    acc.append(last)
    acc.reverse()
    return acc


# ------------------------------------------------------------------ PushforwardProtocol::prove
def pushforward_prove(tr, x_logsize, y_logsize, y_size, d_logsize, claims, p1, p2):
    """pushforward.rs:640-846.  claims = (point [r_y | r_d | r_c], [ev_x, ev_y, ev_z]) = the image-part final claims.
    Returns dict(gamma, matrix=(point, evs), ac_c=(point, evs), ac_d=(point, evs))."""
    point, evs = list(claims[0]), list(claims[1])
    evs[1] = (evs[1] - 1) % P
    r_y = point[:y_logsize]
    r_d = point[y_logsize:y_logsize + d_logsize]
    r_c = point[y_logsize + d_logsize:]
    assert len(r_c) == x_logsize
    c, d, p_0, p_1, ac_c, ac_d = p1["c"], p1["d"], p1["p_0"], p1["p_1"], p1["ac_c"], p1["ac_d"]
    c_pull, d_pull = p2["c_pull"], p2["d_pull"]
    adj_p_1 = [(v - 1) % P for v in p_1]
    x_size = 1 << x_logsize
    mlog = x_logsize + y_logsize
    msize = x_size * y_size
    assert len(c) == msize and len(c_pull) == msize

    psi, tau_c, tau_d, tau_s = [tr.challenge(512) for _ in range(4)]
    gamma = tr.challenge(128)
    c_adj = pad_vector([(cp + psi * cv - tau_c) % P for cp, cv in zip(c_pull, c)], mlog, tau_s)
    d_adj = pad_vector([(dp + psi * dv - tau_d) % P for dp, dv in zip(d_pull, d)], mlog, tau_s)
    c_pull = pad_vector(c_pull, mlog, 0)
    d_pull = pad_vector(d_pull, mlog, 0)

    left, right = map_split_hi(AddInversesFn, [c_adj, d_adj])
    eq_c, eq_d = eq_poly_sequence_last(r_c), eq_poly_sequence_last(r_d)
    table_c = [(eq_c[i] + psi * i - tau_c) % P for i in range(x_size)]
    table_d = [(eq_d[i] + psi * i - tau_d) % P for i in range(1 << d_logsize)]
    supp_total = 2 * ((1 << mlog) - msize) * pow(tau_s, -1, P) % P

    main = logup_mainphase_prove(tr, [mlog - 1, mlog - 1, x_logsize, d_logsize], supp_total,
                                 [left, right, [ac_c, table_c], [ac_d, table_d]])
    assert len(main) == 3
    cd_claims, ac_c_claims, ac_d_claims = main
    cd_claims = split_at_prove(tr, cd_claims, HI(0), 2)

    g = make_gamma_pows(gamma, 5)
    p_folded = [(a + g[1] * b + g[2]) % P for a, b in zip(p_0, adj_p_1)]
    eq_sel_y = eq_trunc_evals(y_logsize, y_size, r_y)
    p_sel = [eq_sel_y[i >> x_logsize] * p_folded[i & (x_size - 1)] % P for i in range(1 << mlog)]
    ev_folded = (evs[0] + g[1] * evs[1] + g[2] * evs[2]) % P

    prod3 = DenseSumcheckObjectSO([p_sel, c_pull, d_pull], Prod3Fn(), mlog, ev_folded)
    cd_point, cd_evs = cd_claims
    assert len(cd_evs) == 2
    claim = ((cd_evs[0] + g[1] * cd_evs[1]) + g[2] * ev_folded) % P
    frac = dense_eq_sumcheck_object([c_adj, d_adj], AddInversesFn, cd_point, cd_evs, gamma)
    out_pt = []
    for _ in range(mlog):
        pr = prod3.unipoly()
        fr = frac.unipoly()
        assert len(pr) == 4 and len(fr) == 4
        comb = [(fr[k] + g[2] * pr[k]) % P for k in range(4)]
        assert (2 * comb[0] + comb[1] + comb[2] + comb[3]) % P == claim
        tr.write_scalars(compress_coefficients(comb))
        t = tr.challenge(128)
        claim = evaluate_univar(comb, t)
        out_pt.append(t)
        prod3.bind(t)
        frac.bind(t)
    out_pt.reverse()
    p_sel_ev, c_pull_ev, d_pull_ev = prod3.final_evals()
    c_adj_ev, d_adj_ev, _ = frac.final_evals()
    adj_p_folded_ev = p_sel_ev * pow(eq_trunc_evaluate(y_logsize, y_size, r_y, out_pt[:y_logsize]), -1, P) % P
    p_folded_ev = (adj_p_folded_ev + gamma) % P
    assert evaluate_poly(p_folded, out_pt[y_logsize:]) == adj_p_folded_ev
    sel_ev = selector_evaluate(y_logsize, y_size, out_pt[:y_logsize])
    tmp = tau_s * (1 - sel_ev) % P
    psi_inv = pow(psi, -1, P)
    c_ev = psi_inv * (c_adj_ev - c_pull_ev + tau_c * sel_ev - tmp) % P
    d_ev = psi_inv * (d_adj_ev - d_pull_ev + tau_d * sel_ev - tmp) % P
    out_evs = [p_folded_ev, c_pull_ev, d_pull_ev, c_ev, d_ev]
    tr.write_scalars(out_evs)
    return dict(gamma=gamma, matrix=(out_pt, out_evs), ac_c=ac_c_claims, ac_d=ac_d_claims)


# ------------------------------------------------------------------ MultiOpenReduction (multiopen_reduction.rs)
class FoldedProdAlgFn:
    """multiopen_reduction.rs:13-42"""
    deg = 2

    def __init__(self, gamma, nargs):
        self.gammas, self.nargs, self.n_ins = make_gamma_pows(gamma, max(nargs, 2))[:nargs], nargs, 2 * nargs

    def exec(self, a):
        return sum(a[i] * a[i + self.nargs] % P * self.gammas[i] for i in range(self.nargs)) % P


def multiopen_prove(tr, nvars, claims, advice):
    """multiopen_reduction.rs:65-93; claims = [(point, ev), ...]; returns (point, evs)"""
    from .polys import zip_with_gamma
    from .sumcheck import generic_sumcheck_prove
    nargs = len(claims)
    gamma = tr.challenge(128)
    fun = FoldedProdAlgFn(gamma, nargs)
    folded = zip_with_gamma(gamma, [ev for _, ev in claims])
    polys = [list(a) for a in advice] + [eq_poly_sequence_last(pt) for pt, _ in claims]
    so = DenseSumcheckObjectSO(polys, fun, nvars, folded)
    (_, pt), poly_evs = generic_sumcheck_prove(tr, [2] * nvars, so.claim, so)
    evs = poly_evs[:nargs]
    tr.write_scalars(evs)
    return (pt, evs)
