"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Python big-int restatement of the whole gen-2 prover, Pippenger::prove (src/cleanup/protocols/pippenger.rs:118-290), on top
of the pieces restated elsewhere in this package: PippengerWG::new (pippenger.rs:37-70: bucketing, the phase-1 G1 commitments
of PushForwardState::new, the bintree / triangle witness), the image part, second_phase and its commitments, the
pushforward argument, MultiOpenReduction and the Knuckles opening.  Transcript: scalars and G1 points are recorded in write
order; challenges come from a tape (merlin itself is SURVEY 8f-3).
"""
from . import g1 as G
from . import gkr as GK
from . import knuckles as KN
from . import pushforward as PF
from .field import P
from .polys import eq_poly_sequence_last, zip_with_gamma


class Transcript:
    """scalars + points in write order, tape challenges"""

    def __init__(self, tape):
        self.tape, self.pos, self.msgs, self.points, self.log, self.wide = list(tape), 0, [], [], [], set()

    def challenge(self, bits=128):
        v = self.tape[self.pos]
        if bits >= 255:
            self.wide.add(self.pos)   # which draws were challenge(512): the device tape holds them reduced mod p
        self.pos += 1
        return v % P if bits >= 255 else v & ((1 << bits) - 1)

    def write_scalars(self, xs):
        self.msgs.append(list(xs))
        self.log.append(("s", len(xs)))

    def write_points(self, ps):
        self.points.extend(ps)
        self.log.append(("p", len(ps)))


def pippenger_wg(points, coefs, y_size, y_logsize, d_logsize, x_logsize, clm, basis):
    """PippengerWG::new (pippenger.rs:37-70)"""
    image, digits, counter, wg = GK.pippenger_witness(points, coefs, y_size, y_logsize, d_logsize, x_logsize)
    p1 = PF.phase1_data(points, digits, counter, x_logsize, d_logsize)
    d_outer, c_outer, d_comm, c_comm = G.pushforward_outer(digits, counter, basis, x_logsize, d_logsize, clm)
    comm = dict(c=c_comm, d=d_comm, p_0=G.kzg_commit(basis, p1["p_0"]), p_1=G.kzg_commit(basis, p1["p_1"]),
                ac_c=G.kzg_commit(basis, p1["ac_c"]), ac_d=G.kzg_commit(basis, p1["ac_d"]))
    return dict(image=image, digits=digits, counter=counter, wg=wg, p1=p1, d_outer=d_outer, c_outer=c_outer, comm=comm)


def pippenger_prove(tr, st, claims, y_size, y_logsize, d_logsize, x_logsize, clm, basis, kn_inverses, k):
    """pippenger.rs:118-290; claims = (r_y, evs of the dense output).  Returns the deferred pairing pair."""
    cm = 1 << clm
    n_mat = -(-y_size // cm)
    c1 = st["comm"]
    assert len(c1["c"]) == n_mat and len(c1["d"]) == n_mat
    tr.write_points(c1["c"]); tr.write_points(c1["d"]); tr.write_points([c1["p_0"]]); tr.write_points([c1["p_1"]])
    tr.write_points([c1["ac_c"]]); tr.write_points([c1["ac_d"]])   # pippenger.rs:131-136
    claims = GK.prove_image_part(tr, y_logsize, d_logsize, x_logsize, claims, st["wg"])
    # commit phase 2
    p2 = PF.phase2_data(st["digits"], st["counter"], claims[0], y_logsize, d_logsize, x_logsize)
    r_d = claims[0][y_logsize:y_logsize + d_logsize]
    r_c = claims[0][y_logsize + d_logsize:]
    d_pull_c, c_pull_c = G.second_phase_comms(st["d_outer"], st["c_outer"], eq_poly_sequence_last(r_d), eq_poly_sequence_last(r_c))
    tr.write_points(c_pull_c); tr.write_points(d_pull_c)
    fin = PF.pushforward_prove(tr, x_logsize, y_logsize, y_size, d_logsize, claims, st["p1"], p2)
    # open
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
    c_comb, d_comb, cp_comb, dp_comb = comb(c1["c"]), comb(c1["d"]), comb(c_pull_c), comb(d_pull_c)
    u = tr.challenge(512)
    us = PF.make_gamma_pows(u, 4)
    combined_comm = G.add(G.add(c_comb, G.mul(d_comb, us[1])), G.add(G.mul(cp_comb, us[2]), G.mul(dp_comb, us[3])))
    combined_ev = (c_ev + d_ev * us[1] + c_pull_ev * us[2] + d_pull_ev * us[3]) % P
    x_size = 1 << x_logsize
    p1 = st["p1"]
    combined_w = []
    for i in range(x_size * cm):
        x, y_rem = i % x_size, i >> x_logsize
        ret = 0
        for y in range(y_size):
            if y % cm == y_rem:
                idx = x + x_size * y
                ret += multirow[y // cm] * (p1["c"][idx] + p1["d"][idx] * us[1] + p2["c_pull"][idx] * us[2] + p2["d_pull"][idx] * us[3])
        combined_w.append(ret % P)
    nv = x_logsize + clm
    n = 1 << nv
    wit = [[(a + gamma * b) % P for a, b in zip(p1["p_0"], p1["p_1"])], list(p1["ac_c"]), list(p1["ac_d"]), combined_w]
    wit = [w + [0] * (n - len(w)) for w in wit]
    mo_claims = [(p_folded_point, (p_folded_ev - gamma * gamma) % P), (ac_c_point, ac_c_evs[0]), (ac_d_point, ac_d_evs[0]),
                 (combined_point, combined_ev)]
    mo_pt, mo_evs = PF.multiopen_prove(tr, nv, mo_claims, wit)
    q = tr.challenge(128)
    qs = PF.make_gamma_pows(q, 4)
    parts = [G.add(c1["p_0"], G.mul(c1["p_1"], gamma)), c1["ac_c"], c1["ac_d"], combined_comm]
    folded_comm = None
    for a, b in zip(qs, parts):
        folded_comm = G.add(folded_comm, G.mul(b, a))
    folded_w = [sum(wit[j][i] * qs[j] for j in range(4)) % P for i in range(n)]
    pts = []
    pair, proof = KN.knuckles_open(tr, pts, basis, kn_inverses, k, nv, folded_comm, mo_pt, zip_with_gamma(q, mo_evs), folded_w)
    return pair
