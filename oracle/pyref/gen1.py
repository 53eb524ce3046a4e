"""TEST INFRASTRUCTURE ONLY (oracle).

gen-1 prover `gkr_msm_prove` (bit-decomposed MSM circuit) restated for full shapes.  Follows
  /root/reference/src/gkr_msm_simple.rs:82-338          gkr_msm_prove, pt_bit_choice, layer list :248-269
  /root/reference/src/protocol/bintree.rs:81-123, 168-288  BintreeParams::unroll, witness, BintreeProver::round
  /root/reference/src/protocol/sumcheck.rs:55-257, 659-701 FragmentedLincomb {split, bind, unipoly}, SumcheckPolyMapProver::round,
                                                           make_folded_claim / make_folded_f
  /root/reference/src/protocol/split.rs:37-82            Split::witness, SplitProver::round
  /root/reference/src/polynomial/fragmented.rs:676-761   split (even / odd), bind_from, evaluate
  /root/reference/src/copoly.rs:457-633                  EqPoly {materialize_split, bind}
  /root/reference/src/utils.rs:104-113, 167-173          make_gamma_pows_legacy, fix_var_top / fix_var_bot
`gkr_msm_prove` only ever builds `Shape::full` polynomials (gkr_msm_simple.rs:150), for which a FragmentedPoly is a plain
vector, split is the even/odd de-interleave and the eq co-polynomial is the plain eq table; that is what is restated.
The G1 commitments (binary_msm / G::msm over BLS12-381, :120-147) are SURVEY 8f-1 and not part of this path.
Transcript stand-in: challenges are full field elements from a tape (gen-1 draws 64 bytes mod p, transcript.rs:96-101);
messages are recorded exactly as appended (full coefficient vectors, final evaluations, outputs).
"""
from .field import P
from .algfn import AlgFn, AFF_L1, AFF_L2, AFF_L3, PROJ_L1, PROJ_L2, PROJ_L3
from .polys import bind_dense, eq_poly_sequence_last
from .sumcheck import unipoly_from_evals


def pt_bit_choice(a):
    """gkr_msm_simple.rs:82-84"""
    return [a[0] * a[1] % P, (a[0] * (a[2] - 1) + 1) % P]


PT_BIT_CHOICE = AlgFn("pt_bit_choice", 2, 3, 2, pt_bit_choice)


class Tape:
    def __init__(self, tape):
        self.tape, self.pos, self.msgs = list(tape), 0, []

    def challenge(self):
        v = self.tape[self.pos] % P
        self.pos += 1
        return v

    def append_scalars(self, xs):
        self.msgs.extend(xs)


def layer_list(log_num_points):
    """gkr_msm_simple.rs:248-269 ; ("map", fn) / ("split", n)"""
    layers = [("map", PT_BIT_CHOICE), ("split", 2), ("map", AFF_L1), ("map", AFF_L2), ("map", AFF_L3)]
    for _ in range(log_num_points - 1):
        layers += [("split", 3), ("map", PROJ_L1), ("map", PROJ_L2), ("map", PROJ_L3)]
    return layers


def unroll(layers, num_vars):
    """bintree.rs:81-123"""
    out = []
    for l in layers:
        out.append((l, num_vars))
        if l[0] == "split":
            assert num_vars > 0
            num_vars -= 1
    assert out[-1][0][0] != "split"
    return out


def map_over_poly(polys, f):
    n = len(polys[0])
    outs = [[0] * n for _ in range(f.n_outs)]
    for i in range(n):
        r = f.exec([p[i] for p in polys])
        for o in range(f.n_outs):
            outs[o][i] = r[o]
    return outs


def split_witness(polys):
    """split.rs:37-52 : [l_0.., r_0..] with l = even, r = odd entries (fragmented.rs:676-732)"""
    return [p[0::2] for p in polys] + [p[1::2] for p in polys]


def bintree_witness(base, layers_unrolled):
    """bintree.rs:168-184 ; trace[i] = input of layer i"""
    trace, cur = [], base
    for (layer, _nv) in layers_unrolled:
        trace.append(cur)
        cur = map_over_poly(cur, layer[1]) if layer[0] == "map" else split_witness(cur)
    return trace, cur


def evaluate(poly, pt):
    """fragmented.rs:748-761"""
    cur = list(poly)
    for f in reversed(pt):
        cur = bind_dense(cur, f)
    return cur[0]


def base_layer(scalars_bits, points, log_num_points, log_num_scalar_bits):
    """gkr_msm_simple.rs:120, 150-186 : index = point * 2^lb + bit"""
    nb = 1 << log_num_scalar_bits
    assert len(points) == 1 << log_num_points and all(len(s) == nb for s in scalars_bits)
    bits = [1 if b else 0 for s in scalars_bits for b in s]
    px = [p[0] for p in points for _ in range(nb)]
    py = [p[1] for p in points for _ in range(nb)]
    return [bits, px, py]


def mapping_layer_prove(tr, f, num_vars, claim, polys, challenge_first):
    """SumcheckPolyMapProver (sumcheck.rs:185-257) driven to completion; returns the new EvalClaim.
    `challenge_first` is the challenge of the call that created the prover (= gamma)."""
    point, evs = claim
    gamma = challenge_first
    gp = [1, gamma]
    for i in range(2, len(evs)):
        gp.append(gp[i - 1] * gamma % P)
    polys = [list(p) for p in polys]
    eq = eq_poly_sequence_last(point)  # EqPoly(point) on the full shape
    rs = []
    if num_vars == 0:
        fe = [p[0] for p in polys]
        tr.append_scalars(fe[: f.n_ins])
        return (rs, fe[: f.n_ins])

    def unipoly():
        half = len(polys[0]) // 2
        res = []
        for k in range(f.deg + 2):
            acc = 0
            for i in range(half):
                args = [(p[2 * i] + k * (p[2 * i + 1] - p[2 * i])) % P for p in polys]
                e = (eq[2 * i] + k * (eq[2 * i + 1] - eq[2 * i])) % P
                out = f.exec(args)
                g = 0
                for o in range(len(evs)):
                    g = (g + out[o] * gp[o]) % P
                acc = (acc + g * e) % P
            res.append(acc)
        return unipoly_from_evals(res)

    tr.append_scalars(unipoly())
    while True:
        r = tr.challenge()
        rs.insert(0, r)  # fix_var_bot
        polys[:] = [bind_dense(p, r) for p in polys]
        eq[:] = bind_dense(eq, r)
        if len(rs) == num_vars:
            fe = [p[0] for p in polys]
            tr.append_scalars(fe[: f.n_ins])
            return (rs, fe[: f.n_ins])
        tr.append_scalars(unipoly())


def split_layer_prove(tr, claim, challenge):
    """split.rs:66-82"""
    point, evs = claim
    h = len(evs) // 2
    new = [(x + challenge * (y - x)) % P for x, y in zip(evs[:h], evs[h:])]
    return (list(point) + [challenge], new)  # fix_var_top


def gkr_msm_prove(scalars_bits, points, log_num_points, log_num_scalar_bits, tape):
    """gkr_msm_simple.rs:86-338 without the commitments; returns (final EvalClaim, output polys, transcript)"""
    tr = Tape(tape)
    nv = log_num_points + log_num_scalar_bits
    base = base_layer(scalars_bits, points, log_num_points, log_num_scalar_bits)
    layers = unroll(layer_list(log_num_points), nv)
    trace, output = bintree_witness(base, layers)
    for p in output:
        tr.append_scalars(p)
        assert len(p) == 1 << log_num_scalar_bits
    claim_point = [tr.challenge() for _ in range(log_num_scalar_bits)]
    claim = (claim_point, [evaluate(p, claim_point) for p in output])
    for (layer, lnv), inp in zip(reversed(layers), reversed(trace)):
        c = tr.challenge()
        if layer[0] == "map":
            claim = mapping_layer_prove(tr, layer[1], lnv, claim, inp, c)
        else:
            claim = split_layer_prove(tr, claim, c)
    return claim, output, tr
