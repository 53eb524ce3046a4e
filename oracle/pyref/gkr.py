"""TEST INFRASTRUCTURE ONLY (oracle).

gen-2 GKR circuits: bucket sums (bintree_add), bucket reduction (triangle_add),
the Pippenger bucketing ("pushforward" image) and the "prove image part" driver.
Restates
  /root/reference/src/cleanup/protocols/gkrs/bintree_add.rs:128-375
  /root/reference/src/cleanup/protocols/gkrs/triangle_add.rs:83-232
  /root/reference/src/cleanup/protocols/gkrs/gkr.rs:39-58
  /root/reference/src/cleanup/protocols/pippenger_ending.rs:26-163
  /root/reference/src/cleanup/protocols/splits.rs:112-202
  /root/reference/src/cleanup/protocols/zero_check.rs:17-33
  /root/reference/src/cleanup/protocols/pushforward/pushforward.rs:329-429 (Fr/Bandersnatch part only;
      the BLS12-381 G1 outer buckets are SURVEY 8f-1, out of first-pass scope)
  /root/reference/src/cleanup/protocols/pippenger.rs:500-606 (claims + final recombination)
"""
from .field import P, te_add_affine, te_double_affine, proj_to_affine
from .algfn import (AFF_L1, AFF_L2, AFF_L3, PROJ_L1, PROJ_L2, PROJ_L3, TRI_L1,
                    BitCheckFn, IdAlgFn, RepeatedAlgFn, StackedAlgFn)
from .polys import (HI, LO, SplitIdx, VecVec, dense_algfn_map, dense_algfn_map_split, evaluate_poly,
                    vecvec_map, vecvec_map_split, vecvec_map_split_to_dense)
from .sumcheck import dense_deg2_sumcheck_prove, vecvec_deg2_sumcheck_prove

EMPTY = ("EMPTY", None)


def _adv_map(advice, f):
    kind, polys = advice
    if kind == "VV":
        return ("VV", vecvec_map(polys, f))
    assert kind == "D"
    return ("D", dense_algfn_map(polys, f))


def _adv_map_split(advice, f, layer_idx, row_logsize, idx, bundle):
    kind, polys = advice
    if kind == "VV":
        if layer_idx + 2 == row_logsize:
            return ("D", vecvec_map_split_to_dense(polys, f, idx, bundle))
        return ("VV", vecvec_map_split(polys, f, idx, bundle))
    assert kind == "D"
    return ("D", dense_algfn_map_split(polys, f, idx, bundle))


# ------------------------------------------------------------------ bintree_add
def bintree_witness_build(advice, row_logsize, num_adds, do_bitcheck):
    """bintree_add.rs:137-184 ; returns the advice list (inputs of each layer)."""
    assert num_adds > 0
    advices = []
    for add_idx in range(num_adds):
        last = add_idx + 1 == num_adds
        for step in ("L1", "L2", "L3"):
            if step == "L1":
                nxt = _adv_map(advice, AFF_L1 if add_idx == 0 else PROJ_L1)
            elif step == "L2":
                nxt = _adv_map(advice, AFF_L2 if add_idx == 0 else PROJ_L2)
            else:
                nxt = None if last else _adv_map_split(
                    advice, AFF_L3 if add_idx == 0 else PROJ_L3, add_idx, row_logsize, LO(0), 3)
            advices.append(advice)
            if add_idx == 0 and step == "L1" and do_bitcheck:
                advices.append(EMPTY)
            advice = nxt
        if not last:
            advices.append(EMPTY)
    return advices


def bintree_last_step(advice, layer_idx):
    """bintree_add.rs:128-135"""
    return _adv_map(advice, AFF_L3 if layer_idx == 0 else PROJ_L3)


def bintree_protocol_layers(num_vars, num_adds, row_logsize, do_bitcheck):
    """bintree_add.rs:247-375 ; list of layer descriptors in forward order."""
    layers = []
    nvert = num_vars - row_logsize
    for i in range(num_adds):
        for step in ("L1", "L2", "L3"):
            vv = (i == 0) or (i + 1 < row_logsize)
            if i == 0:
                f = {"L1": AFF_L1, "L2": AFF_L2, "L3": AFF_L3}[step]
                if step == "L1" and do_bitcheck:
                    f = StackedAlgFn(AFF_L1, RepeatedAlgFn(BitCheckFn(), 2))
            else:
                f = {"L1": PROJ_L1, "L2": PROJ_L2, "L3": PROJ_L3}[step]
            if vv:
                layers.append(("vecvec", f, num_vars - i - 1, nvert))
            else:
                layers.append(("dense", f, num_vars - i - 1))
            if i == 0 and step == "L1" and do_bitcheck:
                layers.append(("zerocheck",))
        if i != num_adds - 1:
            layers.append(("split", LO(0), 3))
    return layers


# ------------------------------------------------------------------ triangle_add
def triangle_witness_build(advice, num_vars, split_idx):
    """triangle_add.rs:101-158"""
    split_idx = split_idx.to_hi(num_vars)
    num_layers = num_vars - split_idx.hi_usize(num_vars)
    advices = []
    for layer_idx in range(num_layers + 1):
        for step in ("L1", "L2", "L3"):
            if step == "L1":
                nxt = dense_algfn_map(advice, StackedAlgFn(TRI_L1, RepeatedAlgFn(PROJ_L1, layer_idx)))
            elif step == "L2":
                nxt = dense_algfn_map(advice, RepeatedAlgFn(PROJ_L2, layer_idx + 3))
            else:
                nxt = None if layer_idx == num_layers else dense_algfn_map_split(
                    advice, RepeatedAlgFn(PROJ_L3, layer_idx + 3), split_idx, 3)
            advices.append(("D", advice))
            advice = nxt
        if layer_idx < num_layers:
            advices.append(EMPTY)
    return advices


def triangle_last_step(advice, layer_idx):
    """triangle_add.rs:88-99"""
    return dense_algfn_map(advice, RepeatedAlgFn(PROJ_L3, layer_idx + 3))


def triangle_protocol_layers(num_vars, split_idx):
    """triangle_add.rs:173-232"""
    split_idx = split_idx.to_hi(num_vars)
    num_layers = num_vars - split_idx.hi_usize(num_vars)
    layers = []
    for layer_idx in range(num_layers + 1):
        layers.append(("dense", StackedAlgFn(TRI_L1, RepeatedAlgFn(PROJ_L1, layer_idx)), num_vars - layer_idx))
        layers.append(("dense", RepeatedAlgFn(PROJ_L2, layer_idx + 3), num_vars - layer_idx))
        layers.append(("dense", RepeatedAlgFn(PROJ_L3, layer_idx + 3), num_vars - layer_idx))
        if layer_idx < num_layers:
            layers.append(("split", split_idx, 3))
    return layers


# ------------------------------------------------------------------ claim-only layers
def split_at_prove(transcript, claims, var_idx, bundle):
    """splits.rs:121-143"""
    r = transcript.challenge(128)
    point, evs = claims
    chunks = [evs[i:i + bundle] for i in range(0, len(evs), bundle)]
    evs_l = [x for c in chunks[0::2] for x in c]
    evs_r = [x for c in chunks[1::2] for x in c]
    new = [(x + r * (y - x)) % P for x, y in zip(evs_l, evs_r)]
    point = list(point)
    pos = len(point) - var_idx.v if var_idx.kind == "LO" else var_idx.v
    point.insert(pos, r)
    return (point, new)


def glue_split_witness(polys):
    """splits.rs:172-176"""
    out = vecvec_map_split(polys[0:2], IdAlgFn(2), LO(0), 2)
    out += vecvec_map_split(polys[2:3], IdAlgFn(1), LO(0), 1)
    return out


def glue_split_prove(transcript, claims):
    """splits.rs:185-197"""
    r = transcript.challenge(128)
    point, evs = claims
    new = [(evs[0] + r * (evs[2] - evs[0])) % P, (evs[1] + r * (evs[3] - evs[1])) % P,
           (evs[4] + r * (evs[5] - evs[4])) % P]
    return (list(point) + [r], new)


def simple_gkr_prove(transcript, layers, advices, claims, record=None):
    """gkr.rs:45-50 : layers reversed, advices popped from the back"""
    advices = list(advices)
    for layer in reversed(layers):
        adv = advices.pop()
        kind = layer[0]
        if kind == "vecvec":
            assert adv[0] == "VV", (adv[0], layer[1].name)
            claims = vecvec_deg2_sumcheck_prove(transcript, layer[1], layer[2], layer[3], claims, adv[1], record)
        elif kind == "dense":
            assert adv[0] == "D", (adv[0], layer[1].name)
            claims = dense_deg2_sumcheck_prove(transcript, layer[1], layer[2], claims, adv[1], record)
        elif kind == "split":
            assert adv is EMPTY
            claims = split_at_prove(transcript, claims, layer[1], layer[2])
        elif kind == "zerocheck":
            assert adv is EMPTY
            claims = (claims[0], list(claims[1]) + [0, 0])
        else:
            raise ValueError(kind)
    assert not advices
    return claims


# ------------------------------------------------------------------ pippenger ending
class PippengerEndingWG:
    """pippenger_ending.rs:32-99 (the reference builds the bintree witness twice; once is enough)."""

    def __init__(self, multirow_vars, bucket_vars, horizontal_vars, inputs):
        assert len(inputs) == 6
        self.bintree_advices = bintree_witness_build(("VV", [p.clone() for p in inputs]),
                                                     horizontal_vars, horizontal_vars, True)
        last = bintree_last_step(self.bintree_advices[-1], horizontal_vars - 1)[1]
        self.bucket_sums = last
        s1 = dense_algfn_map_split(last, IdAlgFn(3), HI(multirow_vars), 3)
        s2 = dense_algfn_map_split(s1, RepeatedAlgFn(IdAlgFn(3), 2), HI(multirow_vars), 3)
        self.triangle_advices = triangle_witness_build(s2, multirow_vars + bucket_vars - 2, HI(multirow_vars))

    def last(self):
        return self.triangle_advices[-1][1]


def pippenger_bucketed_prove(transcript, multirow_vars, bucket_vars, horizontal_vars, claims, wg, record=None):
    """pippenger_ending.rs:142-149"""
    tri_layers = triangle_protocol_layers(multirow_vars + bucket_vars - 2, HI(multirow_vars))
    claims = simple_gkr_prove(transcript, tri_layers, wg.triangle_advices, claims, record)
    claims = split_at_prove(transcript, claims, HI(multirow_vars), 3)
    claims = split_at_prove(transcript, claims, HI(multirow_vars), 3)
    bt_layers = bintree_protocol_layers(multirow_vars + bucket_vars + horizontal_vars, horizontal_vars,
                                        horizontal_vars, True)
    claims = simple_gkr_prove(transcript, bt_layers, wg.bintree_advices, claims, record)
    return claims


# ------------------------------------------------------------------ bucketing
def scalar_digits(coefs, y_size, d_logsize):
    """pushforward.rs:351-361 : digits[y][x] = d-bit window y of the canonical scalar"""
    mask = (1 << d_logsize) - 1
    return [[(c >> (y * d_logsize)) & mask for c in coefs] for y in range(y_size)]


def bucketing_image(points, coefs, y_size, y_logsize, d_logsize, x_logsize):
    """pushforward.rs:342-349, 380-429, 477-487 ; returns (image[3], digits, counter)"""
    x_size = 1 << x_logsize
    assert len(points) == x_size
    polys = [[p[0] for p in points], [p[1] for p in points], [1] * x_size]
    digits = scalar_digits(coefs, y_size, d_logsize)
    row_pad = col_pad = [0, 1, 0]
    counter = [[0] * x_size for _ in range(y_size)]
    buckets = [[[] for _ in range(3)] for _ in range(y_size << d_logsize)]
    for y in range(y_size):
        for x in range(x_size):
            d = digits[y][x]
            b = buckets[(y << d_logsize) + d]
            counter[y][x] = len(b[0])
            for pid in range(3):
                b[pid].append(polys[pid][x])
    image = [VecVec([buckets[r][pid] for r in range(y_size << d_logsize)], row_pad[pid], col_pad[pid],
                    x_logsize, y_logsize + d_logsize) for pid in range(3)]
    return image, digits, counter


def pippenger_witness(points, coefs, y_size, y_logsize, d_logsize, x_logsize):
    """pippenger.rs:37-70 (PippengerWG::new without the G1 commitments)"""
    image, digits, counter = bucketing_image(points, coefs, y_size, y_logsize, d_logsize, x_logsize)
    wg = PippengerEndingWG(y_logsize, d_logsize, x_logsize, glue_split_witness(image))
    return image, digits, counter, wg


def pippenger_dense_output(wg, y_logsize, d_logsize):
    """pippenger.rs:531-534"""
    nv = y_logsize + d_logsize - 2
    return triangle_last_step(wg.last(), nv - HI(y_logsize).hi_usize(nv))


def pippenger_claims(dense_output, r):
    """pippenger.rs:536-539"""
    return (list(r), [evaluate_poly(o, r) for o in dense_output])


def pippenger_final_point(results, d_logsize):
    """pippenger.rs:586-602 : Horner double-and-add over the transposed output points; affine result"""
    assert (d_logsize + 1) * 3 == len(results)
    pts = []
    for c in range(0, len(results), 3):
        X, Y, Z = results[c], results[c + 1], results[c + 2]
        pts.append([proj_to_affine(X[i], Y[i], Z[i]) for i in range(len(X))])
    transposed = []
    for idx in range(len(pts[0])):
        for i in range(1, len(pts)):
            transposed.append(pts[i][idx])
    acc = (0, 1)
    for p in reversed(transposed):
        acc = te_double_affine(acc)
        acc = te_add_affine(acc, p)
    return acc


def prove_image_part(transcript, y_logsize, d_logsize, x_logsize, claims, wg, record=None):
    """pippenger.rs:138-141 : ending.prove + GlueSplit.prove"""
    claims = pippenger_bucketed_prove(transcript, y_logsize, d_logsize, x_logsize, claims, wg, record)
    return glue_split_prove(transcript, claims)


# ------------------------------------------------------------------ pushforward phase data (Fr part)
def pushforward_phase1_polys(digits, counter, x_logsize, d_logsize):
    """pushforward.rs:489-510 : c, d as field elements (flattened [y][x]) and the negated access counts"""
    d = [v % P for row in digits for v in row]
    c = [v % P for row in counter for v in row]
    ac_d = [0] * (1 << d_logsize)
    ac_c = [0] * (1 << x_logsize)
    for row in digits:
        for v in row:
            ac_d[v] += 1
    for row in counter:
        for v in row:
            ac_c[v] += 1
    return c, d, [(-v) % P for v in ac_c], [(-v) % P for v in ac_d]


def pushforward_second_phase(digits, counter, r, y_logsize, d_logsize, x_logsize):
    """pushforward.rs:572-596 : point layout [r_y | r_d | r_c]; c_pull = eq_c[counter], d_pull = eq_d[digit]"""
    from .polys import eq_poly_sequence_last
    assert len(r) == y_logsize + d_logsize + x_logsize
    r_d = r[y_logsize:y_logsize + d_logsize]
    r_c = r[y_logsize + d_logsize:]
    eq_c = eq_poly_sequence_last(r_c)
    eq_d = eq_poly_sequence_last(r_d)
    c_pull = [eq_c[v] for row in counter for v in row]
    d_pull = [eq_d[v] for row in digits for v in row]
    return c_pull, d_pull
