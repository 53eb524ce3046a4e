"""TEST INFRASTRUCTURE ONLY (oracle).

Dense and VecVec multilinear polynomials, eq tables.  Restates
  /root/reference/src/cleanup/polys/dense.rs:22-184
  /root/reference/src/cleanup/polys/vecvec.rs:68-160, 400-654
  /root/reference/src/utils.rs:137-154, 189-291
  /root/reference/src/cleanup/protocols/splits.rs:12-50
Canonical ints mod P; lists of ints.
"""
from .field import P


# ------------------------------------------------------------------ SplitIdx
class SplitIdx:
    def __init__(self, kind, v):
        assert kind in ("LO", "HI")
        self.kind, self.v = kind, v

    def lo_usize(self, num_vars):
        return self.v if self.kind == "LO" else num_vars - self.v - 1

    def hi_usize(self, num_vars):
        return self.v if self.kind == "HI" else num_vars - self.v - 1

    def to_hi(self, num_vars):
        return SplitIdx("HI", self.hi_usize(num_vars))


def LO(v):
    return SplitIdx("LO", v)


def HI(v):
    return SplitIdx("HI", v)


def log2_exact(n):
    # liblasso Math::log_2 : for powers of two = trailing zeros, else ceil(log2)
    assert n > 0
    l = n.bit_length() - 1
    return l if (1 << l) == n else l + 1


def interleave_bundles(l, r, bundle):
    """dense.rs:137-138 / vecvec.rs:596-597 : [L-bundle, R-bundle, L-bundle, ...]"""
    out = []
    lc = [l[i:i + bundle] for i in range(0, len(l), bundle)]
    rc = [r[i:i + bundle] for i in range(0, len(r), bundle)]
    i = 0
    # itertools::interleave alternates until both exhausted
    while i < len(lc) or i < len(rc):
        if i < len(lc):
            out += lc[i]
        if i < len(rc):
            out += rc[i]
        i += 1
    return out


# ------------------------------------------------------------------ dense
def bind_dense(poly, t):
    """sumcheck.rs:160-163"""
    return [(poly[2 * i] + t * (poly[2 * i + 1] - poly[2 * i])) % P for i in range(len(poly) // 2)]


def evaluate_dense(poly, pt):
    """dense.rs:22-31 (binds LSB first with the reversed point)."""
    assert len(poly) == 1 << len(pt)
    cur = list(poly)
    for f in reversed(pt):
        cur = bind_dense(cur, f)
    return cur[0]


def dense_make_21(v):
    """dense.rs:99-112 (in place)"""
    for i in range(len(v) // 2):
        v[2 * i] = (2 * v[2 * i + 1] - v[2 * i]) % P


def dense_bind_21(v, t):
    """dense.rs:54-61 (parallel variant: no odd padding) -> new list"""
    tm1 = (t - 1) % P
    return [(v[2 * i + 1] + tm1 * (v[2 * i] - v[2 * i + 1])) % P for i in range(len(v) // 2)]


def dense_algfn_map(polys, f):
    """dense.rs:141-184"""
    n = len(polys[0])
    outs = [[0] * n for _ in range(f.n_outs)]
    for idx in range(n):
        r = f.exec([p[idx] for p in polys])
        for o in range(f.n_outs):
            outs[o][idx] = r[o]
    return outs


def dense_algfn_map_split(polys, f, var_idx, bundle):
    """dense.rs:115-139"""
    n = len(polys[0])
    num_vars = log2_exact(n)
    seg = 1 << var_idx.lo_usize(num_vars)
    outs = [[[] for _ in range(f.n_outs)] for _ in range(2)]
    for idx in range(n):
        r = f.exec([p[idx] for p in polys])
        tgt = outs[(idx // seg) % 2]
        for o in range(f.n_outs):
            tgt[o].append(r[o])
    return interleave_bundles(outs[0], outs[1], bundle)


# ------------------------------------------------------------------ eq tables
def eq_poly_sequence_from_multiplier(mult, pt):
    """utils.rs:222-250 ; pt[0] is the MSB."""
    ret = [[mult % P]]
    for i in range(1, len(pt) + 1):
        last = ret[i - 1]
        m_ = pt[i - 1]
        inc = [0] * (1 << i)
        for j in range(1 << (i - 1)):
            w = last[j]
            m = m_ * w % P
            inc[2 * j] = (w - m) % P
            inc[2 * j + 1] = m
        ret.append(inc)
    return ret


def eq_poly_sequence(pt):
    return eq_poly_sequence_from_multiplier(1, pt)


def eq_poly_sequence_last(pt):
    return eq_poly_sequence(pt)[-1]


def padded_eq_poly_sequence(padding_size, pt):
    """utils.rs:189-220"""
    l = len(pt)
    ret = [[1]]
    for i in range(1, padding_size + 1):
        ret.append([ret[i - 1][0] * (1 - pt[i - 1]) % P])
    for i in range(padding_size + 1, l + 1):
        last = ret[i - 1]
        m_ = pt[i - 1]
        inc = [0] * (1 << (i - padding_size))
        for j in range(1 << (i - 1 - padding_size)):
            w = last[j]
            m = m_ * w % P
            inc[2 * j] = (w - m) % P
            inc[2 * j + 1] = m
        ret.append(inc)
    return ret


def eq_sum(pt, k):
    """utils.rs:265-291 : sum_{i<k} eq(pt, i)"""
    n = len(pt)
    if k >= (1 << n):
        assert k == 1 << n
        return 1
    mult, acc = 1, 0
    for i in range(n):
        left_bit = k >> (n - i - 1)
        prev = mult
        if left_bit == 1:
            mult = mult * pt[i] % P
            acc = (acc + prev - mult) % P
        else:
            mult = mult * (1 - pt[i]) % P
        k -= left_bit << (n - i - 1)
    return acc


def eq_eval(p1, p2):
    """utils.rs:150-154"""
    assert len(p1) == len(p2)
    r = 1
    for a, b in zip(p1, p2):
        r = r * ((1 - a - b + 2 * a * b) % P) % P
    return r


def evaluate_poly(poly, pt):
    """cleanup/utils/arith.rs:6-9"""
    e = eq_poly_sequence_last(pt)
    assert len(e) == len(poly)
    return sum(a * b for a, b in zip(poly, e)) % P


def zip_with_gamma(gamma, vals):
    """utils.rs:137-148 == sumcheck.rs:591-602 gamma_rlc (Horner, gamma^0 on vals[0])"""
    if not vals:
        return 0
    ret = vals[-1]
    for i in range(len(vals) - 1):
        ret = (ret * gamma + vals[len(vals) - i - 2]) % P
    return ret


gamma_rlc = zip_with_gamma


def make_gamma_pows(gamma, count):
    """utils.rs:126-135 (NB: always emits at least [1, gamma])"""
    g = [1, gamma % P]
    for i in range(2, count):
        g.append(g[i - 1] * gamma % P)
    return g


# ------------------------------------------------------------------ VecVec
class VecVec:
    """vecvec.rs:149-160 ; one polynomial."""

    def __init__(self, data, row_pad, col_pad, row_logsize, col_logsize, checked=True):
        self.data = [list(r) for r in data]
        self.row_pad, self.col_pad = row_pad, col_pad
        self.row_logsize, self.col_logsize = row_logsize, col_logsize
        if checked:  # vecvec.rs:178-189
            assert len(self.data) <= (1 << col_logsize)
            for r in self.data:
                assert len(r) <= (1 << row_logsize)
                if len(r) % 2 == 1:
                    r.append(row_pad)

    def clone(self):
        return VecVec(self.data, self.row_pad, self.col_pad, self.row_logsize, self.col_logsize, checked=False)

    def num_vars(self):
        return self.row_logsize + self.col_logsize

    def to_dense(self):
        """vecvec.rs:446-461"""
        ret = []
        for r in range(1 << self.col_logsize):
            for c in range(1 << self.row_logsize):
                if r >= len(self.data):
                    ret.append(self.col_pad)
                elif c >= len(self.data[r]):
                    ret.append(self.row_pad)
                else:
                    ret.append(self.data[r][c])
        return ret

    def make_21(self):
        """vecvec.rs:400-413"""
        for r in self.data:
            for i in range(len(r) // 2):
                r[2 * i] = (2 * r[2 * i + 1] - r[2 * i]) % P

    def bind_21(self, t):
        """vecvec.rs:420-441"""
        tm1 = (t - 1) % P
        for k, r in enumerate(self.data):
            half = len(r) // 2
            new = [(r[2 * i + 1] + tm1 * (r[2 * i] - r[2 * i + 1])) % P for i in range(half)]
            if half % 2 == 1:
                new.append(self.row_pad)
            self.data[k] = new
        self.row_logsize -= 1


def vecvec_map(polys, f):
    """vecvec.rs:480-540"""
    row_logsize, col_logsize = polys[0].row_logsize, polys[0].col_logsize
    row_pad = f.exec([p.row_pad for p in polys])
    col_pad = f.exec([p.col_pad for p in polys])
    outs = [[] for _ in range(f.n_outs)]
    for ri in range(len(polys[0].data)):
        rows = [[] for _ in range(f.n_outs)]
        for idx in range(len(polys[0].data[ri])):
            r = f.exec([p.data[ri][idx] for p in polys])
            for o in range(f.n_outs):
                rows[o].append(r[o])
        for o in range(f.n_outs):
            outs[o].append(rows[o])
    return [VecVec(outs[o], row_pad[o], col_pad[o], row_logsize, col_logsize) for o in range(f.n_outs)]


def vecvec_map_split(polys, f, var_idx, bundle):
    """vecvec.rs:542-606"""
    num_vars = polys[0].num_vars()
    row_logsize, col_logsize = polys[0].row_logsize, polys[0].col_logsize
    row_pad = f.exec([p.row_pad for p in polys])
    col_pad = f.exec([p.col_pad for p in polys])
    seg = 1 << (var_idx.v if var_idx.kind == "LO" else num_vars - 1 - var_idx.v)
    nrows = len(polys[0].data)
    outs = [[[[] for _ in range(nrows)] for _ in range(f.n_outs)] for _ in range(2)]
    for ri in range(nrows):
        for idx in range(len(polys[0].data[ri])):
            r = f.exec([p.data[ri][idx] for p in polys])
            tgt = outs[(idx // seg) % 2]
            for o in range(f.n_outs):
                tgt[o][ri].append(r[o])
        if f.n_outs > 0 and len(outs[0][0][ri]) % 2 == 1:
            for oi in range(2):
                for o in range(f.n_outs):
                    outs[oi][o][ri].append(row_pad[o])
    l = [(outs[0][o], row_pad[o], col_pad[o]) for o in range(f.n_outs)]
    r = [(outs[1][o], row_pad[o], col_pad[o]) for o in range(f.n_outs)]
    return [VecVec(d, rp, cp, row_logsize - 1, col_logsize, checked=False)
            for (d, rp, cp) in interleave_bundles(l, r, bundle)]


def vecvec_map_split_to_dense(polys, f, var_idx, bundle):
    """vecvec.rs:608-654"""
    num_vars = polys[0].num_vars()
    assert polys[0].row_logsize == 1
    col_logsize = polys[0].col_logsize
    row_pad = f.exec([p.row_pad for p in polys])
    col_pad = f.exec([p.col_pad for p in polys])
    seg = 1 << (var_idx.v if var_idx.kind == "LO" else num_vars - 1 - var_idx.v)
    outs = [[[] for _ in range(f.n_outs)] for _ in range(2)]
    for ri in range(len(polys[0].data)):
        for idx in range(len(polys[0].data[ri])):
            r = f.exec([p.data[ri][idx] for p in polys])
            tgt = outs[(idx // seg) % 2]
            for o in range(f.n_outs):
                tgt[o].append(r[o])
        if len(outs[0][0]) < ri + 1:
            for oi in range(2):
                for o in range(f.n_outs):
                    outs[oi][o].append(row_pad[o])
    l = list(enumerate(outs[0]))
    r = list(enumerate(outs[1]))
    res = []
    for (idx, data) in interleave_bundles(l, r, bundle):
        data = list(data) + [col_pad[idx]] * ((1 << col_logsize) - len(data))
        res.append(data)
    return res


class EQPolyData:
    """vecvec.rs:19-147"""

    def __init__(self, point, col_logsize, max_row_len):
        max_seg_log = log2_exact(max_row_len)
        self.padded_vars_idx = col_logsize
        self.segment_vars_idx = len(point) - max_seg_log
        self.binding_var_idx = len(point) - 1
        self.point = list(point)
        self.multiplier = 1
        self.row_eq_coefs = eq_poly_sequence_last(self.point[0:self.padded_vars_idx])
        tails, acc = [], 0
        for v in reversed(self.row_eq_coefs):
            acc = (acc + v) % P
            tails.append(acc)
        tails.reverse()
        self.row_eq_coefs_tail_sums = tails
        pr = self.padded_vars_range()
        rr = self.row_vars_range()
        self.row_eq_poly_seq = padded_eq_poly_sequence(len(range(*pr)), self.point[rr[0]:rr[1]])
        self.row_eq_poly_prefix_seq = []
        for v in self.row_eq_poly_seq:
            acc = [0]
            for x in v:
                acc.append((acc[-1] + x) % P)
            self.row_eq_poly_prefix_seq.append(acc)
        self.already_bound_vars = 0

    def padded_vars_range(self):
        hi = min(self.segment_vars_idx, self.binding_var_idx)
        return (self.padded_vars_idx, max(hi, self.padded_vars_idx))

    def row_vars_range(self):
        return (self.padded_vars_idx, max(self.segment_vars_idx, self.binding_var_idx))

    def bind(self, t):
        q = self.point[self.binding_var_idx]
        self.multiplier = self.multiplier * ((1 - q - t + 2 * q * t) % P) % P
        if self.binding_var_idx == 0:
            self.binding_var_idx = None
        else:
            self.binding_var_idx -= 1
        self.already_bound_vars += 1

    def get_segment_evals(self, seg_len):
        return self.row_eq_poly_seq[len(self.row_eq_poly_seq) - 1 - self.already_bound_vars][0:seg_len]

    def get_trailing_sum(self, seg_len):
        return (1 - self.row_eq_poly_prefix_seq[len(self.row_eq_poly_prefix_seq) - 1 - self.already_bound_vars][seg_len]) % P
