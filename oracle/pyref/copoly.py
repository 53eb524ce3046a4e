"""TEST INFRASTRUCTURE ONLY (oracle).

gen-1 co-polynomials: standard subsets, the segment tree and the shape-aware eq tables.
Restates
  /root/reference/src/copoly.rs:18-62     (StandardSubset, count_trailing_zeros, log_floor)
  /root/reference/src/copoly.rs:103-136   (StSubIter)
  /root/reference/src/copoly.rs:139-148   (compute_segment_split)
  /root/reference/src/copoly.rs:150-253   (BinTreeNode, BinTree::from_segments / from_stsubs)
  /root/reference/src/copoly.rs:469-489   (materialize_eq_slice)
  /root/reference/src/copoly.rs:491-567   (EqPoly::materialize_eq_with_shape)
  /root/reference/src/copoly.rs:581-633   (EqPoly::bind, materialize, materialize_split)
  /root/reference/src/copoly.rs:637-700   (half_sums_standard_subset, materialize_standard_subset)
  /root/reference/src/copoly.rs:430-453   (half_sums_segment, ip_segment, materialize_segment)
Pinned by the reference's deterministic test `test_segment_split` (copoly.rs:852-869) and its identity tests
test_eq_sum / test_eq_materialize / test_eq_ip (copoly.rs:871-940) at their literal sizes: tests/test_ref_kats_cpu.py.
"""
from .field import P, inv
from .fragmented import CONSTS, DATA

Q_DATA, Q_SUM = "Data", "Sum"


def count_trailing_zeros(x):
    """copoly.rs:43-53 (64 for 0)"""
    if x == 0:
        return 64
    r = 0
    while x & 1 == 0:
        r += 1
        x >>= 1
    return r


def log_floor(x):
    """copoly.rs:55-62"""
    r = 0
    while x > 1:
        x >>= 1
        r += 1
    return r


class StandardSubset:
    def __init__(self, start, loglength):
        assert (start >> loglength) << loglength == start, "Start must be divisible by length."
        self.start, self.loglength = start, loglength

    def end(self):
        return self.start + (1 << self.loglength)


def compute_segment_split(start, end):
    """copoly.rs:139-148"""
    ret = []
    while start < end:
        ll = min(count_trailing_zeros(start), log_floor(end - start))
        ret.append(StandardSubset(start, ll))
        start += 1 << ll
    return ret


def stsub_iter(start, end, mem_idx, content):
    """copoly.rs:116-136: (start, logsize, mem_idx, content) queries of one segment"""
    out = []
    while start != end:
        ls = min(count_trailing_zeros(start), log_floor(end - start))
        out.append((start, ls, mem_idx, content))
        start += 1 << ls
        if content == Q_DATA:
            mem_idx += 1 << ls
    return out


class BinTree:
    """copoly.rs:164-253; nodes[depth] = list of (parent, is_r_child, is_leaf)"""

    def __init__(self, total_logsize, queries):
        self.total_logsize = total_logsize
        self.nodes = [[] for _ in range(total_logsize + 1)]
        self.sum_leaves, self.data_leaves = [], []
        meta = [[] for _ in range(total_logsize + 1)]
        prev_right_end = 0
        for i, (q_start, q_log, q_mem, q_content) in enumerate(queries):
            assert q_start >= prev_right_end, "query sequence is not properly ordered"
            prev_right_end = q_start + (1 << q_log)
            path = q_start >> q_log
            depth = total_logsize - q_log
            leaf = (depth, len(self.nodes[depth]), q_mem)
            (self.data_leaves if q_content == Q_DATA else self.sum_leaves).append(leaf)
            while depth > 0:
                is_leaf = depth == total_logsize - q_log
                bit = path % 2 == 1
                if not meta[depth - 1]:
                    self.nodes[depth].append([0, bit, is_leaf])
                    meta[depth].append(path)
                elif path >> 1 == meta[depth - 1][-1]:
                    assert bit, "should always happen"
                    self.nodes[depth].append([len(meta[depth - 1]) - 1, bit, is_leaf])
                    meta[depth].append(path)
                    break
                else:
                    self.nodes[depth].append([len(meta[depth - 1]), bit, is_leaf])
                    meta[depth].append(path)
                path >>= 1
                depth -= 1
            if i == 0:
                self.nodes[0].append([0, False, False])
                meta[0].append(0)
                if q_log == total_logsize:
                    self.nodes[0][0][2] = True
                    return

    @staticmethod
    def from_shape(total_logsize, shape):
        qs = []
        for f in shape.fragments:
            qs.extend(stsub_iter(f.start, f.start + f.len, f.mem_idx, Q_DATA if f.content == DATA else Q_SUM))
        return BinTree(total_logsize, qs)


def materialize_eq_slice(multiplier, point):
    """copoly.rs:469-489: point[0] is the most significant variable"""
    n = len(point)
    s = [0] * (1 << n)
    s[0] = multiplier % P
    for i in range(n):
        half = 1 << i
        pc = point[n - i - 1]
        for j in range(half):
            b = s[j] * pc % P
            s[half + j] = b
            s[j] = (s[j] - b) % P
    # step i sets bit i (from the LSB) of the index with point[n-1-i]: big-endian indexing, point[0] = MSB
    return s


class EqPoly:
    """copoly.rs:457-467; `point` in the caller's order (point[0] = MSB); bind pops from the back"""

    def __init__(self, point, multiplier=1):
        self.point = [v % P for v in point]
        self.multiplier = multiplier % P
        self.shape = None

    def num_vars(self):
        return len(self.point)

    def take_shape(self, shape):
        assert self.shape is None
        self.shape = shape

    def bind(self, value):
        """copoly.rs:581-588"""
        p0 = self.point.pop()
        self.multiplier = self.multiplier * ((p0 * value + (1 - p0) * (1 - value)) % P) % P
        if self.shape is not None:
            self.shape = self.shape.split()

    def materialize_eq_with_shape(self, shape):
        """copoly.rs:492-567 -> (values, sums)"""
        n = self.num_vars()
        tree = BinTree.from_shape(n, shape)
        mult = [[[self.multiplier, None]]]
        if not tree.nodes[0][0][2]:
            mult[0][0][1] = mult[0][0][0] * self.point[0] % P
        for i in range(1, n + 1):
            row = []
            for parent, is_r, is_leaf in tree.nodes[i]:
                pm = mult[i - 1][parent]
                m = pm[1] if is_r else (pm[0] - pm[1]) % P
                row.append([m, None if is_leaf else self.point[i] * m % P])
            mult.append(row)
        sums = [0] * shape.num_consts
        for depth, idx, mem_idx in tree.sum_leaves:
            sums[mem_idx] = (sums[mem_idx] + mult[depth][idx][0]) % P
        values = [None] * shape.data_len
        for depth, idx, mem_idx in tree.data_leaves:
            sl = materialize_eq_slice(mult[depth][idx][0], self.point[depth:])
            values[mem_idx:mem_idx + len(sl)] = sl
        assert all(v is not None for v in values)
        return values, sums

    def materialize(self):
        return self.materialize_eq_with_shape(self.shape)

    def materialize_split(self):
        """copoly.rs:600-633 -> ((values, sums) even half, (values, sums) odd half)"""
        point = list(self.point)
        m1 = point.pop()
        m0 = (1 - m1) % P
        if m0 == 0:
            eq1 = EqPoly(point, m1 * self.multiplier)
            b = eq1.materialize_eq_with_shape(self.shape.split())
            a = ([0] * len(b[0]), [0] * len(b[1]))
        else:
            m = m1 * inv(m0) % P
            eq0 = EqPoly(point, m0 * self.multiplier)
            a = eq0.materialize_eq_with_shape(self.shape.split())
            b = ([x * m % P for x in a[0]], [x * m % P for x in a[1]])
        return a, b

    def half_sums_standard_subset(self, ss):
        """copoly.rs:637-658"""
        ll = ss.loglength
        prefix = ss.start >> ll
        s = self.multiplier
        n = self.num_vars()
        assert ss.end() <= 1 << n
        for i in reversed(range(n - ll)):
            s = s * (self.point[i] if prefix & 1 else (1 - self.point[i]) % P) % P
            prefix >>= 1
        if ll == 0:
            return (s, 0) if ss.start % 2 == 0 else (0, s)
        dif = s * self.point[n - 1] % P
        return (s - dif) % P, dif

    def materialize_standard_subset(self, ss):
        """copoly.rs:667-699"""
        ll = ss.loglength
        n = self.num_vars()
        assert ss.end() <= 1 << n
        prefix = ss.start >> ll
        m = self.multiplier
        for i in reversed(range(n - ll)):
            m = m * (self.point[i] if prefix & 1 else (1 - self.point[i]) % P) % P
            prefix >>= 1
        target = [0] * (1 << ll)
        target[0] = m
        pt = self.point[n - ll:]
        cur = 1
        for i in reversed(range(ll)):
            # the doubling that uses pt[i] sets index bit (ll-1-i): big-endian, as materialize_eq_slice
            for j in range(cur):
                a = target[j]
                target[cur + j] = pt[i] * a % P
                target[j] = a * ((1 - pt[i]) % P) % P
            cur <<= 1
        return target

    def half_sums_segment(self, start, end):
        s0 = s1 = 0
        for ss in compute_segment_split(start, end):
            a, b = self.half_sums_standard_subset(ss)
            s0, s1 = (s0 + a) % P, (s1 + b) % P
        return s0, s1

    def materialize_segment(self, start, end):
        out = []
        for ss in compute_segment_split(start, end):
            out.extend(self.materialize_standard_subset(ss))
        assert len(out) == end - start
        return out

    def ip_segment(self, start, end, values):
        t = self.materialize_segment(start, end)
        return sum(a * b for a, b in zip(t, values)) % P
