"""TEST INFRASTRUCTURE ONLY (oracle).

gen-1 fragmented multilinear polynomial: Shape / Fragment / FragmentedPoly.
Restates
  /root/reference/src/polynomial/fragmented.rs:29-62    (FragmentContent, Fragment, Shape)
  /root/reference/src/polynomial/fragmented.rs:65-78    (MERGE_THRESH, should_merge)
  /root/reference/src/polynomial/fragmented.rs:80-183   (Shape::{len, new, full, merge_in, add, finalize})
  /root/reference/src/polynomial/fragmented.rs:280-365  (Shape::{split, full_split, prune_consts})
  /root/reference/src/polynomial/fragmented.rs:383-428  (FragmentedPoly::{new, len, get_by_fragment})
  /root/reference/src/polynomial/fragmented.rs:526-674  (split_at)
  /root/reference/src/polynomial/fragmented.rs:676-732  (split)
  /root/reference/src/polynomial/fragmented.rs:736-761  (bind_from, bind, evaluate)
  /root/reference/src/polynomial/fragmented.rs:831-846  (into_vec)
Pinned by the reference's own integer KATs `split_poly` / `split_shape` (fragmented.rs:974-1164): tests/test_ref_kats_cpu.py.
Elements are plain Python ints; `mod` = None keeps them integers (the KATs use u64), `mod` = P reduces.
"""
DATA = "Data"
CONSTS = "Consts"
MERGE_THRESH = 2


class Fragment:
    __slots__ = ("mem_idx", "len", "content", "start")

    def __init__(self, mem_idx, len, content, start):
        self.mem_idx, self.len, self.content, self.start = mem_idx, len, content, start

    def key(self):
        return (self.mem_idx, self.len, self.content, self.start)

    def __eq__(self, o):
        return self.key() == o.key()

    def __repr__(self):
        return "Fragment(mem_idx=%d, len=%d, %s, start=%d)" % (self.mem_idx, self.len, self.content, self.start)


def should_merge(f1, f2):
    """fragmented.rs:67-78"""
    if f1.content == DATA and f2.content == DATA:
        return True
    if f1.content == DATA and f2.content == CONSTS:
        return f2.len < MERGE_THRESH
    if f1.content == CONSTS and f2.content == DATA:
        return False
    return f1.mem_idx == f2.mem_idx


class Shape:
    def __init__(self, fragments, num_consts):
        """Shape::new (fragmented.rs:94-99): takes the fragments as given (no merging) and recounts (finalize :168-183)."""
        self.fragments = [Fragment(*f.key()) for f in fragments]
        self.num_consts = num_consts
        self.data_len = 0
        self.dedup_consts_len = 0
        self._split = None
        self._split_perm = None
        for f in self.fragments:
            if f.content == DATA:
                assert f.mem_idx == self.data_len, "Shape data incorrect"
                self.data_len += f.len
            else:
                self.dedup_consts_len += 1
                assert f.mem_idx < self.num_consts

    @staticmethod
    def empty(num_consts):
        return Shape([], num_consts)

    @staticmethod
    def full(length):
        """fragmented.rs:101-118"""
        return Shape([Fragment(0, length, DATA, 0)], 0)

    def __len__(self):
        if not self.fragments:
            return 0
        f = self.fragments[-1]
        return f.start + f.len

    def __eq__(self, o):
        # derive(PartialEq) compares fragments and the three counters (the OnceLock caches compare equal when both unset;
        # the reference's KAT compares a freshly made Shape with a computed one whose caches are unset too)
        return (self.fragments == o.fragments and self.data_len == o.data_len and self.num_consts == o.num_consts
                and self.dedup_consts_len == o.dedup_consts_len)

    def add(self, frag):
        """fragmented.rs:121-166 (merge_in + add)"""
        if self.fragments and should_merge(self.fragments[-1], frag):
            prev = self.fragments[-1]
            assert not (prev.content == CONSTS and frag.content == DATA)
            prev.len += frag.len
            if prev.content == DATA:
                self.data_len += frag.len
            return
        if frag.content == DATA:
            assert frag.mem_idx == self.data_len
            self.data_len += frag.len
        else:
            assert frag.mem_idx < self.num_consts
            self.dedup_consts_len += 1
        self.fragments.append(Fragment(*frag.key()))

    def assert_correct(self):
        data_len = dedup = 0
        for f in self.fragments:
            if f.content == DATA:
                assert f.mem_idx == data_len
                data_len += f.len
            else:
                dedup += 1
                assert f.mem_idx < self.num_consts
        assert data_len == self.data_len and dedup == self.dedup_consts_len

    def prune_consts(self):
        """fragmented.rs:351-364"""
        hits, perm = {}, []
        for f in self.fragments:
            if f.content == CONSTS:
                if f.mem_idx not in hits:
                    perm.append(f.mem_idx)
                    hits[f.mem_idx] = len(perm) - 1
                f.mem_idx = hits[f.mem_idx]
        return perm

    def full_split(self):
        """fragmented.rs:285-349 -> (shape of both halves, permutation of the constants)"""
        if self._split is None:
            l = Shape.empty(self.num_consts)
            for frag in self.fragments:
                length, content, start, mem_idx = frag.len, frag.content, frag.start, frag.mem_idx
                if start % 2 == 1:
                    if content == DATA:
                        length += 1
                        start -= 1
                    else:
                        length -= 1
                        start += 1
                        l.add(Fragment(l.data_len, 1, DATA, (start - 2) // 2))
                if length % 2 == 1:
                    length -= 1
                if length > 0:
                    if content == DATA:
                        l.add(Fragment(l.data_len, length // 2, DATA, start // 2))
                    elif length // 2 < MERGE_THRESH:
                        l.add(Fragment(l.data_len, length // 2, DATA, start // 2))
                    else:
                        l.add(Fragment(mem_idx, length // 2, CONSTS, start // 2))
            self._split_perm = l.prune_consts()
            l.assert_correct()
            self._split = l
        return self._split, self._split_perm

    def split(self):
        return self.full_split()[0]


class FragmentedPoly:
    def __init__(self, data, consts, shape, mod=None):
        for f in shape.fragments:
            if f.content == CONSTS:
                assert f.mem_idx < len(consts)
        self.data, self.consts, self.shape, self.mod = list(data), list(consts), shape, mod

    def __len__(self):
        return len(self.shape)

    def num_vars(self):
        n = len(self)
        assert n and n & (n - 1) == 0
        return n.bit_length() - 1

    def get_by_fragment(self, frag, idx):
        return self.data[frag.mem_idx + idx] if frag.content == DATA else self.consts[frag.mem_idx]

    def into_vec(self):
        out = []
        for f in self.shape.fragments:
            for i in range(f.len):
                out.append(self.get_by_fragment(f, i))
        return out

    def split(self):
        """fragmented.rs:676-732: walks source and target fragments in step exactly as the reference does"""
        source = self.shape
        target, perm = source.full_split()
        new_consts = [self.consts[i] for i in perm]
        l = FragmentedPoly([], new_consts, target, self.mod)
        r = FragmentedPoly([], list(new_consts), target, self.mod)
        src = iter(source.fragments)
        sf = next(src, None)
        cnt = 0
        for tf in target.fragments:
            if tf.content == DATA:
                for _ in range(tf.len):
                    l.data.append(self.get_by_fragment(sf, cnt))
                    cnt += 1
                    if cnt >= sf.len:
                        sf = next(src, None)
                        cnt = 0
                    r.data.append(self.get_by_fragment(sf, cnt))
                    cnt += 1
                    if cnt >= sf.len:
                        sf = next(src, None)
                        cnt = 0
            else:
                cnt += tf.len * 2
                if cnt >= sf.len:
                    sf = next(src, None)
                    cnt = 0
        return l, r

    def bind_from(self, r, f):
        """fragmented.rs:736-741: l += f * (r - l) over data then consts"""
        m = self.mod

        def op(a, b):
            v = a + f * (b - a)
            return v % m if m else v
        self.data = [op(a, b) for a, b in zip(self.data, r.data)]
        self.consts = [op(a, b) for a, b in zip(self.consts, r.consts)]

    def bind(self, f):
        l, r = self.split()
        l.bind_from(r, f)
        return l

    def evaluate(self, pt):
        """fragmented.rs:748-761: binds the LSB with the LAST coordinate first"""
        assert self.num_vars() == len(pt)
        cur = self
        for f in reversed(pt):
            cur = cur.bind(f)
        return cur.get_by_fragment(cur.shape.fragments[0], 0)

    def split_at(self, idx):
        """fragmented.rs:526-674: split by the idx-th variable from the top; shapes of <= 2 fragments (Data [, Consts])"""
        source = self.shape
        n = len(self)
        chunk_len = n >> (1 + idx)
        assert 0 < len(source.fragments) <= 2
        assert source.data_len % chunk_len == 0 and (source.data_len // chunk_len) % 2 == 0
        const_idx, merge_consts = None, False
        if len(source.fragments) == 1:
            assert source.fragments[0].content == DATA
            target = Shape([Fragment(0, source.fragments[0].len >> 1, DATA, 0)], 0)
        else:
            assert source.fragments[0].content == DATA and source.fragments[1].content == CONSTS
            m_len = source.fragments[0].len
            chunk_count = m_len // chunk_len
            split_len = len(source) >> 1
            split_data = (chunk_count // 2) * chunk_len
            split_consts = split_len - split_data
            const_idx = source.fragments[1].mem_idx
            if split_consts <= 1:
                split_data += split_consts
                split_consts = 0
                merge_consts = True
            frags = [Fragment(0, split_data, DATA, 0)]
            if split_consts == 0:
                target = Shape(frags, 0)
            else:
                frags.append(Fragment(0, split_consts, CONSTS, split_data))
                target = Shape(frags, 1)
        halves = ([], [])
        t = 0
        for c in range(0, len(self.data), chunk_len):
            halves[t].extend(self.data[c:c + chunk_len])
            t = 1 - t
        l_data, r_data = halves
        if const_idx is None:
            return FragmentedPoly(l_data, [], target, self.mod), FragmentedPoly(r_data, [], target, self.mod)
        c = self.consts[const_idx]
        if merge_consts:
            l_data = l_data + [c] * (target.data_len - len(l_data))
            r_data = r_data + [c] * (target.data_len - len(r_data))
        return FragmentedPoly(l_data, [c], target, self.mod), FragmentedPoly(r_data, [c], target, self.mod)
