"""CPU tests of the host-side ProofTranscript2 (csrc/merlin.hip): Keccak-f[1600] against SHA3-256 known answers (via hashlib
and the published digest of the empty string), the merlin framing against merlin's own published test vector and against an
independent Python restatement of STROBE-128 / merlin, and the scalar / point / challenge encodings of
cleanup/proof_transcript.rs:33-69."""
import ctypes as C
import hashlib

import numpy as np

from gkr_msm_amd import codec, ffi
from pyref import field as F
from pyref import g1 as G


def _keccak(state):
    buf = (C.c_uint8 * 200)(*state)
    ffi.check(ffi.lib().gm_keccak_f1600(buf))
    return bytes(buf)


def _sha3_256(msg):
    rate = 136
    st = bytearray(200)
    m = bytearray(msg) + b"\x06"
    m += b"\x00" * (-len(m) % rate)
    m[-1] |= 0x80
    for off in range(0, len(m), rate):
        for i in range(rate):
            st[i] ^= m[off + i]
        st = bytearray(_keccak(st))
    return bytes(st[:32])


def test_keccak_permutation_via_sha3():
    assert _sha3_256(b"").hex() == "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a"
    for msg in (b"abc", b"x" * 135, b"y" * 136, bytes(range(256)) * 3):
        assert _sha3_256(msg) == hashlib.sha3_256(msg).digest()


class PyStrobe:
    """independent restatement of merlin/src/strobe.rs, on hashlib-free Python (the permutation comes from the library under test
    only through SHA3-pinned _keccak)"""
    R = 166

    def __init__(self, label):
        self.st = bytearray(200)
        self.st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        self.st[6:18] = b"STROBEv1.0.2"
        self.st = bytearray(_keccak(self.st))
        self.pos = self.pos_begin = self.cur = 0
        self.meta_ad(label, False)

    def run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[self.R + 1] ^= 0x80
        self.st = bytearray(_keccak(self.st))
        self.pos = self.pos_begin = 0

    def absorb(self, data):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == self.R:
                self.run_f()

    def squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.st[self.pos])
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self.run_f()
        return bytes(out)

    def begin_op(self, flags, more):
        if more:
            assert self.cur == flags
            return
        old = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur = flags
        self.absorb(bytes([old, flags]))
        if flags & (4 | 32) and self.pos != 0:
            self.run_f()

    def meta_ad(self, d, more):
        self.begin_op(16 | 2, more)
        self.absorb(d)

    def ad(self, d, more):
        self.begin_op(2, more)
        self.absorb(d)

    def prf(self, n, more):
        self.begin_op(1 | 2 | 4, more)
        return self.squeeze(n)


class PyMerlin:
    def __init__(self, label):
        self.s = PyStrobe(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, msg):
        self.s.meta_ad(label, False)
        self.s.meta_ad(len(msg).to_bytes(4, "little"), True)
        self.s.ad(msg, False)

    def challenge_bytes(self, label, n):
        self.s.meta_ad(label, False)
        self.s.meta_ad(n.to_bytes(4, "little"), True)
        return self.s.prf(n, False)


def _new(label):
    h = C.c_void_p()
    ffi.check(ffi.lib().gm_merlin_create(label, len(label), C.byref(h)))
    return h


def test_merlin_published_vector_and_python_restatement():
    L = ffi.lib()
    h = _new(b"test protocol")
    ffi.check(L.gm_merlin_append_message(h, b"some label", 10, b"some data", 9))
    out = (C.c_uint8 * 32)()
    ffi.check(L.gm_merlin_challenge_bytes(h, b"challenge", 9, out, 32))
    got = bytes(out)
    py = PyMerlin(b"test protocol")
    py.append_message(b"some label", b"some data")
    assert got == py.challenge_bytes(b"challenge", 32)
    # merlin's own test vector (merlin/src/transcript.rs, `equivalence_simple`)
    assert got.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    L.gm_merlin_destroy(h)
    # longer interaction crossing the rate boundary, empty labels as ProofTranscript2 uses them
    h = _new(b"pippenger")
    py = PyMerlin(b"pippenger")
    for i in range(5):
        msg = bytes((7 * i + j) & 255 for j in range(100 + 37 * i))
        ffi.check(L.gm_merlin_append_message(h, None, 0, msg, len(msg)))
        py.append_message(b"", msg)
        out = (C.c_uint8 * 200)()
        ffi.check(L.gm_merlin_challenge_bytes(h, None, 0, out, 16 + 40 * i))
        assert bytes(out)[:16 + 40 * i] == py.challenge_bytes(b"", 16 + 40 * i)
    L.gm_merlin_destroy(h)


def test_proof_transcript_encodings():
    """write_scalars / write_points / challenge through the gm_transcript callbacks vs the Python restatement"""
    L = ffi.lib()
    h = _new(b"enc")
    tr = ffi.GmTranscript()
    ffi.check(L.gm_merlin_transcript(h, C.byref(tr)))
    py = PyMerlin(b"enc")
    rng = F.SplitMix64(3)
    vals = [rng.next_fr() for _ in range(3)] + [0, F.P - 1]
    limbs = codec.to_mont_limbs(vals)
    assert tr.write_scalars(tr.ctx, limbs.ctypes.data_as(C.POINTER(C.c_uint64)), len(vals)) == 0
    raw = b"".join(v.to_bytes(32, "little") for v in vals)
    py.append_message(b"", raw)
    pts = G.random_points(3, 5) + [None]
    pl = codec.g1_aff_to_limbs(pts)
    assert tr.write_points(tr.ctx, pl.ctypes.data_as(C.POINTER(C.c_uint64)), len(pts)) == 0
    enc = b""
    for p in pts:
        if p is None:
            enc += bytes([0xC0]) + bytes(47)
        else:
            b = bytearray(p[0].to_bytes(48, "big"))
            b[0] |= 0x80 | (0x20 if p[1] > (G.Q - 1) // 2 else 0)
            enc += bytes(b)
    py.append_message(b"", enc)
    proof_p, proof_n = C.c_void_p(), C.c_uint64()
    ffi.check(L.gm_merlin_proof(h, C.byref(proof_p), C.byref(proof_n)))
    assert C.string_at(proof_p, proof_n.value) == raw + enc
    # challenge_vec(4, 512) then challenge(128)
    out = (C.c_uint64 * 16)()
    assert tr.challenge(tr.ctx, 4, 512, out) == 0
    b = py.challenge_bytes(b"", 256)
    want = [int.from_bytes(b[64 * i:64 * (i + 1)], "little") % F.P for i in range(4)]
    assert codec.limbs_to_ints(np.array(list(out), dtype=np.uint64)) == want
    assert tr.challenge(tr.ctx, 1, 128, out) == 0
    assert codec.limbs_to_ints(np.array(list(out)[:4], dtype=np.uint64)) == [int.from_bytes(py.challenge_bytes(b"", 16), "little")]
    L.gm_merlin_destroy(h)
