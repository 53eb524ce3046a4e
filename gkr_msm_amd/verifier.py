"""Host-side wrappers of the verifier entry points of the C ABI (include/gkrmsm.h, "the verifier"): Pippenger::verify and
KzgVerifyingKey::verify_pair.  No GPU is involved -- the reference's verifier is CPU code and so is the library's.
Plumbing for tests and bench.py; values cross as Python ints (canonical), points as (x, y) tuples or None.
"""
import ctypes as C

import numpy as np

from . import codec, ffi

GM_ERR_VERIFY = 5


class Rejected(Exception):
    """the verifier rejected the proof (GM_ERR_VERIFY); str() names the failed check"""


def _check(rc):
    if rc == GM_ERR_VERIFY:
        raise Rejected(ffi.lib().gm_last_error().decode())
    ffi.check(rc)


def g2_to_limbs(q):
    """((x0, x1), (y0, y1)) or None -> (24,) uint64: x.c0, x.c1, y.c0, y.c1 in Montgomery form"""
    flat = [0, 0, 0, 0] if q is None else [q[0][0], q[0][1], q[1][0], q[1][1]]
    return codec.fq_to_mont_limbs(flat).reshape(-1)


def pippenger_verify(x_log, d_log, y_size, y_log, clm, claim_point, claim_evs, g0, k, scalars, points, tape):
    """gm_pippenger_verify over recorded messages + challenge tape; returns dict(pair=(A, B), tape_used)"""
    L = ffi.lib()
    cp, ce = codec.to_mont_limbs(list(claim_point)), codec.to_mont_limbs(list(claim_evs))
    g0l = codec.g1_aff_to_limbs([g0])
    kk = codec.to_mont_limbs([k])
    sc = codec.to_mont_limbs(list(scalars)) if len(scalars) else np.zeros((1, 4), dtype=np.uint64)
    pts = codec.g1_aff_to_limbs(list(points)) if len(points) else np.zeros((1, 12), dtype=np.uint64)
    tp = codec.ints_to_limbs(list(tape))
    pair = np.zeros(24, dtype=np.uint64)
    used = C.c_uint64()
    _check(L.gm_pippenger_verify(x_log, d_log, y_size, y_log, clm, cp.ctypes.data, ce.ctypes.data, g0l.ctypes.data, kk.ctypes.data,
                                 sc.ctypes.data, len(scalars), pts.ctypes.data, len(points), tp.ctypes.data, len(tape),
                                 pair.ctypes.data, C.byref(used)))
    return dict(pair=tuple(codec.g1_aff_from_limbs(pair)), tape_used=used.value)


def pippenger_verify_merlin(x_log, d_log, y_size, y_log, clm, claim_point, claim_evs, g0, k, pparam, proof):
    """the same through the built-in merlin transcript in verifier mode, over proof bytes; returns (A, B)"""
    L = ffi.lib()
    h = C.c_void_p()
    pb = (C.c_uint8 * max(len(proof), 1)).from_buffer_copy(bytes(proof) or b"\0")
    ffi.check(L.gm_merlin_create_verifier(pparam, len(pparam), pb, len(proof), C.byref(h)))
    try:
        rd = ffi.GmTranscriptReader()
        ffi.check(L.gm_merlin_reader(h, C.byref(rd)))
        cp, ce = codec.to_mont_limbs(list(claim_point)), codec.to_mont_limbs(list(claim_evs))
        g0l = codec.g1_aff_to_limbs([g0])
        kk = codec.to_mont_limbs([k])
        pair = np.zeros(24, dtype=np.uint64)
        _check(L.gm_pippenger_verify_tr(x_log, d_log, y_size, y_log, clm, cp.ctypes.data, ce.ctypes.data, g0l.ctypes.data,
                                        kk.ctypes.data, C.byref(rd), pair.ctypes.data))
        left = C.c_uint64()
        ffi.check(L.gm_merlin_unread(h, C.byref(left)))
        if left.value:
            raise Rejected("proof has %d unread bytes" % left.value)
        return tuple(codec.g1_aff_from_limbs(pair))
    finally:
        L.gm_merlin_destroy(h)


def g2_from_limbs(arr):
    v = codec.fq_from_mont_limbs(np.asarray(arr, dtype=np.uint64).reshape(4, 6))
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def kzg_mock_vk(tau):
    """([1]_2, [tau]_2): the G2 part of KzgProvingKey::mock_setup's verifying key for a known tau"""
    t = codec.to_mont_limbs([tau])
    h0, h1 = np.zeros(24, dtype=np.uint64), np.zeros(24, dtype=np.uint64)
    ffi.check(ffi.lib().gm_kzg_mock_vk(t.ctypes.data, h0.ctypes.data, h1.ctypes.data))
    return g2_from_limbs(h0), g2_from_limbs(h1)


def kzg_verify_pair(pair, h0, h1):
    """KzgVerifyingKey::verify_pair: True iff e(A, h0) == e(B, h1)"""
    pl = codec.g1_aff_to_limbs(list(pair)).reshape(-1)
    a, b = g2_to_limbs(h0), g2_to_limbs(h1)
    try:
        _check(ffi.lib().gm_kzg_verify_pair(pl.ctypes.data, a.ctypes.data, b.ctypes.data))
    except Rejected:
        return False
    return True


def pairing(p, q):
    """e(P, Q) as 12 canonical Fq coordinates in the library's tower order (see gm_pairing)"""
    pl = codec.g1_aff_to_limbs([p]).reshape(-1)
    ql = g2_to_limbs(q)
    out = np.zeros(72, dtype=np.uint64)
    ffi.check(ffi.lib().gm_pairing(pl.ctypes.data, ql.ctypes.data, out.ctypes.data))
    return codec.fq_from_mont_limbs(out.reshape(12, 6))


def gkr_msm_verify(log_num_points, log_num_scalar_bits, msgs, tape):
    """gen-1: gm_gkr_msm_verify over the prover's message stream + challenge tape; returns the final claim"""
    L = ffi.lib()
    ms = codec.to_mont_limbs(list(msgs)) if len(msgs) else np.zeros((1, 4), dtype=np.uint64)
    tp = codec.ints_to_limbs(list(tape))
    fpt = np.zeros((64, 4), dtype=np.uint64)
    fev = np.zeros((8, 4), dtype=np.uint64)
    npt, used, rounds = C.c_uint32(), C.c_uint64(), C.c_uint64()
    _check(L.gm_gkr_msm_verify(log_num_points, log_num_scalar_bits, ms.ctypes.data, len(msgs), tp.ctypes.data, len(tape),
                               fpt.ctypes.data, C.byref(npt), fev.ctypes.data, C.byref(used), C.byref(rounds)))
    return dict(point=codec.from_mont_limbs(fpt[: npt.value]), evs=codec.from_mont_limbs(fev[:3]), tape_used=used.value,
                rounds=rounds.value)
