"""TEST INFRASTRUCTURE ONLY: ctypes view of oracle/liboracle.so (the C restatement).
Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")


class OrFn(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("prim", C.c_int32 * 4), ("count", C.c_int32 * 4)]


def make_fn(*segs):
    f = OrFn()
    f.nseg = len(segs)
    for i, (p, c) in enumerate(segs):
        f.prim[i] = p
        f.count[i] = c
    return f


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.or_msm.restype = C.c_int
        _lib.or_msm.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_int] + [C.c_void_p] * 7
        _lib.or_msm_combine.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.or_fr_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        _lib.or_fn_exec.argtypes = [C.POINTER(OrFn), C.c_void_p, C.c_void_p]
        _lib.or_fn_n_ins.argtypes = [C.POINTER(OrFn)]
        _lib.or_fn_n_outs.argtypes = [C.POINTER(OrFn)]
        _lib.or_dense_bind.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        _lib.or_dense_bind21.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        _lib.or_dense_make21.argtypes = [C.c_void_p, C.c_uint64]
        _lib.or_eq_table.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.or_coeff_d.argtypes = [C.c_void_p]
    return _lib


def msm(points_mont, scalars, x_log, d_log, y_size, y0=0, y1=None, threads=1, want_aux=True):
    """points_mont (N,8) u64, scalars (N,4) u64 canonical -> dict of numpy outputs (Montgomery limbs)"""
    L = lib()
    y1 = y_size if y1 is None else y1
    n = 1 << x_log
    nwin = y1 - y0
    nrows = nwin << d_log
    pts = np.ascontiguousarray(points_mont, dtype=np.uint64)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = {
        "digits": np.zeros((nwin, n), dtype=np.uint16) if want_aux else None,
        "counter": np.zeros((nwin, n), dtype=np.uint32) if want_aux else None,
        "row_len": np.zeros(nrows, dtype=np.uint32),
        "bx": np.zeros((nrows, 4), dtype=np.uint64),
        "by": np.zeros((nrows, 4), dtype=np.uint64),
        "bz": np.zeros((nrows, 4), dtype=np.uint64),
        "window_cols": np.zeros((3 * (d_log + 1), nwin, 4), dtype=np.uint64),
    }

    def p(a):
        return a.ctypes.data if a is not None else None
    rc = L.or_msm(pts.ctypes.data, sc.ctypes.data, x_log, d_log, y_size, y0, y1, threads, p(out["digits"]),
                  p(out["counter"]), p(out["row_len"]), p(out["bx"]), p(out["by"]), p(out["bz"]),
                  p(out["window_cols"]))
    if rc != 0:
        raise ValueError("or_msm rejected the shape (rc=%d)" % rc)
    return out


def msm_combine(window_cols, d_log):
    L = lib()
    w = np.ascontiguousarray(window_cols, dtype=np.uint64)
    out = np.zeros((2, 4), dtype=np.uint64)
    L.or_msm_combine(w.ctypes.data, d_log, w.shape[1], out.ctypes.data)
    return out


class PipWitness:
    """or_pip_witness: CPU witness + image-part prover of the C oracle"""

    def __init__(self, points_mont, scalars, x_log, d_log, y_size, y_log, threads=1):
        L = lib()
        L.or_pip_witness_create.restype = C.c_void_p
        L.or_pip_witness_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.or_pip_witness_destroy.argtypes = [C.c_void_p]
        L.or_pip_witness_output.argtypes = [C.c_void_p, C.c_void_p]
        L.or_pip_prove_image_part.restype = C.c_int
        L.or_pip_prove_image_part.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                              C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int]
        self.L, self.d_log, self.y_log, self.threads = L, d_log, y_log, threads
        pts = np.ascontiguousarray(points_mont, dtype=np.uint64)
        sc = np.ascontiguousarray(scalars, dtype=np.uint64)
        self.h = L.or_pip_witness_create(pts.ctypes.data, sc.ctypes.data, x_log, d_log, y_size, y_log, threads)
        if not self.h:
            raise ValueError("or_pip_witness_create rejected the shape")

    def close(self):
        if self.h:
            self.L.or_pip_witness_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def output(self):
        out = np.zeros((3 * (self.d_log + 1), 1 << self.y_log, 4), dtype=np.uint64)
        self.L.or_pip_witness_output(self.h, out.ctypes.data)
        return out

    def prove_image_part(self, claim_point_mont, claim_evs_mont, tape_limbs, msgs_cap=1 << 16):
        cp = np.ascontiguousarray(claim_point_mont, dtype=np.uint64)
        ce = np.ascontiguousarray(claim_evs_mont, dtype=np.uint64)
        tp = np.ascontiguousarray(tape_limbs, dtype=np.uint64)
        msgs = np.zeros((msgs_cap, 4), dtype=np.uint64)
        fpt = np.zeros((80, 4), dtype=np.uint64)
        fev = np.zeros((3, 4), dtype=np.uint64)
        nm, used, rounds = C.c_uint64(), C.c_uint64(), C.c_uint64()
        npt = C.c_uint32()
        rc = self.L.or_pip_prove_image_part(self.h, cp.ctypes.data, ce.ctypes.data, tp.ctypes.data, tp.shape[0],
                                            msgs.ctypes.data, msgs_cap, C.addressof(nm), fpt.ctypes.data,
                                            C.addressof(npt), fev.ctypes.data, C.addressof(used), C.addressof(rounds),
                                            self.threads)
        if rc != 0:
            raise ValueError("or_pip_prove_image_part failed (rc=%d)" % rc)
        return dict(msgs=msgs[: nm.value], point=fpt[: npt.value], evs=fev, tape_used=used.value, rounds=rounds.value)


def gkr_msm_prove(points_mont, bits_u8, lp, lb, tape_limbs, threads=1, msgs_cap=1 << 18):
    """C oracle gen-1 prover (or_gkr_msm_prove); returns Montgomery limb arrays"""
    L = lib()
    L.or_gkr_msm_prove.restype = C.c_int
    L.or_gkr_msm_prove.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                   C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int]
    pts = np.ascontiguousarray(points_mont, dtype=np.uint64)
    bits = np.ascontiguousarray(bits_u8, dtype=np.uint8)
    tp = np.ascontiguousarray(tape_limbs, dtype=np.uint64)
    msgs = np.zeros((msgs_cap, 4), dtype=np.uint64)
    nout = 1 << lb
    outp = np.zeros((3 * nout, 4), dtype=np.uint64)
    fpt = np.zeros((80, 4), dtype=np.uint64)
    fev = np.zeros((3, 4), dtype=np.uint64)
    nm, used, rounds = C.c_uint64(), C.c_uint64(), C.c_uint64()
    npt = C.c_uint32()
    rc = L.or_gkr_msm_prove(pts.ctypes.data, bits.ctypes.data, lp, lb, tp.ctypes.data, tp.shape[0], msgs.ctypes.data,
                            msgs_cap, C.addressof(nm), outp.ctypes.data, fpt.ctypes.data, C.addressof(npt),
                            fev.ctypes.data, C.addressof(used), C.addressof(rounds), threads)
    if rc != 0:
        raise ValueError("or_gkr_msm_prove failed (rc=%d)" % rc)
    return dict(msgs=msgs[: nm.value], output=outp, point=fpt[: npt.value], evs=fev, tape_used=used.value,
                rounds=rounds.value)


# ------------------------------------------------------------------ G1 side (gkrmsm_oracle_g1.c)
def g1_msm_wnaf_nonaff(bases_jac_limbs, scalars_limbs, threads=1):
    """msm_bigint_wnaf_nonaff over (n, 18) uint64 Jacobian bases and (n, 4) canonical scalars -> (12,) affine limbs"""
    L = lib()
    b = np.ascontiguousarray(bases_jac_limbs, dtype=np.uint64)
    s = np.ascontiguousarray(scalars_limbs, dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    rc = L.or_g1_msm_wnaf_nonaff(C.c_void_p(b.ctypes.data), C.c_void_p(s.ctypes.data), C.c_uint64(len(s)), C.c_int(threads),
                                 C.c_void_p(out.ctypes.data))
    assert rc == 0
    return out


def g1_msm_affine(bases_aff_limbs, scalars_limbs, threads=1):
    L = lib()
    b = np.ascontiguousarray(bases_aff_limbs, dtype=np.uint64)
    s = np.ascontiguousarray(scalars_limbs, dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    rc = L.or_g1_msm_affine(C.c_void_p(b.ctypes.data), C.c_void_p(s.ctypes.data), C.c_uint64(len(s)), C.c_int(threads),
                            C.c_void_p(out.ctypes.data))
    assert rc == 0
    return out


def g1_binary_msm(coefs_u8, tables_aff_limbs, gamma):
    L = lib()
    c = np.ascontiguousarray(coefs_u8, dtype=np.uint8)
    t = np.ascontiguousarray(tables_aff_limbs, dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    rc = L.or_g1_binary_msm(C.c_void_p(c.ctypes.data), C.c_void_p(t.ctypes.data), C.c_uint64(len(c)), C.c_uint(gamma),
                            C.c_void_p(out.ctypes.data))
    assert rc == 0
    return out


def g1_pushforward_outer(digits_u16, counter_u32, basis_aff_limbs, x_log, d_log, y_size, clm, threads=1):
    """-> dict(d_outer (n_mat*2^d, 18), c_outer (n_mat*c_stride, 18), c_stride, d_comm (n_mat, 12), c_comm (n_mat, 12))"""
    L = lib()
    dg = np.ascontiguousarray(digits_u16, dtype=np.uint16)
    ct = np.ascontiguousarray(counter_u32, dtype=np.uint32)
    bs = np.ascontiguousarray(basis_aff_limbs, dtype=np.uint64)
    n_mat = (y_size + (1 << clm) - 1) >> clm
    cmax = int(ct.max()) + 1
    d_outer = np.zeros((n_mat << d_log, 18), dtype=np.uint64)
    c_outer = np.zeros((n_mat * cmax, 18), dtype=np.uint64)
    d_comm = np.zeros((n_mat, 12), dtype=np.uint64)
    c_comm = np.zeros((n_mat, 12), dtype=np.uint64)
    stride = C.c_uint32()
    rc = L.or_g1_pushforward_outer(C.c_void_p(dg.ctypes.data), C.c_void_p(ct.ctypes.data), C.c_void_p(bs.ctypes.data),
                                   C.c_uint(x_log), C.c_uint(d_log), C.c_uint(y_size), C.c_uint(clm), C.c_int(threads),
                                   C.c_void_p(d_outer.ctypes.data), C.c_void_p(c_outer.ctypes.data), C.c_uint64(n_mat * cmax),
                                   C.byref(stride), C.c_void_p(d_comm.ctypes.data), C.c_void_p(c_comm.ctypes.data))
    assert rc == 0 and stride.value == cmax
    return dict(d_outer=d_outer, c_outer=c_outer, c_stride=cmax, d_comm=d_comm, c_comm=c_comm)


def g1_to_affine(jac_limbs):
    L = lib()
    j = np.ascontiguousarray(jac_limbs, dtype=np.uint64).reshape(-1, 18)
    out = np.zeros((len(j), 12), dtype=np.uint64)
    L.or_g1_to_affine(C.c_void_p(j.ctypes.data), C.c_uint64(len(j)), C.c_void_p(out.ctypes.data))
    return out
