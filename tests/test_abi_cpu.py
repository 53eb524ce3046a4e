"""CPU tests: the C-ABI library loads, exports every symbol include/gkrmsm.h declares, and its host-side
scalar glue (same source as the device code) agrees with the Python big-int oracle.  No GPU compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gkr_msm_amd import codec, ffi
from pyref import algfn as A
from pyref import field as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gkrmsm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 20
    raw = C.CDLL(ffi.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "libgkrmsm_hip.so does not export %s" % s
    # the ctypes table covers the header and vice versa
    assert sorted(ffi.declared_symbols()) == syms


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(ffi, "_lib", None)
    monkeypatch.setattr(ffi, "LIB_PATH", "/nonexistent/libgkrmsm_hip.so")
    with pytest.raises(ffi.GmError):
        ffi.lib()


def test_error_codes_not_exceptions():
    L = ffi.lib()
    h = C.c_void_p()
    assert L.gm_msm_plan_create(4, 1, 4, 0, 4, C.byref(h)) == 1          # d_logsize < 2
    assert b"d_logsize" in L.gm_last_error()
    assert L.gm_msm_plan_create(4, 8, 33, 0, 33, C.byref(h)) == 1         # y_size*d > 256
    assert L.gm_msm_plan_create(4, 8, 4, 3, 2, C.byref(h)) == 1           # empty window range
    assert L.gm_fr_host(99, None, None, None, 0) == 1


def test_host_field_glue_vs_bigint():
    L = ffi.lib()
    rng = F.SplitMix64(5)
    n = 300
    a = [rng.next_fr() for _ in range(n)]
    b = [rng.next_fr() for _ in range(n)]
    a[:3] = [0, F.P - 1, 1]
    b[:3] = [0, F.P - 1, F.P - 1]
    A_, B_ = codec.to_mont_limbs(a), codec.to_mont_limbs(b)
    o = np.zeros_like(A_)

    def run(op):
        ffi.check(L.gm_fr_host(op, A_.ctypes.data, B_.ctypes.data, o.ctypes.data, n))
        return codec.from_mont_limbs(o)
    assert run(0) == [(x + y) % F.P for x, y in zip(a, b)]
    assert run(1) == [(x - y) % F.P for x, y in zip(a, b)]
    assert run(2) == [(x * y) % F.P for x, y in zip(a, b)]
    assert run(9) == [(x * y) % F.P for x, y in zip(a, b)]   # the 32-bit-limb formulation of the device code
    assert run(3) == [(-x) % F.P for x in a]
    assert run(7) == [F.mul_by_a(x) for x in a]
    assert run(8) == [F.mul_by_d(x) for x in a]
    inv = run(4)
    assert all(x * y % F.P == 1 for x, y in zip(a[1:], inv[1:]))


def test_host_algfn_vs_pyref():
    L = ffi.lib()
    cases = {
        "tri": (ffi.make_fn((7, 1), (4, 3)), A.StackedAlgFn(A.TRI_L1, A.RepeatedAlgFn(A.PROJ_L1, 3))),
        "bc": (ffi.make_fn((1, 1), (9, 2)), A.StackedAlgFn(A.AFF_L1, A.RepeatedAlgFn(A.BitCheckFn(), 2))),
        "l2x4": (ffi.make_fn((5, 4)), A.RepeatedAlgFn(A.PROJ_L2, 4)),
        "l3x4": (ffi.make_fn((6, 4)), A.RepeatedAlgFn(A.PROJ_L3, 4)),
    }
    rng = F.SplitMix64(8)
    for name, (f, pyf) in cases.items():
        ni, no, dg = C.c_int32(), C.c_int32(), C.c_int32()
        ffi.check(L.gm_fn_shape(C.byref(f), C.byref(ni), C.byref(no), C.byref(dg)))
        assert (ni.value, no.value, dg.value) == (pyf.n_ins, pyf.n_outs, pyf.deg)
        rows = [[rng.next_fr() for _ in range(pyf.n_ins)] for _ in range(4)]
        i = codec.to_mont_limbs([v for r in rows for v in r])
        o = np.zeros((4 * pyf.n_outs, 4), dtype=np.uint64)
        ffi.check(L.gm_fn_host(C.byref(f), i.ctypes.data, o.ctypes.data, 4))
        assert codec.from_mont_limbs(o) == [v for r in rows for v in pyf.exec(r)], name


def test_host_combine_vs_oracle():
    import oracle_ffi as O
    from gkr_msm_amd import harness
    x_log, d_log, y_size = 6, 4, 8
    n = 1 << x_log
    pts = F.random_points(n, 3)
    sc = F.random_scalars(n, 32, 4)
    r = O.msm(codec.points_to_mont(pts), codec.ints_to_limbs(sc), x_log, d_log, y_size)
    got = harness.combine_host(r["window_cols"], d_log)
    acc = (0, 1)
    for p, s in zip(pts, sc):
        acc = F.te_add_affine(acc, F.te_mul_affine(p, s))
    assert got == acc


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """argument validation happens before any device work: error codes, not crashes (the reference panics on the same conditions)"""
    L = ffi.lib()
    assert L.gm_g1_msm(None, None, 5, 0, 255, None, None) == 1
    assert L.gm_g1_binary_msm(None, None, 3, 9, None, None) == 1                 # gamma > 8: coefficients are u8
    assert L.gm_g1_prepare_bases(None, 4, 0, None, None) == 1
    assert L.gm_pushforward_prove(None, None, 2, None, None, None, 0, None, 0, None, None, None, None, None, None, None, None, None,
                                  None, None) == 1
    assert L.gm_multiopen_prove(0, 4, None, None, None, None, 0, None, 0, None, None, None, None, None, None) == 1
    assert L.gm_knuckles_setup(None, 3, None, None) == 1
    assert L.gm_knuckles_open(None, None, None, 3, None, 1, None, None, None, None, 0, None, None, None) == 1
    assert L.gm_pippenger_wg_create(None, None, 2, 0, None, None, None) == 1
    assert L.gm_pip_witness_create_sharded(None, None, 2, None, None, None) == 1
    assert L.gm_comm_sum_fr(None, None, 0) == 1
    assert L.gm_release_cached_memory() == 0 and L.gm_g1_release_scratch() == 0   # nothing cached: no device call
    assert b"argument" in L.gm_last_error() or b"null" in L.gm_last_error() or len(L.gm_last_error()) > 0


def test_header_is_plain_c_and_cpp():
    """the drop-in boundary is a C ABI: the header must compile, warning-free, as C11 and as C++17"""
    import subprocess
    hdr = os.path.join(ROOT, "include", "gkrmsm.h")
    for cmd in (["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_verifier_entry_points_validate_arguments():
    import ctypes as C
    from gkr_msm_amd import ffi
    L = ffi.lib()
    z = C.c_void_p(None)
    assert L.gm_pippenger_verify(4, 2, 4, 2, 0, z, z, z, z, z, 0, z, 0, z, 0, z, None) == 1          # GM_ERR_INVALID: null claims
    assert L.gm_pippenger_verify_tr(4, 2, 4, 2, 0, z, z, z, z, None, z) == 1
    assert L.gm_kzg_verify_pair(z, z, z) == 1 and L.gm_gkr_msm_verify_tr(3, 2, None, z, None, z, None) == 1
    buf = (C.c_uint64 * 64)()
    # shape checks come before any read: x_logsize < d_logsize is the reference's assert (pippenger.rs:93)
    tape = (C.c_uint64 * 4)()
    assert L.gm_pippenger_verify(2, 3, 4, 2, 0, buf, buf, buf, buf, buf, 0, buf, 0, tape, 1, buf, None) == 1
    # an empty proof is rejected, not crashed on
    assert L.gm_pippenger_verify(4, 2, 4, 2, 0, buf, buf, buf, buf, buf, 0, buf, 0, tape, 1, buf, None) == 5   # GM_ERR_VERIFY
    assert b"ran out of points" in L.gm_last_error()
    assert L.gm_gkr_msm_verify(3, 2, buf, 0, tape, 1, buf, None, buf, None, None) == 5
