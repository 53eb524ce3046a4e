"""Harness plumbing (tests / bench): torch tensors as device memory for the C ABI."""
import ctypes as C
import time

import numpy as np

from . import codec, ffi


def torch_mod():
    import torch
    return torch


def to_dev(arr_u64):
    """numpy uint64 array -> int64 CUDA tensor with the same bits"""
    torch = torch_mod()
    a = np.ascontiguousarray(arr_u64, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


def dev_empty(n_u64):
    torch = torch_mod()
    return torch.empty(int(n_u64), dtype=torch.int64, device="cuda")


def cur_stream():
    torch = torch_mod()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def read_dev(ptr, nbytes, dtype=np.uint64):
    """copy nbytes from a raw device pointer (owned by a plan) into a numpy array"""
    buf = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
    ffi.check(ffi.lib().gm_memcpy_d2h(buf.ctypes.data, ptr, nbytes, cur_stream()))
    return buf


class MsmPlan:
    """RAII wrapper of gm_msm_plan"""

    def __init__(self, x_logsize, d_logsize, y_size, y_begin=0, y_end=None):
        self.L = ffi.lib()
        self.x_logsize, self.d_logsize, self.y_size = x_logsize, d_logsize, y_size
        self.y_begin = y_begin
        self.y_end = y_size if y_end is None else y_end
        self.h = C.c_void_p()
        ffi.check(self.L.gm_msm_plan_create(x_logsize, d_logsize, y_size, self.y_begin, self.y_end, C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.gm_msm_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, d_points, d_scalars):
        ffi.check(self.L.gm_msm_run(self.h, C.c_void_p(d_points.data_ptr()), C.c_void_p(d_scalars.data_ptr()),
                                    cur_stream()))

    @property
    def nwin(self):
        return self.y_end - self.y_begin

    def bucket_sums(self):
        px, py, pz, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        ffi.check(self.L.gm_msm_bucket_sums(self.h, C.byref(px), C.byref(py), C.byref(pz), C.byref(n)))
        return [codec.from_mont_limbs(read_dev(p, n.value * 32)) for p in (px, py, pz)]

    def window_points_raw(self):
        p, nc, cl = C.c_void_p(), C.c_uint64(), C.c_uint64()
        ffi.check(self.L.gm_msm_window_points(self.h, C.byref(p), C.byref(nc), C.byref(cl)))
        return read_dev(p, nc.value * cl.value * 32).reshape(nc.value, cl.value, 4)

    def window_points(self):
        raw = self.window_points_raw()
        return [codec.from_mont_limbs(raw[c]) for c in range(raw.shape[0])]

    def digits_counter_rowlen(self):
        pd, pc, pr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ffi.check(self.L.gm_msm_digits(self.h, C.byref(pd), C.byref(pc), C.byref(pr)))
        n = 1 << self.x_logsize
        dg = read_dev(pd, self.nwin * n * 2, np.uint16).reshape(self.nwin, n)
        ct = read_dev(pc, self.nwin * n * 4, np.uint32).reshape(self.nwin, n)
        rl = read_dev(pr, (self.nwin << self.d_logsize) * 4, np.uint32)
        return dg, ct, rl


def combine_host(raw_cols, d_logsize):
    """raw_cols: (3*(d+1), n_windows, 4) uint64 Montgomery -> affine canonical (x, y)"""
    raw = np.ascontiguousarray(raw_cols, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    ffi.check(ffi.lib().gm_msm_combine_host(raw.ctypes.data, d_logsize, raw.shape[1], out.ctypes.data))
    x, y = codec.from_mont_limbs(out.reshape(2, 4))
    return (x, y)


# ------------------------------------------------------------------------------------------------
# dense columns / VecVec / sumcheck object wrappers (test + bench plumbing over the C ABI)
def ptr_array(tensors):
    """host array of device pointers for `const uint64_t* const*` arguments; keeps nothing alive"""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if hasattr(t, "data_ptr") else t
    return arr


def cols_to_dev(cols):
    """list of lists of canonical ints -> list of device tensors (Montgomery)"""
    return [to_dev(codec.to_mont_limbs(c)) for c in cols]


def cols_to_host(tensors):
    return [codec.from_mont_limbs(to_host(t)) for t in tensors]


def fr_arg(vals):
    """canonical ints -> contiguous (n,4) uint64 Montgomery array (keep a reference while it is in use)"""
    return codec.to_mont_limbs(list(vals))


def dense_map(fn, cols, n_outs):
    n = cols[0].numel() // 4
    outs = [dev_empty(n * 4) for _ in range(n_outs)]
    ffi.check(ffi.lib().gm_dense_map(C.byref(fn), ptr_array(cols), ptr_array(outs), n, cur_stream()))
    return outs


def dense_map_split(fn, cols, n_outs, lo_bit, bundle):
    n = cols[0].numel() // 4
    outs = [dev_empty((n // 2) * 4) for _ in range(2 * n_outs)]
    ffi.check(ffi.lib().gm_dense_map_split(C.byref(fn), ptr_array(cols), ptr_array(outs), n, lo_bit, bundle,
                                           cur_stream()))
    return outs


def dense_bind(cols, t):
    n = cols[0].numel() // 4
    outs = [dev_empty((n // 2) * 4) for _ in cols]
    ta = fr_arg([t])
    ffi.check(ffi.lib().gm_dense_bind(ptr_array(cols), ptr_array(outs), len(cols), n, ta.ctypes.data, cur_stream()))
    return outs


def eq_table(mult, point):
    nv = len(point)
    out = dev_empty((1 << nv) * 4)
    scratch = dev_empty((1 << nv) * 4)
    m, p = fr_arg([mult]), fr_arg(point if nv else [0])
    ffi.check(ffi.lib().gm_eq_table(m.ctypes.data, p.ctypes.data, nv, C.c_void_p(scratch.data_ptr()),
                                    C.c_void_p(out.data_ptr()), cur_stream()))
    return out


class VV:
    """owner of a gm_vv handle"""

    def __init__(self, handle):
        self.h = handle
        self.L = ffi.lib()

    @staticmethod
    def from_host(rows_per_poly, row_pads, col_pads, row_logsize, col_logsize):
        """rows_per_poly[c] = list of rows (lists of canonical ints); all polys share the row lengths"""
        k = len(rows_per_poly)
        nrows = len(rows_per_poly[0])
        lens = np.array([len(r) for r in rows_per_poly[0]], dtype=np.uint32)
        datas = [codec.to_mont_limbs([v for r in rows for v in r]) if sum(len(r) for r in rows) else
                 np.zeros((1, 4), dtype=np.uint64) for rows in rows_per_poly]
        dptr = (C.c_void_p * k)(*[d.ctypes.data for d in datas])
        rp, cp = fr_arg(row_pads), fr_arg(col_pads)
        h = C.c_void_p()
        ffi.check(ffi.lib().gm_vv_from_host(k, nrows, lens.ctypes.data if nrows else None, dptr, rp.ctypes.data,
                                            cp.ctypes.data, row_logsize, col_logsize, C.byref(h), cur_stream()))
        return VV(h)

    @staticmethod
    def from_msm(plan, d_points, y_logsize):
        h = C.c_void_p()
        ffi.check(ffi.lib().gm_vv_from_msm(plan.h, C.c_void_p(d_points.data_ptr()), y_logsize, C.byref(h),
                                           cur_stream()))
        return VV(h)

    def close(self):
        if self.h:
            self.L.gm_vv_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        k, nr, rl, cl = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        tot = C.c_uint64()
        ffi.check(self.L.gm_vv_info(self.h, C.byref(k), C.byref(nr), C.byref(tot), C.byref(rl), C.byref(cl)))
        return dict(k=k.value, nrows=nr.value, total=tot.value, row_logsize=rl.value, col_logsize=cl.value)

    def map(self, fn):
        h = C.c_void_p()
        ffi.check(self.L.gm_vv_map(C.byref(fn), self.h, C.byref(h), cur_stream()))
        return VV(h)

    def map_split(self, fn, bundle):
        h = C.c_void_p()
        ffi.check(self.L.gm_vv_map_split(C.byref(fn), self.h, bundle, C.byref(h), cur_stream()))
        return VV(h)

    def map_split_to_dense(self, fn, bundle, n_outs):
        n = 1 << self.info()["col_logsize"]
        outs = [dev_empty(n * 4) for _ in range(2 * n_outs)]
        ffi.check(self.L.gm_vv_map_split_to_dense(C.byref(fn), self.h, bundle, ptr_array(outs), cur_stream()))
        return outs

    def slice(self, first, count):
        h = C.c_void_p()
        ffi.check(self.L.gm_vv_slice(self.h, first, count, C.byref(h)))
        return VV(h)

    def concat(self, other):
        h = C.c_void_p()
        ffi.check(self.L.gm_vv_concat(self.h, other.h, C.byref(h)))
        return VV(h)

    def to_dense(self):
        inf = self.info()
        n = 1 << (inf["row_logsize"] + inf["col_logsize"])
        outs = [dev_empty(n * 4) for _ in range(inf["k"])]
        ffi.check(self.L.gm_vv_to_dense(self.h, ptr_array(outs), cur_stream()))
        return cols_to_host(outs)

    def rows(self):
        """[(poly c) -> list of stored rows (canonical ints)], plus pads"""
        inf = self.info()
        off = np.zeros(inf["nrows"] + 1, dtype=np.uint32)
        res = []
        for c in range(inf["k"]):
            cells = np.zeros((max(inf["total"], 1), 4), dtype=np.uint64)
            ffi.check(self.L.gm_vv_read(self.h, c, off.ctypes.data, cells.ctypes.data, cur_stream()))
            vals = codec.from_mont_limbs(cells[: inf["total"]]) if inf["total"] else []
            res.append([vals[off[r]:off[r + 1]] for r in range(inf["nrows"])])
        rp = np.zeros((inf["k"], 4), dtype=np.uint64)
        cp = np.zeros((inf["k"], 4), dtype=np.uint64)
        ffi.check(self.L.gm_vv_pads(self.h, rp.ctypes.data, cp.ctypes.data))
        return res, codec.from_mont_limbs(rp), codec.from_mont_limbs(cp)


class Sumcheckable:
    """`Sumcheckable` (vecvec_eq.rs:218-225) over a gm_sc handle; values are canonical ints"""

    def __init__(self, handle, keep=()):
        self.h = handle
        self.L = ffi.lib()
        self.keep = keep  # device tensors / VV the object reads from

    @staticmethod
    def dense_deg2(fn, num_vars, cols, point, gamma, claims):
        h = C.c_void_p()
        p, g, c = fr_arg(point), fr_arg([gamma]), fr_arg(claims)
        ffi.check(ffi.lib().gm_sc_dense_deg2_create(C.byref(fn), num_vars, ptr_array(cols), p.ctypes.data,
                                                    g.ctypes.data, c.ctypes.data, C.byref(h), cur_stream()))
        return Sumcheckable(h, keep=tuple(cols))

    @staticmethod
    def vecvec_deg2(fn, vv, point, gamma, claims):
        h = C.c_void_p()
        p, g, c = fr_arg(point), fr_arg([gamma]), fr_arg(claims)
        ffi.check(ffi.lib().gm_sc_vecvec_deg2_create(C.byref(fn), vv.h, p.ctypes.data, g.ctypes.data, c.ctypes.data,
                                                     C.byref(h), cur_stream()))
        return Sumcheckable(h, keep=(vv,))

    @staticmethod
    def dense(kind, fn, num_vars, cols, gamma, claim):
        h = C.c_void_p()
        g, c = fr_arg([gamma]), fr_arg([claim])
        ffi.check(ffi.lib().gm_sc_dense_create(kind, C.byref(fn) if fn is not None else None, num_vars,
                                               ptr_array(cols), g.ctypes.data, c.ctypes.data, C.byref(h),
                                               cur_stream()))
        return Sumcheckable(h, keep=tuple(cols))

    def close(self):
        if self.h:
            self.L.gm_sc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def unipoly(self):
        buf = np.zeros((8, 4), dtype=np.uint64)
        n = C.c_uint32()
        ffi.check(self.L.gm_sc_unipoly(self.h, buf.ctypes.data, C.byref(n)))
        return codec.from_mont_limbs(buf[: n.value])

    def bind(self, t):
        ta = fr_arg([t])
        ffi.check(self.L.gm_sc_bind(self.h, ta.ctypes.data))

    def final_evals(self):
        buf = np.zeros((64, 4), dtype=np.uint64)
        n = C.c_uint32()
        ffi.check(self.L.gm_sc_final_evals(self.h, buf.ctypes.data, C.byref(n)))
        return codec.from_mont_limbs(buf[: n.value])

    def claim(self):
        buf = np.zeros((1, 4), dtype=np.uint64)
        ffi.check(self.L.gm_sc_claim(self.h, buf.ctypes.data))
        return codec.from_mont_limbs(buf)[0]


class PipWitness:
    """gm_pip_witness: the witness of the image part (bintree + triangle traces) on the device"""

    def __init__(self, plan, d_points, y_logsize, comm=None):
        self.L = ffi.lib()
        self.plan, self.y_logsize, self.comm = plan, y_logsize, comm
        self.h = C.c_void_p()
        if comm is None:
            ffi.check(self.L.gm_pip_witness_create(plan.h, C.c_void_p(d_points.data_ptr()), y_logsize, C.byref(self.h),
                                                   cur_stream()))
        else:  # sharded: plan covers this rank's windows (gkr_msm_amd.dist.Comm)
            ffi.check(self.L.gm_pip_witness_create_sharded(plan.h, C.c_void_p(d_points.data_ptr()), y_logsize,
                                                           C.byref(comm.c), C.byref(self.h), cur_stream()))

    def close(self):
        if self.h:
            self.L.gm_pip_witness_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def outputs(self):
        """(dense output columns, bucket sum columns) as canonical ints"""
        d = self.plan.d_logsize
        ncol = 3 * (d + 1)
        out_ptrs = (C.c_void_p * ncol)()
        bs_ptrs = (C.c_void_p * 3)()
        n, ln = C.c_uint32(), C.c_uint64()
        ffi.check(self.L.gm_pip_witness_outputs(self.h, out_ptrs, C.byref(n), C.byref(ln), bs_ptrs))
        outs = [codec.from_mont_limbs(read_dev(out_ptrs[i], ln.value * 32)) for i in range(n.value)]
        nb = 1 << (self.y_logsize + d)
        bs = [codec.from_mont_limbs(read_dev(bs_ptrs[i], nb * 32)) for i in range(3)]
        return outs, bs

    def prove_image_part(self, claim_point, claim_evs, tape, msgs_cap=1 << 16):
        cp, ce = fr_arg(claim_point), fr_arg(claim_evs)
        tp = codec.ints_to_limbs(tape)
        msgs = np.zeros((msgs_cap, 4), dtype=np.uint64)
        fpt = np.zeros((64, 4), dtype=np.uint64)
        fev = np.zeros((8, 4), dtype=np.uint64)
        nm, used, rounds, npt = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        t0 = time.perf_counter()
        ffi.check(self.L.gm_pip_prove_image_part(self.h, cp.ctypes.data, ce.ctypes.data, tp.ctypes.data, len(tape),
                                                 msgs.ctypes.data, msgs_cap, C.byref(nm), fpt.ctypes.data,
                                                 C.byref(npt), fev.ctypes.data, C.byref(used), C.byref(rounds)))
        call_s = time.perf_counter() - t0   # the library call alone (the conversions below are Python big-int plumbing)
        return dict(msgs=codec.from_mont_limbs(msgs[: nm.value]), point=codec.from_mont_limbs(fpt[: npt.value]),
                    evs=codec.from_mont_limbs(fev[:3]), tape_used=used.value, rounds=rounds.value, call_s=call_s)


def pushforward_prove(plan, d_points, y_logsize, claim_point, claim_evs, tape, msgs_cap=1 << 16, comm=None):
    """gm_pushforward_prove -> dict(msgs, gamma, matrix=(point, evs), ac_c=(point, evs), ac_d=(point, evs), tape_used, rounds);
    comm: gm_comm wrapper (dist.Comm / ShmComm) -> gm_pushforward_prove_sharded on a window-sharded plan"""
    L = ffi.lib()
    x, d = plan.x_logsize, plan.d_logsize
    cp, ce = fr_arg(claim_point), fr_arg(claim_evs)
    tp = codec.ints_to_limbs(tape)
    msgs = np.zeros((msgs_cap, 4), dtype=np.uint64)
    g = np.zeros((1, 4), dtype=np.uint64)
    mp, me = np.zeros((x + y_logsize, 4), dtype=np.uint64), np.zeros((5, 4), dtype=np.uint64)
    cpt, cev = np.zeros((max(x, 1), 4), dtype=np.uint64), np.zeros((2, 4), dtype=np.uint64)
    dpt, dev = np.zeros((max(d, 1), 4), dtype=np.uint64), np.zeros((2, 4), dtype=np.uint64)
    nm, used, rounds = C.c_uint64(), C.c_uint64(), C.c_uint64()
    t0 = time.perf_counter()
    if comm is not None:
        ffi.check(L.gm_pushforward_prove_sharded(plan.h, C.c_void_p(d_points.data_ptr()), y_logsize, C.byref(comm.c), cp.ctypes.data,
                                                 ce.ctypes.data, tp.ctypes.data, len(tape), msgs.ctypes.data, msgs_cap, C.byref(nm),
                                                 g.ctypes.data, mp.ctypes.data, me.ctypes.data, cpt.ctypes.data, cev.ctypes.data,
                                                 dpt.ctypes.data, dev.ctypes.data, C.byref(used), C.byref(rounds), cur_stream()))
    else:
        ffi.check(L.gm_pushforward_prove(plan.h, C.c_void_p(d_points.data_ptr()), y_logsize, cp.ctypes.data, ce.ctypes.data,
                                         tp.ctypes.data, len(tape), msgs.ctypes.data, msgs_cap, C.byref(nm), g.ctypes.data,
                                         mp.ctypes.data, me.ctypes.data, cpt.ctypes.data, cev.ctypes.data, dpt.ctypes.data,
                                         dev.ctypes.data, C.byref(used), C.byref(rounds), cur_stream()))
    call_s = time.perf_counter() - t0
    f = codec.from_mont_limbs
    return dict(msgs=f(msgs[: nm.value]), gamma=f(g)[0], matrix=(f(mp), f(me)), ac_c=(f(cpt[:x]), f(cev)), ac_d=(f(dpt[:d]), f(dev)),
                tape_used=used.value, rounds=rounds.value, call_s=call_s)


def multiopen_prove(cols, nvars, points, evs, tape, msgs_cap=4096):
    """gm_multiopen_prove over device columns (list of tensors)"""
    L = ffi.lib()
    nargs = len(cols)
    pts = fr_arg([c for p in points for c in p])
    ev = fr_arg(evs)
    tp = codec.ints_to_limbs(tape)
    msgs = np.zeros((msgs_cap, 4), dtype=np.uint64)
    op, oe = np.zeros((nvars, 4), dtype=np.uint64), np.zeros((nargs, 4), dtype=np.uint64)
    nm, used, rounds = C.c_uint64(), C.c_uint64(), C.c_uint64()
    ffi.check(L.gm_multiopen_prove(nvars, nargs, ptr_array(cols), pts.ctypes.data, ev.ctypes.data, tp.ctypes.data, len(tape),
                                   msgs.ctypes.data, msgs_cap, C.byref(nm), op.ctypes.data, oe.ctypes.data, C.byref(used),
                                   C.byref(rounds), cur_stream()))
    f = codec.from_mont_limbs
    return dict(msgs=f(msgs[: nm.value]), point=f(op), evs=f(oe), tape_used=used.value, rounds=rounds.value)


def knuckles_setup(k, num_vars):
    d_inv = dev_empty(4 * ((2 << num_vars) - 1))
    ffi.check(ffi.lib().gm_knuckles_setup(fr_arg([k]).ctypes.data, num_vars, _p(d_inv), cur_stream()))
    return d_inv


def knuckles_open(d_basis_aff, d_inverses, k, num_vars, d_poly, poly_len, point, claimed_ev, commitment, tape):
    """-> (proof dict, (A, B)) with G1 points as affine int pairs"""
    kk, pt, ev = fr_arg([k]), fr_arg(point), fr_arg([claimed_ev])
    cm = codec.g1_aff_to_limbs([commitment])
    tp = codec.ints_to_limbs(tape)
    proof = np.zeros(48, dtype=np.uint64)
    pair = np.zeros(24, dtype=np.uint64)
    ffi.check(ffi.lib().gm_knuckles_open(_p(d_basis_aff), _p(d_inverses), kk.ctypes.data, num_vars, _p(d_poly), poly_len, pt.ctypes.data,
                                         ev.ctypes.data, cm.ctypes.data, tp.ctypes.data, len(tape), proof.ctypes.data, pair.ctypes.data,
                                         cur_stream()))
    f, g = codec.from_mont_limbs, codec.g1_aff_from_limbs
    pr = dict(t_comm=g(proof[0:12])[0], t_x=f(proof[12:16])[0], p_x=f(proof[16:20])[0], p_lt_x_proof=g(proof[20:32])[0],
              t_kx=f(proof[32:36])[0], t_kx_proof=g(proof[36:48])[0])
    return pr, tuple(g(pair))


class PippengerWG:
    """gm_pippenger_wg: PippengerWG::new on the device (witness + phase-1 commitments)"""

    def __init__(self, plan, d_points, y_logsize, clm, d_basis_aff):
        self.L = ffi.lib()
        self.plan, self.y_logsize, self.clm = plan, y_logsize, clm
        self.keep = (d_points, d_basis_aff)
        self.h = C.c_void_p()
        ffi.check(self.L.gm_pippenger_wg_create(plan.h, _p(d_points), y_logsize, clm, _p(d_basis_aff), C.byref(self.h), cur_stream()))

    def close(self):
        if self.h:
            self.L.gm_pippenger_wg_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dense_output(self):
        w = C.c_void_p()
        ffi.check(self.L.gm_pippenger_wg_witness(self.h, C.byref(w)))
        d = self.plan.d_logsize
        ncol = 3 * (d + 1)
        out_ptrs = (C.c_void_p * ncol)()
        n, ln = C.c_uint32(), C.c_uint64()
        ffi.check(self.L.gm_pip_witness_outputs(w, out_ptrs, C.byref(n), C.byref(ln), None))
        return [codec.from_mont_limbs(read_dev(out_ptrs[i], ln.value * 32)) for i in range(n.value)]

    def prove(self, claim_point, claim_evs, d_kn_inverses, k, tape, msgs_cap=1 << 16, points_cap=256):
        cp, ce, kk = fr_arg(claim_point), fr_arg(claim_evs), fr_arg([k])
        tp = codec.ints_to_limbs(tape)
        msgs = np.zeros((msgs_cap, 4), dtype=np.uint64)
        pts = np.zeros((points_cap, 12), dtype=np.uint64)
        pair = np.zeros(24, dtype=np.uint64)
        nm, npnt, used, rounds = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        ffi.check(self.L.gm_pippenger_prove(self.h, cp.ctypes.data, ce.ctypes.data, _p(d_kn_inverses), kk.ctypes.data, tp.ctypes.data,
                                            len(tape), msgs.ctypes.data, msgs_cap, C.byref(nm), pts.ctypes.data, points_cap,
                                            C.byref(npnt), pair.ctypes.data, C.byref(used), C.byref(rounds)))
        return dict(msgs=codec.from_mont_limbs(msgs[: nm.value]), points=codec.g1_aff_from_limbs(pts[: npnt.value]),
                    pair=tuple(codec.g1_aff_from_limbs(pair)), tape_used=used.value, rounds=rounds.value)


class KeyView:
    """gm_key_view over device tensors: segments = [(tensor of affine points, index of its first point in kzg_basis(), count)]"""

    def __init__(self, segments):
        self.keep = [t for t, _, _ in segments]
        n = len(segments)
        self.ptrs = (C.c_void_p * n)(*[t.data_ptr() for t, _, _ in segments])
        self.first = (C.c_uint64 * n)(*[f for _, f, _ in segments])
        self.count = (C.c_uint64 * n)(*[c for _, _, c in segments])
        self.c = ffi.GmKeyView(n, 0, self.ptrs, self.first, self.count)

    @staticmethod
    def whole(d_basis_aff, n_points):
        return KeyView([(d_basis_aff, 0, n_points)])

    @staticmethod
    def ranges_for(x_log, d_log, y_log, clm, rank, world):
        """[(first, count)] x 4: what gm_pippenger_sharded_key_ranges says this rank reads"""
        f4, c4 = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
        ffi.check(ffi.lib().gm_pippenger_sharded_key_ranges(x_log, d_log, y_log, clm, rank, world, f4, c4))
        return [(f4[i], c4[i]) for i in range(4)]

    @staticmethod
    def minimal(d_basis_aff, x_log, d_log, y_log, clm, rank, world):
        """only what the rank reads: the four ranges, merged where they touch or overlap, each copied out of the whole key
        (a device tensor, 12 words per point) into an allocation of its own -- what a deployment would upload per rank"""
        rg = sorted((f, f + c) for f, c in KeyView.ranges_for(x_log, d_log, y_log, clm, rank, world) if c)
        merged = []
        for a, b in rg:
            if merged and a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        return KeyView([(d_basis_aff[12 * a: 12 * b].clone(), a, b - a) for a, b in merged])


def knuckles_setup_range(k, num_vars, first, count):
    d_inv = dev_empty(4 * max(count, 1))
    ffi.check(ffi.lib().gm_knuckles_setup_range(fr_arg([k]).ctypes.data, num_vars, first, count, _p(d_inv), cur_stream()))
    return d_inv


def knuckles_slice_of(num_vars, rank, world):
    """(first, count) of the inverses table / key range rank `rank` holds in a sharded opening"""
    n2 = 2 << num_vars
    s = n2 // world
    first = rank * s
    return first, max(0, min(s, n2 - 1 - first))


def knuckles_open_sharded(comm, key, d_inverses_slice, k, num_vars, d_poly_slice, point, claimed_ev, commitment, tape):
    kk, pt, ev = fr_arg([k]), fr_arg(point), fr_arg([claimed_ev])
    cm = codec.g1_aff_to_limbs([commitment])
    tp = codec.ints_to_limbs(tape)
    proof = np.zeros(48, dtype=np.uint64)
    pair = np.zeros(24, dtype=np.uint64)
    ffi.check(ffi.lib().gm_knuckles_open_sharded(C.byref(comm.c), C.byref(key.c), _p(d_inverses_slice), kk.ctypes.data, num_vars,
                                                 _p(d_poly_slice), pt.ctypes.data, ev.ctypes.data, cm.ctypes.data, tp.ctypes.data,
                                                 len(tape), proof.ctypes.data, pair.ctypes.data, cur_stream()))
    f, g = codec.from_mont_limbs, codec.g1_aff_from_limbs
    pr = dict(t_comm=g(proof[0:12])[0], t_x=f(proof[12:16])[0], p_x=f(proof[16:20])[0], p_lt_x_proof=g(proof[20:32])[0],
              t_kx=f(proof[32:36])[0], t_kx_proof=g(proof[36:48])[0])
    return pr, tuple(g(pair))


class PippengerWGSharded(PippengerWG):
    """one rank of gm_pippenger_wg_create_sharded: `plan` covers the rank's windows, `key` (KeyView) the key ranges resident here"""

    def __init__(self, plan, d_points, y_logsize, clm, key, comm):
        self.L = ffi.lib()
        self.plan, self.y_logsize, self.clm = plan, y_logsize, clm
        self.keep = (d_points, key, comm)
        self.h = C.c_void_p()
        ffi.check(self.L.gm_pippenger_wg_create_sharded(plan.h, _p(d_points), y_logsize, clm, C.byref(key.c), C.byref(comm.c),
                                                        C.byref(self.h), cur_stream()))


def pippenger_prove_tr(wg, claim_point, claim_evs, d_kn_inverses, k, transcript):
    """gm_pippenger_prove_tr on a PippengerWG (sharded or not) -> dict(pair, rounds, n_challenges, call_s)"""
    cp, ce, kk = fr_arg(claim_point), fr_arg(claim_evs), fr_arg([k])
    pair = np.zeros(24, dtype=np.uint64)
    used, rounds = C.c_uint64(), C.c_uint64()
    t0 = time.perf_counter()
    ffi.check(wg.L.gm_pippenger_prove_tr(wg.h, cp.ctypes.data, ce.ctypes.data, _p(d_kn_inverses), kk.ctypes.data, C.byref(transcript.c),
                                         pair.ctypes.data, C.byref(used), C.byref(rounds)))
    return dict(pair=tuple(codec.g1_aff_from_limbs(pair)), rounds=rounds.value, n_challenges=used.value, call_s=time.perf_counter() - t0)


class LiveTranscript:
    """A gm_transcript whose callbacks run Python code: `on_write(list of canonical ints)` and `draw() -> int`.
    Stands in for the Rust shim's wrappers over ProofTranscript2 (tests drive it from a tape or a hash)."""

    def __init__(self, draw, on_write=None):
        self.writes, self.n_challenges = [], 0

        def _w(ctx, ptr, n):
            arr = np.ctypeslib.as_array(ptr, shape=(n * 4,)).reshape(n, 4).copy()
            vals = codec.from_mont_limbs(arr)
            self.writes.append(vals)
            if on_write:
                on_write(vals)
            return 0

        self.requests = []   # (n, bitsize) of every challenge request, in order

        def _c(ctx, cnt, bits, out):
            self.requests.append((cnt, bits))
            for j in range(cnt):
                v = draw()
                if v is None:
                    return 7
                self.n_challenges += 1
                v = v % codec.P if bits >= 255 else v & ((1 << bits) - 1)
                for i in range(4):
                    out[4 * j + i] = (v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
            return 0
        self.points = []

        def _pts(ctx, ptr, n):
            arr = np.ctypeslib.as_array(ptr, shape=(n * 12,)).reshape(n, 12).copy()
            self.points.extend(codec.g1_aff_from_limbs(arr))
            return 0
        self._w, self._c, self._pt = ffi.WRITE_SCALARS_CB(_w), ffi.CHALLENGE_CB(_c), ffi.WRITE_SCALARS_CB(_pts)  # keep alive
        self.c = ffi.GmTranscript(None, self._w, self._c, self._pt)


class MerlinTranscript:
    """The library's ProofTranscript2 clone (gm_merlin_*, csrc/merlin.hip) as a gm_transcript: real Fiat-Shamir -- every message is
    absorbed into STROBE-128 / Keccak-f[1600] and every challenge squeezed from it, inside the prover's round loop, with no Python
    on the path (the callbacks are the library's own functions)"""

    def __init__(self, label=b"gkr-msm"):
        self.L = ffi.lib()
        self.h = C.c_void_p()
        ffi.check(self.L.gm_merlin_create(label, len(label), C.byref(self.h)))
        self.c = ffi.GmTranscript()
        ffi.check(self.L.gm_merlin_transcript(self.h, C.byref(self.c)))

    def proof(self):
        pp, pn = C.c_void_p(), C.c_uint64()
        ffi.check(self.L.gm_merlin_proof(self.h, C.byref(pp), C.byref(pn)))
        return C.string_at(pp, pn.value)

    def close(self):
        if self.h:
            self.L.gm_merlin_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prove_image_part_tr(w, claim_point, claim_evs, transcript):
    cp, ce = fr_arg(claim_point), fr_arg(claim_evs)
    fpt = np.zeros((64, 4), dtype=np.uint64)
    fev = np.zeros((8, 4), dtype=np.uint64)
    used, rounds, npt = C.c_uint64(), C.c_uint64(), C.c_uint32()
    t0 = time.perf_counter()
    ffi.check(w.L.gm_pip_prove_image_part_tr(w.h, cp.ctypes.data, ce.ctypes.data, C.byref(transcript.c), fpt.ctypes.data,
                                             C.byref(npt), fev.ctypes.data, C.byref(used), C.byref(rounds)))
    call_s = time.perf_counter() - t0
    return dict(point=codec.from_mont_limbs(fpt[: npt.value]), evs=codec.from_mont_limbs(fev[:3]), n_challenges=used.value,
                rounds=rounds.value, call_s=call_s)


def pushforward_prove_tr(plan, d_points, y_logsize, claim_point, claim_evs, transcript):
    """gm_pushforward_prove_tr: the argument under the caller's live transcript -> dict(rounds, n_challenges, call_s)"""
    L = ffi.lib()
    x, d = plan.x_logsize, plan.d_logsize
    cp, ce = fr_arg(claim_point), fr_arg(claim_evs)
    g = np.zeros((1, 4), dtype=np.uint64)
    mp, me = np.zeros((x + y_logsize, 4), dtype=np.uint64), np.zeros((5, 4), dtype=np.uint64)
    cpt, cev = np.zeros((max(x, 1), 4), dtype=np.uint64), np.zeros((2, 4), dtype=np.uint64)
    dpt, dev = np.zeros((max(d, 1), 4), dtype=np.uint64), np.zeros((2, 4), dtype=np.uint64)
    used, rounds = C.c_uint64(), C.c_uint64()
    t0 = time.perf_counter()
    ffi.check(L.gm_pushforward_prove_tr(plan.h, C.c_void_p(d_points.data_ptr()), y_logsize, cp.ctypes.data, ce.ctypes.data,
                                        C.byref(transcript.c), g.ctypes.data, mp.ctypes.data, me.ctypes.data, cpt.ctypes.data,
                                        cev.ctypes.data, dpt.ctypes.data, dev.ctypes.data, C.byref(used), C.byref(rounds), cur_stream()))
    return dict(rounds=rounds.value, n_challenges=used.value, call_s=time.perf_counter() - t0)


def pippenger_last_spans():
    """wall time of the calling thread's last gm_pippenger_prove(_tr) by the reference's tracing spans (pippenger.rs:121-159), ms"""
    o = (C.c_double * 8)()
    ffi.check(ffi.lib().gm_pippenger_last_spans(o))
    names = ("prove_image_part_ms", "commit_phase_2_ms", "prove_pushforward_ms", "open_witnesses_ms", "open_multiopen_ms", "open_knuckles_ms")
    d = {k: round(o[i], 2) for i, k in enumerate(names)}
    d["open_ms"] = round(o[3] + o[4] + o[5], 2)
    return d


def gkr_msm_prove_tr(d_points, d_bits_u8, log_num_points, log_num_scalar_bits, transcript):
    L = ffi.lib()
    nout = 1 << log_num_scalar_bits
    outp = np.zeros((3 * nout, 4), dtype=np.uint64)
    fpt = np.zeros((64, 4), dtype=np.uint64)
    fev = np.zeros((8, 4), dtype=np.uint64)
    used, rounds, npt = C.c_uint64(), C.c_uint64(), C.c_uint32()
    t0 = time.perf_counter()
    ffi.check(L.gm_gkr_msm_prove_tr(C.c_void_p(d_points.data_ptr()), C.c_void_p(d_bits_u8.data_ptr()), log_num_points,
                                    log_num_scalar_bits, C.byref(transcript.c), outp.ctypes.data, fpt.ctypes.data,
                                    C.byref(npt), fev.ctypes.data, C.byref(used), C.byref(rounds), cur_stream()))
    call_s = time.perf_counter() - t0
    return dict(output=[codec.from_mont_limbs(outp[c * nout:(c + 1) * nout]) for c in range(3)],
                point=codec.from_mont_limbs(fpt[: npt.value]), evs=codec.from_mont_limbs(fev[:3]), n_challenges=used.value,
                rounds=rounds.value, call_s=call_s)


def gkr_msm_prove(d_points, d_bits_u8, log_num_points, log_num_scalar_bits, tape, msgs_cap=1 << 18):
    """gen-1 gkr_msm_prove through the C ABI; d_bits_u8: uint8 CUDA tensor of 2^lp * 2^lb entries"""
    L = ffi.lib()
    tp = codec.ints_to_limbs(tape)
    msgs = np.zeros((msgs_cap, 4), dtype=np.uint64)
    nout = 1 << log_num_scalar_bits
    outp = np.zeros((3 * nout, 4), dtype=np.uint64)
    fpt = np.zeros((64, 4), dtype=np.uint64)
    fev = np.zeros((8, 4), dtype=np.uint64)
    nm, used, rounds, npt, wms = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_double()
    t0 = time.perf_counter()
    ffi.check(L.gm_gkr_msm_prove(C.c_void_p(d_points.data_ptr()), C.c_void_p(d_bits_u8.data_ptr()), log_num_points,
                                 log_num_scalar_bits, tp.ctypes.data, len(tape), msgs.ctypes.data, msgs_cap, C.byref(nm),
                                 outp.ctypes.data, fpt.ctypes.data, C.byref(npt), fev.ctypes.data, C.byref(used),
                                 C.byref(rounds), C.byref(wms), cur_stream()))
    call_s = time.perf_counter() - t0
    return dict(msgs=codec.from_mont_limbs(msgs[: nm.value]), output=[codec.from_mont_limbs(outp[c * nout:(c + 1) * nout])
                                                                        for c in range(3)],
                point=codec.from_mont_limbs(fpt[: npt.value]), evs=codec.from_mont_limbs(fev[:3]), tape_used=used.value,
                rounds=rounds.value, witness_ms=wms.value, call_s=call_s)


# ------------------------------------------------------------------ BLS12-381 G1 (include/gkrmsm.h, G1 section)
def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def g1_aff_dev(points):
    return to_dev(codec.g1_aff_to_limbs(points))


def g1_jac_dev(points, zs=None):
    return to_dev(codec.g1_jac_to_limbs(points, zs))


def g1_read_jac(t):
    return codec.g1_jac_from_limbs(to_host(t))


def g1_read_aff(t):
    return codec.g1_aff_from_limbs(to_host(t))


def _one_aff(buf):
    return codec.g1_aff_from_limbs(buf)[0]


def g1_msm(d_bases_aff, d_scalars, n, mont=False, nbits=255):
    out = np.zeros(12, dtype=np.uint64)
    ffi.check(ffi.lib().gm_g1_msm(_p(d_bases_aff), _p(d_scalars), n, 1 if mont else 0, nbits, out.ctypes.data, cur_stream()))
    return _one_aff(out)


def g1_msm_nonaff(d_bases_jac, d_scalars, n, mont=False, nbits=255):
    out = np.zeros(12, dtype=np.uint64)
    ffi.check(ffi.lib().gm_g1_msm_nonaff(_p(d_bases_jac), _p(d_scalars), n, 1 if mont else 0, nbits, out.ctypes.data,
                                         cur_stream()))
    return _one_aff(out)


def g1_msm_nonaff_grouped(d_bases_jac, stride, ns, d_scalars, mont=False, nbits=255):
    hn = np.asarray(ns, dtype=np.uint32)
    out = np.zeros((len(ns), 12), dtype=np.uint64)
    ffi.check(ffi.lib().gm_g1_msm_nonaff_grouped(_p(d_bases_jac), stride, hn.ctypes.data, len(ns), _p(d_scalars), 1 if mont else 0, nbits,
                                                 out.ctypes.data, cur_stream()))
    return codec.g1_aff_from_limbs(out)


def g1_bucket_sums(d_bases_aff, mapping, n_buckets):
    torch = torch_mod()
    d_map = torch.from_numpy(np.asarray(mapping, dtype=np.uint32).view(np.int32)).cuda()
    out = dev_empty(18 * n_buckets)
    ffi.check(ffi.lib().gm_g1_bucket_sums(_p(d_bases_aff), _p(d_map), len(mapping), n_buckets, _p(out), cur_stream()))
    return out


def g1_pullback_msm(d_bases_aff, mapping, image):
    torch = torch_mod()
    d_map = torch.from_numpy(np.asarray(mapping, dtype=np.uint32).view(np.int32)).cuda()
    d_img = to_dev(codec.to_mont_limbs(image))
    out = np.zeros(12, dtype=np.uint64)
    ffi.check(ffi.lib().gm_g1_pullback_msm(_p(d_bases_aff), _p(d_map), len(mapping), _p(d_img), len(image), out.ctypes.data,
                                           cur_stream()))
    return _one_aff(out)


def g1_weighted_sum(d_buckets_jac, n_groups, group_len):
    out = np.zeros((n_groups, 12), dtype=np.uint64)
    ffi.check(ffi.lib().gm_g1_weighted_sum(_p(d_buckets_jac), n_groups, group_len, out.ctypes.data, cur_stream()))
    return codec.g1_aff_from_limbs(out)


def g1_prepare_bases(d_bases_aff, n, gamma):
    nchunks = (n + gamma - 1) // gamma
    out = dev_empty(12 * nchunks * ((1 << gamma) - 1))
    ffi.check(ffi.lib().gm_g1_prepare_bases(_p(d_bases_aff), n, gamma, _p(out), cur_stream()))
    return out


def g1_binary_msm(coefs_u8, d_tables, gamma):
    torch = torch_mod()
    d_c = torch.from_numpy(np.asarray(coefs_u8, dtype=np.uint8)).cuda()
    out = np.zeros(12, dtype=np.uint64)
    ffi.check(ffi.lib().gm_g1_binary_msm(_p(d_c), _p(d_tables), len(coefs_u8), gamma, out.ctypes.data, cur_stream()))
    return _one_aff(out)


def g1_batch(op, d_a, d_b, n, out_words):
    out = dev_empty(out_words * n)
    ffi.check(ffi.lib().gm_g1_batch(op, _p(d_a), _p(d_b), _p(out), n, cur_stream()))
    return out


def g1_mock_srs(tau, n, g0):
    """tau^i * g0 for i < n (KzgProvingKey::mock_setup) as a device tensor of affine points"""
    out = dev_empty(12 * n)
    t, g = fr_arg([tau]), codec.g1_aff_to_limbs([g0])
    ffi.check(ffi.lib().gm_g1_mock_srs(t.ctypes.data, g.ctypes.data, n, _p(out), cur_stream()))
    return out


def g1_gen_points(n, seed):
    out = dev_empty(12 * n)
    ffi.check(ffi.lib().gm_g1_gen_points(_p(out), n, seed, cur_stream()))
    return out


def msm_g1_outer(plan, d_basis_aff, clm, c_cap_per_mat):
    """(d_outer tensor, c_outer tensor, c_stride, d_comm, c_comm) of PushForwardState::new's G1 part"""
    n_mat = (plan.nwin + (1 << clm) - 1) >> clm
    nd = 1 << plan.d_logsize
    d_d = dev_empty(18 * n_mat * nd)
    d_c = dev_empty(18 * n_mat * c_cap_per_mat)
    stride = C.c_uint32()
    hd = np.zeros((n_mat, 12), dtype=np.uint64)
    hc = np.zeros((n_mat, 12), dtype=np.uint64)
    ffi.check(ffi.lib().gm_msm_g1_outer(plan.h, _p(d_basis_aff), clm, _p(d_d), _p(d_c), n_mat * c_cap_per_mat, C.byref(stride),
                                        hd.ctypes.data, hc.ctypes.data, cur_stream()))
    return d_d, d_c, stride.value, codec.g1_aff_from_limbs(hd), codec.g1_aff_from_limbs(hc)


def msm_g1_outer_part(plan, d_basis_local, slot_of_slice, clm, c_cap_per_mat):
    """one rank's part of the outer buckets (gm_msm_g1_outer_part): returns dict(first_matrix, n_matrices, c_stride, d_part, c_part)
    with d_part / c_part = numpy (n_matrices, 18) uint64 Jacobian points, the rank's share of d_comm / c_comm"""
    cm = 1 << clm
    m0, m1 = plan.y_begin >> clm, (plan.y_end - 1) >> clm
    n_loc = m1 - m0 + 1
    nd = 1 << plan.d_logsize
    d_d = dev_empty(18 * n_loc * nd)
    d_c = dev_empty(18 * n_loc * c_cap_per_mat)
    slots = np.full(cm, -1, dtype=np.int32)
    for s_, v in slot_of_slice.items():
        slots[s_] = v
    stride, fm, nm = C.c_uint32(), C.c_uint32(), C.c_uint32()
    hd = np.zeros((n_loc, 18), dtype=np.uint64)
    hc = np.zeros((n_loc, 18), dtype=np.uint64)
    ffi.check(ffi.lib().gm_msm_g1_outer_part(plan.h, _p(d_basis_local), slots.ctypes.data, clm, _p(d_d), _p(d_c), n_loc * c_cap_per_mat,
                                             C.byref(stride), C.byref(fm), C.byref(nm), hd.ctypes.data, hc.ctypes.data, cur_stream()))
    return dict(first_matrix=fm.value, n_matrices=nm.value, c_stride=stride.value, d_part=hd, c_part=hc, d_outer=d_d, c_outer=d_c)


def g1_combine_parts(comm, parts_jac, n_total, first):
    """cross-rank EC combine (gm_g1_combine_parts): parts_jac (k, 18) placed at slots first .. first + k - 1 of n_total, the other slots
    the point at infinity; returns the n_total affine sums (the same on every rank)"""
    buf = np.zeros((n_total, 18), dtype=np.uint64)
    buf[first:first + len(parts_jac)] = parts_jac
    out = np.zeros((n_total, 12), dtype=np.uint64)
    ffi.check(ffi.lib().gm_g1_combine_parts(C.byref(comm.c), buf.ctypes.data, n_total, out.ctypes.data))
    return codec.g1_aff_from_limbs(out)


def g1_fixed_base_register(d_bases_aff, n):
    """precompute the window multiples of a base array (a proving key); later g1_msm calls on the same tensor take the fixed-base path"""
    ffi.check(ffi.lib().gm_g1_fixed_base_register(_p(d_bases_aff), n, cur_stream()))


def g1_fixed_base_release(d_bases_aff):
    ffi.check(ffi.lib().gm_g1_fixed_base_release(_p(d_bases_aff)))


# ------------------------------------------------------------------ gen-1 fragmented polynomials (include/gkrmsm.h, fragmented section)
def _frag_array(frags):
    """[(mem_idx, len, content, start)] with content 'Data' / 'Consts' -> ctypes array of gm_fragment"""
    arr = (ffi.GmFragment * max(len(frags), 1))()
    for i, (m, ln, c, s) in enumerate(frags):
        arr[i] = ffi.GmFragment(m, ln, s, 0 if c == "Data" else 1, 0)
    return arr


def frag_shape_full_split(frags, num_consts):
    """Shape::full_split -> (fragments of the halves, perm, data_len)"""
    cap = 2 * len(frags) + 4
    out = (ffi.GmFragment * cap)()
    perm = np.zeros(max(num_consts, 1), dtype=np.uint64)
    n_out, n_perm, dl = C.c_uint32(), C.c_uint32(), C.c_uint64()
    ffi.check(ffi.lib().gm_frag_shape_full_split(_frag_array(frags), len(frags), num_consts, out, cap, C.byref(n_out),
                                                 perm.ctypes.data_as(ffi.u64p), len(perm), C.byref(n_perm), C.byref(dl)))
    res = [(int(f.mem_idx), int(f.len), "Data" if f.content == 0 else "Consts", int(f.start)) for f in out[: n_out.value]]
    return res, [int(v) for v in perm[: n_perm.value]], dl.value


def segment_split(start, end):
    st = np.zeros(130, dtype=np.uint64)
    ll = np.zeros(130, dtype=np.uint8)
    n = C.c_uint32()
    ffi.check(ffi.lib().gm_segment_split(start, end, st.ctypes.data_as(ffi.u64p), ll.ctypes.data_as(C.POINTER(C.c_uint8)), 130,
                                         C.byref(n)))
    return [(int(st[i]), int(ll[i])) for i in range(n.value)]


class FragPoly:
    """FragmentedPoly with data / consts on the device (canonical ints in and out)"""

    def __init__(self, frags, num_consts, d_data, d_consts):
        self.frags, self.num_consts, self.d_data, self.d_consts = list(frags), num_consts, d_data, d_consts

    @staticmethod
    def from_host(frags, data, consts):
        z = np.zeros((1, 4), dtype=np.uint64)
        return FragPoly(frags, len(consts), to_dev(codec.to_mont_limbs(data) if data else z),
                        to_dev(codec.to_mont_limbs(consts) if consts else z))

    def _args(self):
        return _frag_array(self.frags), len(self.frags), self.num_consts, _p(self.d_data), _p(self.d_consts)

    def data(self):
        n = sum(f[1] for f in self.frags if f[2] == "Data")
        return codec.from_mont_limbs(to_host(self.d_data).reshape(-1, 4)[:n]) if n else []

    def consts(self):
        return codec.from_mont_limbs(to_host(self.d_consts).reshape(-1, 4)[: self.num_consts]) if self.num_consts else []

    def split(self):
        tf, perm, dl = frag_shape_full_split(self.frags, self.num_consts)
        outs = [dev_empty(4 * max(dl, 1)) for _ in range(2)] + [dev_empty(4 * max(len(perm), 1)) for _ in range(2)]
        ffi.check(ffi.lib().gm_frag_split(*self._args(), _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), cur_stream()))
        return FragPoly(tf, len(perm), outs[0], outs[2]), FragPoly(tf, len(perm), outs[1], outs[3])

    def bind(self, t):
        tf, perm, dl = frag_shape_full_split(self.frags, self.num_consts)
        od, oc = dev_empty(4 * max(dl, 1)), dev_empty(4 * max(len(perm), 1))
        ta = fr_arg([t])
        ffi.check(ffi.lib().gm_frag_bind(*self._args(), ta.ctypes.data, _p(od), _p(oc), cur_stream()))
        return FragPoly(tf, len(perm), od, oc)

    def to_dense(self):
        n = sum(f[1] for f in self.frags)
        out = dev_empty(4 * max(n, 1))
        ffi.check(ffi.lib().gm_frag_to_dense(*self._args(), _p(out), cur_stream()))
        return codec.from_mont_limbs(to_host(out).reshape(-1, 4)[:n])


def frag_eq_materialize(frags, num_consts, multiplier, point):
    """EqPoly::materialize_eq_with_shape -> (values, sums)"""
    dl = sum(f[1] for f in frags if f[2] == "Data")
    d_vals = dev_empty(4 * max(dl, 1))
    sums = np.zeros((max(num_consts, 1), 4), dtype=np.uint64)
    m, p = fr_arg([multiplier]), fr_arg(point if point else [0])
    ffi.check(ffi.lib().gm_frag_eq_materialize(_frag_array(frags), len(frags), num_consts, m.ctypes.data, p.ctypes.data, len(point),
                                               _p(d_vals), sums.ctypes.data, cur_stream()))
    vals = codec.from_mont_limbs(to_host(d_vals).reshape(-1, 4)[:dl]) if dl else []
    return vals, codec.from_mont_limbs(sums[:num_consts]) if num_consts else []


# ------------------------------------------------------------------ the two GKR circuits on their own
class GkrWitness:
    """gm_gkr_witness: TriangleAddWG / VecVecBintreeAddWG + their SimpleGKR prover"""

    def __init__(self, handle, keep):
        self.h, self.keep, self.L = handle, keep, ffi.lib()

    @staticmethod
    def triangle(cols, num_vars, split_hi):
        h = C.c_void_p()
        ffi.check(ffi.lib().gm_triangle_witness_create(ptr_array(cols), num_vars, split_hi, C.byref(h), cur_stream()))
        return GkrWitness(h, tuple(cols))

    @staticmethod
    def bintree(vv, num_adds, do_bitcheck=False):
        h = C.c_void_p()
        ffi.check(ffi.lib().gm_bintree_witness_create(vv.h, num_adds, 1 if do_bitcheck else 0, C.byref(h), cur_stream()))
        return GkrWitness(h, (vv,))

    def close(self):
        if self.h:
            self.L.gm_gkr_witness_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def output(self):
        """(last_step columns as canonical ints, num_vars)"""
        n, nv = C.c_uint32(), C.c_uint32()
        ffi.check(self.L.gm_gkr_witness_output(self.h, None, 0, C.byref(n), C.byref(nv)))
        ptrs = (C.c_void_p * n.value)()
        ffi.check(self.L.gm_gkr_witness_output(self.h, ptrs, n.value, C.byref(n), C.byref(nv)))
        return [codec.from_mont_limbs(read_dev(ptrs[i], 32 << nv.value)) for i in range(n.value)], nv.value

    def prove(self, claim_point, claim_evs, tape, msgs_cap=1 << 16):
        cp, ce = fr_arg(claim_point if claim_point else [0]), fr_arg(claim_evs)
        tp = codec.ints_to_limbs(tape)
        msgs = np.zeros((msgs_cap, 4), dtype=np.uint64)
        fpt, fev = np.zeros((64, 4), dtype=np.uint64), np.zeros((64, 4), dtype=np.uint64)
        nm, used, rounds, npt, nev = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        ffi.check(self.L.gm_gkr_prove(self.h, cp.ctypes.data, ce.ctypes.data, tp.ctypes.data, len(tape), msgs.ctypes.data, msgs_cap,
                                      C.byref(nm), fpt.ctypes.data, C.byref(npt), fev.ctypes.data, C.byref(nev), C.byref(used),
                                      C.byref(rounds)))
        return dict(msgs=codec.from_mont_limbs(msgs[: nm.value]), point=codec.from_mont_limbs(fpt[: npt.value]),
                    evs=codec.from_mont_limbs(fev[: nev.value]), tape_used=used.value, rounds=rounds.value)


def sc_profile(mode):
    ffi.check(ffi.lib().gm_sc_profile(mode))


def sc_profile_read():
    """-> (rows [dict], other_round_bytes, fold_bytes) of the sumcheck round kernels since the last read"""
    rows = (ffi.GmScProfileRow * 64)()
    n, orb, fb = C.c_uint32(), C.c_double(), C.c_double()
    ffi.check(ffi.lib().gm_sc_profile_read(rows, 64, C.byref(n), C.byref(orb), C.byref(fb), cur_stream()))
    out = [dict(kernel=r.kernel.decode(), launches=r.launches, k_cols=r.k_cols, total_ms=r.total_ms, max_ms=r.max_ms, pairs=r.pairs,
                alg_bytes=r.alg_bytes, fr_mul=r.fr_mul, max_ms_pairs=r.max_ms_pairs) for r in rows[: n.value]]
    return out, orb.value, fb.value
