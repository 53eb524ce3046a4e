"""Harness plumbing (tests / bench): torch tensors as device memory for the C ABI."""
import ctypes as C

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
