"""Quick timing of the G1 calls (development aid): MSM over 2^k affine bases, pushforward outer buckets at a given shape."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import codec, ffi, harness as H

L = ffi.lib()
logs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,20").split(",")]
outer = sys.argv[2] if len(sys.argv) > 2 else "20,8,256,0"
rng = np.random.default_rng(1)
for lg in logs:
    n = 1 << lg
    d_b = H.g1_gen_points(n, 3)
    sc = rng.integers(0, 2 ** 64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 62) - 1)
    d_sc = H.to_dev(sc)
    torch.cuda.synchronize()
    for it in range(3):
        t = time.time()
        r = H.g1_msm(d_b, d_sc, n)
        dt = time.time() - t
        print("g1_msm 2^%d: %.2f ms -> %.2f M points/s  (x=%s...)" % (lg, dt * 1e3, n / dt / 1e6, hex(r[0])[:14]), flush=True)
if outer != "none":
    x_log, d_log, nbits, clm = [int(v) for v in outer.split(",")]
    n = 1 << x_log
    y_size = (nbits + d_log - 1) // d_log
    d_pts = H.dev_empty(n * 8)
    ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 0x474b524d534d, H.cur_stream()))
    sc = rng.integers(0, 2 ** 64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(sc))
    basis = H.g1_gen_points(n << clm, 9)
    torch.cuda.synchronize()
    for it in range(2):
        t = time.time()
        g_d, g_c, stride, dc, cc = H.msm_g1_outer(plan, basis, clm, n >> 3)
        dt = time.time() - t
        print("msm_g1_outer x=%d d=%d y=%d clm=%d: %.1f ms (%d G1 adds -> %.1f M adds/s), c_stride %d" % (
            x_log, d_log, y_size, clm, dt * 1e3, 2 * n * y_size, 2 * n * y_size / dt / 1e6, stride), flush=True)
