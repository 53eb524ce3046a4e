"""Quick timing of the MSM path at a given shape (development aid, not the bench)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import codec, ffi, harness

x_log = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d_log = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nbits = int(sys.argv[3]) if len(sys.argv) > 3 else 256
L = ffi.lib()
n = 1 << x_log
y_size = (nbits + d_log - 1) // d_log
d_pts = harness.dev_empty(n * 8)
t = time.time()
ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 0x474b524d534d, harness.cur_stream()))
torch.cuda.synchronize()
print("gen_points %.1f ms" % ((time.time() - t) * 1e3))
rng = np.random.default_rng(1)
sc = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 60) - 1)  # < 2^252 < group order
d_sc = harness.to_dev(sc)
plan = harness.MsmPlan(x_log, d_log, y_size)
print("workspace %.2f GiB" % (L.gm_msm_plan_workspace_bytes(plan.h) / 2**30))
for it in range(3):
    torch.cuda.synchronize()
    t = time.time()
    plan.run(d_pts, d_sc)
    torch.cuda.synchronize()
    dt = time.time() - t
    print("msm run %.3f ms  -> %.1f Mpoints/s" % (dt * 1e3, n / dt / 1e6))
res = harness.combine_host(plan.window_points_raw(), d_log)
print("result", hex(res[0])[:20], hex(res[1])[:20])
