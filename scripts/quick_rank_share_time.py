"""One rank's share of the window-sharded MSM on ONE GPU (development aid): MsmPlan over windows [y0, y1) of y_size, timed
unpipelined (run + read-back of the window points, as bench.py's N > 1 step without the gather) and with the stage breakdown.
usage: quick_rank_share_time.py x_log d_log nbits world [rank]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import dist as gdist, ffi, harness

x_log, d_log, nbits, world = (int(v) for v in sys.argv[1:5])
rank = int(sys.argv[5]) if len(sys.argv) > 5 else 0
L = ffi.lib()
n = 1 << x_log
y_size = (nbits + d_log - 1) // d_log
y0, y1 = gdist.window_range(rank, world, y_size)
d_pts = harness.dev_empty(n * 8)
ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 0x474b524d534d, harness.cur_stream()))
sc = np.random.default_rng(1).integers(0, 2**63, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 60) - 1)
d_sc = harness.to_dev(sc)
plan = harness.MsmPlan(x_log, d_log, y_size, y0, y1)
for _ in range(3):
    plan.run(d_pts, d_sc); plan.window_points_raw()
steps = 20
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    plan.run(d_pts, d_sc)
    raw = plan.window_points_raw()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print("rank %d of %d: windows [%d, %d) x_logsize %d: %.3f ms per step (run + window-point read-back), %.1f M points/s x %d ranks = %.1f M" % (
    rank, world, y0, y1, x_log, dt * 1e3, n / dt / 1e6, world, n / dt / 1e6))
prof = (C.c_float * 7)()
ffi.check(L.gm_msm_profile(plan.h, 2))
plan.run(d_pts, d_sc); plan.window_points_raw()
ffi.check(L.gm_msm_profile_read(plan.h, prof, 7))
print(dict(zip(["digits", "histogram", "chunk_scan_offsets", "scatter", "add_level0", "add_levels_ge1", "triangle"], [round(float(v), 4) for v in prof])))

# pipelined over `depth` plans / streams, as bench.py's timed loop (without the all-gather)
for depth in (2, 4, 6, 8):
    plans = [plan] + [harness.MsmPlan(x_log, d_log, y_size, y0, y1) for _ in range(depth - 1)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    for pl, st in zip(plans, streams):
        ffi.check(L.gm_msm_profile(pl.h, 0))
        with torch.cuda.stream(st):
            pl.run(d_pts, d_sc); pl.window_points_raw()
    torch.cuda.synchronize()

    def finish(j):
        with torch.cuda.stream(streams[j % depth]):
            return harness.combine_host(plans[j % depth].window_points_raw(), d_log)
    steps = 40
    t = time.perf_counter()
    for j in range(steps):
        with torch.cuda.stream(streams[j % depth]):
            plans[j % depth].run(d_pts, d_sc)
        if j >= depth - 1:
            res = finish(j - depth + 1)
    for j in range(steps - depth + 1, steps):
        res = finish(j)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print("pipelined depth %d: %.3f ms per step incl. host combine -> %.1f M points/s" % (depth, dt * 1e3, n / dt / 1e6))
    for pl in plans[1:]:
        pl.close()
