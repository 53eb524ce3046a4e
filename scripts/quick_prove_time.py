"""Quick timing of witness build + image-part prover at a given shape (development aid)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import codec, ffi, harness as H

x_log = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d_log = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nbits = int(sys.argv[3]) if len(sys.argv) > 3 else 256
L = ffi.lib()
n = 1 << x_log
y_size = (nbits + d_log - 1) // d_log
y_log = (y_size - 1).bit_length()
d_pts = H.dev_empty(n * 8)
ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 0x474b524d534d, H.cur_stream()))
rng = np.random.default_rng(1)
sc = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 60) - 1)
d_sc = H.to_dev(sc)
plan = H.MsmPlan(x_log, d_log, y_size)
plan.run(d_pts, d_sc)
torch.cuda.synchronize()
t = time.time()
w = H.PipWitness(plan, d_pts, y_log)
torch.cuda.synchronize()
t_w = time.time() - t
print("witness build %.1f ms, trace %.2f GiB" % (t_w * 1e3, L.gm_pip_witness_bytes(w.h) / 2**30))
# claims: evaluate dense output at a point with the device (small): use host python for y_log vars
outs, _ = w.outputs()
from gkr_msm_amd.codec import P
pr = np.random.default_rng(2)
r = [int.from_bytes(pr.bytes(64), "little") % P for _ in range(y_log)]
def ev(poly, pt):
    cur = list(poly)
    for f in reversed(pt):
        cur = [(cur[2*i] + f * (cur[2*i+1] - cur[2*i])) % P for i in range(len(cur)//2)]
    return cur[0]
evs = [ev(o, r) for o in outs]
tape = [int.from_bytes(pr.bytes(16), "little") for _ in range(4000)]
reps = int(os.environ.get("GM_QUICK_REPS", "2"))
best = 1e9
for it in range(reps):
    torch.cuda.synchronize()
    res = w.prove_image_part(r, evs, tape)
    dt = res["call_s"]
    best = min(best, dt)
    if it < 2 or it == reps - 1:
        print("prove image part %.1f ms: %d rounds, %d challenges, %d msgs -> %.0f rounds/s" % (
            dt * 1e3, res["rounds"], res["tape_used"], len(res["msgs"]), res["rounds"] / dt))
if reps > 2:
    print("best of %d: %.2f ms" % (reps, best * 1e3))
