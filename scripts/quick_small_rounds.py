"""Latency of the small sumcheck rounds (development aid): the image-part prover at a tiny x_logsize is nothing but dense-stage
and thin rounds (8192 bucket rows at d_logsize 8, 256-bit scalars), so its time / rounds is the per-round latency."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import codec, ffi, harness as H

x_log = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d_log = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nbits = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
L = ffi.lib()
n = 1 << x_log
y_size = (nbits + d_log - 1) // d_log
y_log = (y_size - 1).bit_length()
d_pts = H.dev_empty(n * 8)
ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 0x474b524d534d, H.cur_stream()))
sc = np.random.default_rng(1).integers(0, 2**64, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 60) - 1)
plan = H.MsmPlan(x_log, d_log, y_size)
plan.run(d_pts, H.to_dev(sc))
w = H.PipWitness(plan, d_pts, y_log)
outs, _ = w.outputs()
P = codec.P
pr = np.random.default_rng(2)
r = [int.from_bytes(pr.bytes(64), "little") % P for _ in range(y_log)]


def ev(poly):
    cur = list(poly)
    for f in reversed(r):
        cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
    return cur[0]


evs = [ev(o) for o in outs]
tape = [int.from_bytes(pr.bytes(16), "little") for _ in range(4000)]
res = w.prove_image_part(r, evs, tape)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    res = w.prove_image_part(r, evs, tape)
dt = (time.perf_counter() - t) / reps
print("x=%d d=%d nbits=%d: %.2f ms per proof, %d rounds -> %.1f us per round" % (x_log, d_log, nbits, dt * 1e3, res["rounds"], dt * 1e6 / res["rounds"]))
