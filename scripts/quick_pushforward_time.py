"""Quick timing of the pushforward prover at a given shape (development aid)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
plan = H.MsmPlan(x_log, d_log, y_size)
plan.run(d_pts, H.to_dev(sc))
P = codec.P
r = [int.from_bytes(rng.bytes(64), "little") % P for _ in range(y_log + d_log + x_log)]
evs = [int.from_bytes(rng.bytes(64), "little") % P for _ in range(3)]   # any claim triple: the prover folds, it does not check
tape = [int.from_bytes(rng.bytes(64), "little") % P for _ in range(4)] + [int.from_bytes(rng.bytes(16), "little") for _ in range(1500)]
for it in range(3):
    torch.cuda.synchronize()
    t = time.time()
    try:
        res = H.pushforward_prove(plan, d_pts, y_log, r, evs, tape)
        print("pushforward %.1f ms: %d rounds, %d msgs" % ((time.time() - t) * 1e3, res["rounds"], len(res["msgs"])), flush=True)
    except Exception as e:
        print("pushforward failed after %.1f ms: %s" % ((time.time() - t) * 1e3, e), flush=True)
