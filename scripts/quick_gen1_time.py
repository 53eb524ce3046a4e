"""Quick timing of the gen-1 prover gkr_msm_prove at a given size (development aid): time, device memory in use."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import codec, ffi, harness as H

lp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = ffi.lib()
d_pts = H.dev_empty((1 << lp) * 8)
ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), 1 << lp, 0x474b524d534d, H.cur_stream()))
rng = np.random.default_rng(11)
d_bits = torch.from_numpy(rng.integers(0, 2, size=(1 << (lp + lb)), dtype=np.uint8)).cuda()
tape = [int.from_bytes(rng.bytes(64), "little") % codec.P for _ in range(6000)]
for it in range(3):
    torch.cuda.synchronize()
    t = time.time()
    r = H.gkr_msm_prove(d_pts, d_bits, lp, lb, tape, msgs_cap=1 << 16)
    dt = time.time() - t
    free, total = torch.cuda.mem_get_info()
    print("gkr_msm_prove 2^%d x 2^%d: %.1f ms (witness %.1f ms), %d rounds; device memory in use (pool incl.) %.1f GiB" % (
        lp, lb, dt * 1e3, r["witness_ms"], r["rounds"], (total - free) / 2**30), flush=True)
