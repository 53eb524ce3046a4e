"""Quick timing of the whole gen-2 prover (gm_pippenger_wg_create + gm_pippenger_prove) at a given shape (development aid).
The SRS is synthetic (random multiples of the generator), which costs the same as a real one."""
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
clm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
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
nv = x_log + clm
d_basis = H.g1_gen_points((2 << nv) - 1, 7)
if os.environ.get("FIXED_BASE", "1") == "1":
    torch.cuda.synchronize()
    t = time.time()
    H.g1_fixed_base_register(d_basis, (2 << nv) - 1)
    torch.cuda.synchronize()
    print("fixed-base tables %.1f ms" % ((time.time() - t) * 1e3))
d_inv = H.knuckles_setup(2, nv)
P = codec.P
for it in range(int(os.environ.get("ITERS", "2"))):
    torch.cuda.synchronize()
    t = time.time()
    plan.run(d_pts, d_sc)
    torch.cuda.synchronize()
    t_msm = time.time() - t
    t = time.time()
    wg = H.PippengerWG(plan, d_pts, y_log, clm, d_basis)
    torch.cuda.synchronize()
    t_wg = time.time() - t
    outs = wg.dense_output()
    r = [int.from_bytes(rng.bytes(64), "little") % P for _ in range(y_log)]

    def ev(poly):
        cur = list(poly)
        for f in reversed(r):
            cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
        return cur[0]
    evs = [ev(o) for o in outs]
    tape = [int.from_bytes(rng.bytes(16), "little") for _ in range(6000)]
    t = time.time()
    res = wg.prove(r, evs, d_inv, 2, tape)
    t_pr = time.time() - t
    print("bucketing+MSM %.1f ms | PippengerWG::new %.1f ms | Pippenger::prove %.1f ms (%d rounds, %d scalars, %d points)" % (
        t_msm * 1e3, t_wg * 1e3, t_pr * 1e3, res["rounds"], len(res["msgs"]), len(res["points"])), flush=True)
    wg.close()
