"""Image-part proofs from T host threads at once (development aid): each thread owns a stream, an MSM plan, a witness, and proves
`reps` times; prints rounds/s of one thread alone and of all threads together.  usage: quick_concurrent_proofs.py x_log d_log nbits T"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import codec, ffi, harness as H
from gkr_msm_amd.codec import P

x_log, d_log, nbits, T = (int(v) for v in sys.argv[1:5])
reps = 4
L = ffi.lib()
n = 1 << x_log
y_size = (nbits + d_log - 1) // d_log
y_log = (y_size - 1).bit_length()
d_pts = H.dev_empty(n * 8)
ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 0x474b524d534d, H.cur_stream()))
sc = np.random.default_rng(1).integers(0, 2**64, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 60) - 1)
d_sc = H.to_dev(sc)
torch.cuda.synchronize()
pr = np.random.default_rng(2)
r = [int.from_bytes(pr.bytes(64), "little") % P for _ in range(y_log)]
tape = [int.from_bytes(pr.bytes(16), "little") for _ in range(4000)]


def ev(poly, pt):
    cur = list(poly)
    for f in reversed(pt):
        cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
    return cur[0]


def worker(k, T, bar, out):
    """one prover thread: its own stream, plan, witness; alive until its proofs are done"""
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        plan = H.MsmPlan(x_log, d_log, y_size)
        plan.run(d_pts, d_sc)
        w = H.PipWitness(plan, d_pts, y_log)
        outs, _ = w.outputs()
        evs = [ev(o, r) for o in outs]
        first = w.prove_image_part(r, evs, tape)   # warm
        out["rounds"] = first["rounds"]
        for phase in range(2):                     # phase 0: thread 0 alone; phase 1: everybody
            bar.wait()
            if phase == 1 or k == 0:
                for _ in range(reps):
                    res = w.prove_image_part(r, evs, tape)
                    assert res["msgs"] == first["msgs"], "thread %d: proof differs" % k
            bar.wait()
        w.close()
        plan.close()


out = {}
bar = threading.Barrier(T + 1)
ths = [threading.Thread(target=worker, args=(k, T, bar, out)) for k in range(T)]
for th in ths:
    th.start()
walls = []
for phase in range(2):
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    walls.append(time.perf_counter() - t0)
for th in ths:
    th.join()
rounds = out["rounds"]
print("1 thread : %d proofs in %.1f ms -> %.0f rounds/s" % (reps, walls[0] * 1e3, reps * rounds / walls[0]))
print("%d threads: %d proofs in %.1f ms -> %.0f rounds/s aggregate, every proof identical to the first" % (
    T, reps * T, walls[1] * 1e3, reps * T * rounds / walls[1]))
