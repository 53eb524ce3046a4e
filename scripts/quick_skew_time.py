"""MSM step + image-part proof at config B's shape with skewed scalars (every point in ONE bucket per window, half of them, uniform):
does a degenerate bucket population cost more than the uniform one?  python scripts/quick_skew_time.py [x_logsize] [kinds,comma,separated] [msm|full]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from gkr_msm_amd import ffi, harness as H

P = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
x_log = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
msm_only = len(sys.argv) > 3 and sys.argv[3] == "msm"
full_mode = len(sys.argv) > 3 and sys.argv[3] == "full"     # the whole gen-2 proof (PippengerWG::new + prove under merlin), second call
d_basis = d_inv = None
d_log, nbits = 8, 256
y_size = nbits // d_log
y_log = 5
n = 1 << x_log
d_pts = H.dev_empty(n * 8)
ffi.check(ffi.lib().gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, 77, H.cur_stream()))
if full_mode:
    from gkr_msm_amd import codec
    tau = int.from_bytes(np.random.default_rng(23).bytes(32), "little") % P
    d_basis = H.g1_mock_srs(tau, (2 << x_log) - 1, codec.G1_GEN)
    H.g1_fixed_base_register(d_basis, (2 << x_log) - 1)
    d_inv = H.knuckles_setup(2, x_log)
rng = np.random.default_rng(3)
uni = rng.integers(0, 2 ** 64, size=(n, 4), dtype=np.uint64)
uni[:, 3] &= np.uint64((1 << 60) - 1)
same = np.tile(np.array([0x9B1B00FF5A3C7E01, 0x1122334455667788, 0x0F0E0D0C0B0A0908, 0x0102030405060708], dtype=np.uint64), (n, 1))
half = uni.copy()
half[::2] = same[::2]
for name, sc in (("uniform", uni), ("half_in_one_bucket", half), ("all_same", same), ("zero", np.zeros((n, 4), dtype=np.uint64))):
    if only and name not in only:
        continue
    d_sc = H.to_dev(sc)
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        plan.run(d_pts, d_sc)
    torch.cuda.synchronize()
    msm_ms = (time.perf_counter() - t0) / 5 * 1e3
    if msm_only:
        print("%-20s MSM step %.2f ms (unpipelined)" % (name, msm_ms), flush=True)
        plan.close()
        continue
    if full_mode:
        res = []
        for it in range(2):
            plan.run(d_pts, d_sc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            wg = H.PippengerWG(plan, d_pts, y_log, 0, d_basis)
            torch.cuda.synchronize()
            t_w = time.perf_counter() - t0
            outs = wg.dense_output()
            pr = np.random.default_rng(1)
            r_pt = [int.from_bytes(pr.bytes(64), "little") % P for _ in range(y_log)]

            def ev(poly):
                cur = list(poly)
                for f in reversed(r_pt):
                    cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
                return cur[0]
            r_evs = [ev(o) for o in outs]
            mt = H.MerlinTranscript(b"skew")
            fm = H.pippenger_prove_tr(wg, r_pt, r_evs, d_inv, 2, mt)
            mt.close()
            res = (t_w, fm["call_s"], H.pippenger_last_spans())
            wg.close()
        print("%-20s MSM step %.2f ms, WG::new %.1f ms, prove %.1f ms, spans %s" % (name, msm_ms, res[0] * 1e3, res[1] * 1e3, res[2]), flush=True)
        plan.close()
        del d_sc
        ffi.lib().gm_release_cached_memory()
        continue
    w = H.PipWitness(plan, d_pts, y_log)
    outs, _ = w.outputs()
    pr = np.random.default_rng(1)
    r_pt = [int.from_bytes(pr.bytes(64), "little") % P for _ in range(y_log)]

    def ev(poly):
        cur = list(poly)
        for f in reversed(r_pt):
            cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
        return cur[0]
    r_evs = [ev(o) for o in outs]
    tape = [int.from_bytes(pr.bytes(16), "little") for _ in range(4000)]
    w.prove_image_part(r_pt, r_evs, tape)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        g = w.prove_image_part(r_pt, r_evs, tape)
        best = min(best, time.perf_counter() - t0)
    print("%-20s MSM step %.2f ms (unpipelined), image part %.1f ms, %d rounds" % (name, msm_ms, best * 1e3, g["rounds"]), flush=True)
    w.close()
    plan.close()
    del d_sc
    ffi.lib().gm_release_cached_memory()
