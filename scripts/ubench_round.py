"""Micro-benchmark of the dense round kernels (development aid): DenseDeg2 / generic dense objects over random columns,
time of the first unipoly() (one round kernel over 2^(nv-1) pairs) and of the first bind()."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import codec, ffi, harness as H

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 22
P = codec.P
rng = np.random.default_rng(1)


def rand_cols(k, n):
    cols = []
    for _ in range(k):
        t = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda")
        cols.append(t.reshape(-1))
    return cols


cases = [("aff_l1", ffi.make_fn((1, 1)), 4, 3, 4 + 2), ("aff_l2", ffi.make_fn((2, 1)), 3, 3, 1 + 2), ("aff_l3", ffi.make_fn((3, 1)), 3, 3, 4 + 2),
         ("proj_l1", ffi.make_fn((4, 1)), 6, 4, 5 + 3), ("proj_l2", ffi.make_fn((5, 1)), 4, 4, 4 + 3), ("proj_l3", ffi.make_fn((6, 1)), 4, 3, 4 + 2)]
n = 1 << nv
pt = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(nv)]
for name, fn, k, m, mul_pt in cases:
    cols = rand_cols(k, n)
    for kind in ("deg2", "generic"):
        if kind == "deg2":
            so = H.Sumcheckable.dense_deg2(fn, nv, cols, pt, 7, [1] * m)
            muls = (2 * mul_pt + 2) * (n // 2)
        else:
            eq = rand_cols(1, n)
            so = H.Sumcheckable.dense(0, fn, nv, cols + eq, 7, 1)
            muls = (3 * (mul_pt + 1)) * (n // 2)
        torch.cuda.synchronize()
        t = time.time()
        so.unipoly()
        dt = time.time() - t
        t = time.time()
        so.bind(5)
        torch.cuda.synchronize()
        db = time.time() - t
        so.unipoly(); so.bind(3); so.unipoly(); so.bind(3)
        torch.cuda.synchronize()
        t = time.time()
        so.unipoly()
        dt3 = time.time() - t
        print("%-8s %-7s nv=%d: round %.3f ms (%.1f G mul/s, %.0f GB/s read), bind %.3f ms, round@nv-3 %.3f ms" % (
            name, kind, nv, dt * 1e3, muls / dt / 1e9, (k + (kind == "generic")) * n * 32 / dt / 1e9, db * 1e3, dt3 * 1e3), flush=True)
        so.close()
