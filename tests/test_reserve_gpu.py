"""gm_reserve (verdict r03, weak 9: a prover that proves ONCE pays the driver's allocation rate for its whole trace): memory taken from
the driver at set-up time serves the first proof -- no driver allocation during it, same proof as without the reserve; gm_unreserve
refuses while handles still hold blocks of the reserve.  Runs in a process of its own: the pool is process-wide."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from gkr_msm_amd import codec, ffi, harness as H
from pyref import field as F
L = ffi.lib()
def stats():
    o = (C.c_uint64 * 8)()
    ffi.check(L.gm_memory_stats(o))
    return dict(driver_allocs=o[0], driver_bytes=o[1], idle=o[2], reserved=o[3], reserved_used=o[4], cuts=o[5])
x_log, d_log, nbits = 12, 4, 32
y_size = (nbits + d_log - 1) // d_log
y_log = (y_size - 1).bit_length()
n = 1 << x_log
d_pts = H.to_dev(codec.points_to_mont(F.random_points(n, 5)))
d_sc = H.to_dev(codec.ints_to_limbs(F.random_scalars(n, nbits, 6)))
rng = F.SplitMix64(3)
r = [rng.next_fr() for _ in range(y_log)]
tape = [rng.next_bits(128) for _ in range(3000)]
def proof():
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    w = H.PipWitness(plan, d_pts, y_log)
    outs, _ = w.outputs()
    def ev(poly):
        cur = list(poly)
        for f in reversed(r):
            cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) %% F.P for i in range(len(cur) // 2)]
        return cur[0]
    g = w.prove_image_part(r, [ev(o) for o in outs], tape)
    return plan, w, g
ffi.check(L.gm_reserve(1 << 30))
s0 = stats()
assert s0["reserved"] == 1 << 30 and s0["reserved_used"] == 0
plan, w, g1 = proof()              # the FIRST proof of the process: everything comes out of the reserve
s1 = stats()
assert s1["driver_allocs"] == s0["driver_allocs"], "the first proof went to the driver %%d times" %% (s1["driver_allocs"] - s0["driver_allocs"])
assert s1["cuts"] > 50 and s1["reserved_used"] > 0
rc = L.gm_unreserve()
assert rc != 0, "gm_unreserve succeeded while handles hold blocks of the reserve"
w.close(); plan.close()
ffi.check(L.gm_release_cached_memory())
s2 = stats()
assert s2["reserved_used"] == 0, "blocks did not return to the reserve: %%r" %% s2
plan, w, g2 = proof()              # after the release: cut from the reserve again (coalesced free ranges)
assert g2["msgs"] == g1["msgs"] and g2["evs"] == g1["evs"]
assert stats()["driver_allocs"] == s0["driver_allocs"]
w.close(); plan.close()
ffi.check(L.gm_unreserve())
assert stats()["reserved"] == 0
plan, w, g3 = proof()              # and without a reserve: the same proof, from driver blocks
assert g3["msgs"] == g1["msgs"]
assert stats()["driver_allocs"] > s0["driver_allocs"]
print("reserve ok: %%d blocks cut, %%d rounds" %% (s1["cuts"], g1["rounds"]))
''' % (ROOT, os.path.join(ROOT, "oracle"))


def test_reserve_serves_the_first_proof_without_driver_allocations():
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "reserve ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
