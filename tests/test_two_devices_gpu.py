"""One host thread driving TWO GPUs (gm_set_device between calls): the persistent round kernel's device-side state (relay, round and
merge counters, residency word, limb accumulators) and its pinned staging are kept per (thread, device) -- a launch on device B must
not run on device A's counters (advisor finding, round 2).  Needs two visible GPUs; skipped on the one-GPU boxes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_one_thread_proves_on_two_devices():
    from gkr_msm_amd import codec, ffi, harness as H
    from pyref import field as F
    L = ffi.lib()
    x_log, d_log, nbits = 9, 4, 32
    y_size = nbits // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    pts = codec.points_to_mont(F.random_points(n, 5))
    sc = codec.ints_to_limbs(F.random_scalars(n, nbits, 6))
    rng = F.SplitMix64(9)
    r = [rng.next_fr() for _ in range(y_log)]
    tape = [rng.next_bits(128) for _ in range(4000)]
    results = []
    for dev in (0, 1, 0, 1):
        ffi.check(L.gm_set_device(dev))
        torch.cuda.set_device(dev)
        with torch.cuda.device(dev):
            d_pts, d_sc = H.to_dev(pts), H.to_dev(sc)
            plan = H.MsmPlan(x_log, d_log, y_size)
            plan.run(d_pts, d_sc)
            w = H.PipWitness(plan, d_pts, y_log)
            outs, _ = w.outputs()

            def ev(poly):
                cur = list(poly)
                for f in reversed(r):
                    cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % F.P for i in range(len(cur) // 2)]
                return cur[0]
            g = w.prove_image_part(r, [ev(o) for o in outs], tape)
            results.append((g["msgs"], g["evs"], g["rounds"]))
            w.close()
            plan.close()
    ffi.check(L.gm_set_device(0))
    torch.cuda.set_device(0)
    assert all(res == results[0] for res in results[1:]), "the proofs made on the two devices differ"
