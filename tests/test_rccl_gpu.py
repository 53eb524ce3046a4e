"""The library's own RCCL backing of the multi-GPU seam (csrc/rccl_comm.hip, gm_comm_rccl_*).

  * world 1 (runs on the one-GPU box): RCCL is found and bound at run time, a communicator comes up on the device, the host
    all-gather / device all-gather / broadcast entry points run (trivially: one rank), and a prover created with the RCCL gm_comm
    gives the unsharded messages.
  * world 2 (needs two GPUs; skipped otherwise -- the driver's multi-GPU node runs it): every rank on its own GPU, windows
    sharded, per-round sums and bucket sums exchanged by ncclAllGather inside the library: window points, bucket sums and every
    prover message equal the unsharded run, bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_entry_points_and_prover():
    from gkr_msm_amd import codec, dist as gd, ffi, harness as H
    from pyref import field as F
    rc = gd.RcclComm(None, 0, 1)
    L = ffi.lib()
    # host all-gather through the gm_comm seam (one rank: the sum is the value itself)
    vals = codec.to_mont_limbs([3, 5, F.P - 1])
    before = vals.copy()
    ffi.check(L.gm_comm_sum_fr(C.byref(rc.c), vals.ctypes.data, 3))
    assert np.array_equal(vals, before) and rc.calls == 1
    # device all-gather and broadcast
    src = torch.arange(1000, dtype=torch.int64, device="cuda")
    dst = torch.zeros_like(src)
    rc.all_gather_dev(C.c_void_p(src.data_ptr()), dst, 8000)
    rc.broadcast_dev(src, 0)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    # a window-sharded plan with one rank = the whole MSM; window points through ncclAllGather
    x_log, d_log, nbits = 6, 3, 12
    y_size = nbits // d_log
    n = 1 << x_log
    d_pts = H.to_dev(codec.points_to_mont(F.random_points(n, 5)))
    d_sc = H.to_dev(codec.ints_to_limbs(F.random_scalars(n, nbits, 6)))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, d_sc)
    assert np.array_equal(rc.gather_window_points(plan), plan.window_points_raw())
    rc.close()


def _worker(rank, world, port, x_log, d_log, nbits, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(rank)
        dist.init_process_group("gloo", rank=rank, world_size=world)   # side channel for the ncclUniqueId only
        from gkr_msm_amd import codec, dist as gd, harness as H
        from pyref import field as F
        y_size = (nbits + d_log - 1) // d_log
        y_log = (y_size - 1).bit_length()
        n = 1 << x_log
        rcomm = gd.RcclComm(dist, rank, world)
        # operands exist on rank 0 only and are replicated by ncclBroadcast
        pts = codec.points_to_mont(F.random_points(n, 5))
        sc = F.random_scalars(n, nbits, 6)
        sc[0] = 0
        d_pts = H.to_dev(pts if rank == 0 else np.zeros_like(pts))
        d_sc = H.to_dev(codec.ints_to_limbs(sc) if rank == 0 else np.zeros((n, 4), dtype=np.uint64))
        rcomm.broadcast_dev(d_pts, 0)
        rcomm.broadcast_dev(d_sc, 0)
        torch.cuda.synchronize()
        assert np.array_equal(H.to_host(d_pts).reshape(-1, 8), pts)
        rng = F.SplitMix64(9)
        r = [rng.next_fr() for _ in range(y_log)]
        tape = [rng.next_bits(128) for _ in range(4000)]
        plan = H.MsmPlan(x_log, d_log, y_size)
        plan.run(d_pts, d_sc)
        w = H.PipWitness(plan, d_pts, y_log)
        outs, bs = w.outputs()
        P = codec.P

        def ev(poly):
            cur = list(poly)
            for f in reversed(r):
                cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
            return cur[0]
        evs = [ev(o) for o in outs]
        ref = w.prove_image_part(r, evs, tape)
        y0, y1 = gd.window_range(rank, world, y_size)
        plan_s = H.MsmPlan(x_log, d_log, y_size, y0, y1)
        plan_s.run(d_pts, d_sc)
        wp_ok = np.array_equal(rcomm.gather_window_points(plan_s), plan.window_points_raw())
        ws = H.PipWitness(plan_s, d_pts, y_log, comm=rcomm)
        outs_s, bs_s = ws.outputs()
        got = ws.prove_image_part(r, evs, tape)
        ok = (wp_ok and outs_s == outs and bs_s == bs and got["msgs"] == ref["msgs"] and got["point"] == ref["point"] and
              got["evs"] == ref["evs"] and got["rounds"] == ref["rounds"])
        q.put((rank, ok, rcomm.calls, got["rounds"]))
        dist.barrier()
        ws.close()
        rcomm.close()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc(), 0))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one rank per GPU over RCCL / xGMI)")
@pytest.mark.parametrize("x_log,d_log,nbits", [(7, 4, 32), (10, 8, 64)])
def test_rccl_world2_sharded_msm_and_prover(x_log, d_log, nbits):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, x_log, d_log, nbits, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=300))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert len(res) == world
    for rank, ok, calls, rounds in sorted(res):
        assert ok is True, "rank %d: %s" % (rank, calls)
        assert calls > rounds // 2
