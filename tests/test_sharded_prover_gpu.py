"""GPU test of the sharded "prove image part" (SURVEY 8e): world_size-2 (and 4) processes share the one GPU of the box and
exchange through gloo or through the library's shared-memory communicator (gm_comm_shm_*); every rank owns half (a quarter) of
the MSM windows = bucket rows.  The sharded run must give the same prover messages and final claims as the unsharded run, bit for
bit, on every rank -- and it must take the same device path: pre-enqueued rounds and the persistent stage kernel (the ranks agree
on the latter per layer; GM_STAGE_FORCE_NONRESIDENT on ONE rank makes everybody fall back)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, x_log, d_log, nbits, q, transport="gloo", sabotage_rank=-1):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if rank == sabotage_rank:   # this rank's stage launches give up at their residency barrier (read once per process)
            os.environ["GM_STAGE_FORCE_NONRESIDENT"] = "1"
        dist = None
        if transport == "gloo":
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import ctypes as C
        from gkr_msm_amd import codec, dist as gd, ffi, harness as H
        from pyref import field as F
        y_size = (nbits + d_log - 1) // d_log
        y_log = (y_size - 1).bit_length()
        n = 1 << x_log
        d_pts = H.to_dev(codec.points_to_mont(F.random_points(n, 5)))
        sc = F.random_scalars(n, nbits, 6)
        sc[0] = 0
        d_sc = H.to_dev(codec.ints_to_limbs(sc))
        rng = F.SplitMix64(9)
        r = [rng.next_fr() for _ in range(y_log)]
        tape = [rng.next_bits(128) for _ in range(4000)]
        # unsharded reference (every rank computes it: small)
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
        a0, b0 = C.c_uint64(), C.c_uint64()
        ffi.lib().gm_sc_stage_counts(C.byref(a0), C.byref(b0))
        # sharded: this rank's windows only
        y0, y1 = gd.window_range(rank, world, y_size)
        comm = gd.Comm(dist, rank, world) if transport == "gloo" else gd.ShmComm("/gm-test-%d-%d" % (port, world), rank, world)
        plan_s = H.MsmPlan(x_log, d_log, y_size, y0, y1)
        plan_s.run(d_pts, d_sc)
        ws = H.PipWitness(plan_s, d_pts, y_log, comm=comm)
        outs_s, bs_s = ws.outputs()
        got = ws.prove_image_part(r, evs, tape)
        ok = (outs_s == outs and bs_s == bs and got["msgs"] == ref["msgs"] and got["point"] == ref["point"] and
              got["evs"] == ref["evs"] and got["rounds"] == ref["rounds"] and got["tape_used"] == ref["tape_used"])
        a1, b1 = C.c_uint64(), C.c_uint64()
        ffi.lib().gm_sc_stage_counts(C.byref(a1), C.byref(b1))
        q.put((rank, ok, comm.calls, got["rounds"], (a0.value, a1.value - a0.value, b1.value - b0.value)))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:  # report instead of hanging the parent
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc(), 0, (0, 0, 0)))


def _run(world, x_log, d_log, nbits, transport, sabotage_rank=-1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world + (7 if transport == "shm" else 0) + (13 if sabotage_rank >= 0 else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, x_log, d_log, nbits, q, transport, sabotage_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=240))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert len(res) == world
    for rank, ok, calls, rounds, stage in sorted(res):
        assert ok is True, "rank %d: %s" % (rank, calls)
        assert calls > rounds // 2      # the round sums really went through the collective
    return sorted(res)


@pytest.mark.parametrize("transport", ["gloo", "shm"])
@pytest.mark.parametrize("world,x_log,d_log,nbits", [(2, 5, 3, 12), (2, 7, 4, 32), (4, 6, 2, 16), (2, 4, 2, 4)])
def test_sharded_prove_image_part_matches_unsharded(world, x_log, d_log, nbits, transport):
    res = _run(world, x_log, d_log, nbits, transport)
    for rank, ok, calls, rounds, (unsharded, sharded, left) in res:
        # the sharded proof takes the stage kernel wherever the unsharded one does (bintree layers; the triangle layers are
        # replicated and small), and no launch is left early
        assert sharded >= 1 and left == 0, "rank %d: %d stage launches unsharded, %d sharded, %d left early" % (rank, unsharded, sharded, left)


def _worker_q(rank, world, q, port, x_log, d_log, nbits):
    _worker(rank, world, port, x_log, d_log, nbits, q, "shm")


def test_sharded_prove_image_part_world_8():
    """the target machine's rank count: 8 ranks x 2 windows (4 processes x 2 rank threads share the GPU, tests/rank_threads.py): the
    last log2(8) = 3 rounds of every dense stage run replicated, the leader proves the bucket reduction for seven followers, the stage
    kernel's agreement word is exchanged between 8"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from rank_threads import run_ranks
    res = run_ranks(_worker_q, 8, (35100 + os.getpid() % 2000, 6, 2, 32), threads_per_proc=2)
    for rank, ok, calls, rounds, (unsharded, sharded, left) in res:
        assert calls > rounds // 2


def test_sharded_ranks_agree_when_one_cannot_run_the_stage_kernel():
    """rank 1's stage launches never become resident (forced): every rank must leave its launch of that layer and run ordinary rounds
    -- same messages as the unsharded proof, no hang"""
    res = _run(2, 7, 4, 32, "shm", sabotage_rank=1)
    for rank, ok, calls, rounds, (unsharded, sharded, left) in res:
        assert left >= 1, "rank %d: %d launches, none left early" % (rank, sharded)
