"""GPU test of the pushforward argument with the matrix sharded by windows (gm_pushforward_prove_sharded): world_size 2
and 4 processes share the one GPU of the box and exchange through the library's shared-memory communicator; every rank owns a
contiguous block of windows.  The sharded run must give the unsharded argument's messages, gamma and final claims, bit for bit, on
every rank (the unsharded argument is pinned to the oracle by tests/test_pushforward_gpu.py).  GM_PF_DIST_MIN = 2 keeps the levels of
the logup tree distributed down to two elements per rank, so the re-spreading of the halves runs at every level of these small
shapes; the default threshold is exercised by the larger shape."""
import os
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _claims_worker(rank, world, tag, x_log, d_log, nbits, dist_min, q, host_staged=False):
    """one rank: the unsharded argument as the reference (every rank computes it: small), then the sharded one on its windows;
    the incoming claims are the image's evaluations, as in Pippenger::prove"""
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        if dist_min:
            os.environ["GM_PF_DIST_MIN"] = str(dist_min)
        if host_staged is True:
            os.environ["GM_PF_HOST_STAGED"] = "1"
            # ... and the sharded dense objects exchange round sums down to ONE element per rank (the default gathers the slices once
            # they are down to 2^8 elements and finishes unsharded: at these small shapes that would be every round)
            os.environ["GM_SC_SHARD_GATHER_LOG"] = "0"
        if host_staged == "rank1-cannot-export" and rank == 1:
            os.environ["GM_SHM_NO_IPC"] = "1"
        from gkr_msm_amd import codec, dist as gd, harness as H
        from pyref import field as F
        from pyref import gkr as G
        from pyref import polys as PL
        y_size = (nbits + d_log - 1) // d_log
        y_log = (y_size - 1).bit_length()
        n = 1 << x_log
        pts = F.random_points(n, 5)
        sc = F.random_scalars(n, nbits, 6)
        sc[0] = 0
        d_pts = H.to_dev(codec.points_to_mont(pts))
        d_sc = H.to_dev(codec.ints_to_limbs(sc))
        image, digits, counter = G.bucketing_image(pts, sc, y_size, y_log, d_log, x_log)
        rng = F.SplitMix64(9)
        r = [rng.next_fr() for _ in range(y_log + d_log + x_log)]
        evs = [PL.evaluate_poly(p.to_dense(), r) for p in image]
        tape = [rng.next_fr() for _ in range(4)] + [rng.next_bits(128) for _ in range(3000)]
        plan = H.MsmPlan(x_log, d_log, y_size)
        plan.run(d_pts, d_sc)
        ref = H.pushforward_prove(plan, d_pts, y_log, r, evs, tape)
        y0, y1 = gd.window_range(rank, world, y_size)
        comm = gd.ShmComm("/gm-test-pf-%s" % tag, rank, world)
        plan_s = H.MsmPlan(x_log, d_log, y_size, y0, y1)
        plan_s.run(d_pts, d_sc)
        got = H.pushforward_prove(plan_s, d_pts, y_log, r, evs, tape, comm=comm)
        keys = ("msgs", "gamma", "tape_used", "rounds")
        ok = all(got[k] == ref[k] for k in keys) and all(
            (list(got[k][0]), list(got[k][1])) == (list(ref[k][0]), list(ref[k][1])) for k in ("matrix", "ac_c", "ac_d"))
        msg = "" if ok else "sharded argument differs (rounds %d vs %d, %d vs %d messages)" % (
            got["rounds"], ref["rounds"], len(got["msgs"]), len(ref["msgs"]))
        calls = comm.calls
        if host_staged == "two-comms":
            # a second communicator of the same job after the first one is gone: the peers' allocations are still open in this process
            # (IpcCache, csrc/shm_comm.hip) -- nothing is imported a second time
            opened_1 = comm.ipc_stats()
            comm.close()
            comm = gd.ShmComm("/gm-test-pf-%s-b" % tag, rank, world)
            again = H.pushforward_prove(plan_s, d_pts, y_log, r, evs, tape, comm=comm)
            opened_2 = comm.ipc_stats()
            if again["msgs"] != ref["msgs"] or again["gamma"] != ref["gamma"]:
                ok, msg = False, "the argument through the second communicator differs"
            elif opened_1[0] == 0:
                ok, msg = False, "the first communicator never opened a mapping: the device path was not exercised"
            elif opened_2[1] != 0 or opened_2[2] != opened_1[2] + opened_2[0]:
                # (the pool may hand the second proof other blocks as sources: NEW allocations are imported, none is closed or imported twice)
                ok, msg = False, "mappings were closed or re-imported: first comm %r, second comm %r (opened, closed, held)" % (opened_1, opened_2)
        q.put((rank, ok, msg, calls))
        comm.close()
    except Exception as e:
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc(), 0))


def _run(target, world, x_log, d_log, nbits, dist_min, host_staged=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = "%d-%d-%d-%d-%s" % (os.getpid(), world, x_log, dist_min, str(host_staged)[:5])
    procs = [ctx.Process(target=target, args=(r, world, tag, x_log, d_log, nbits, dist_min, q, host_staged)) for r in range(world)]
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
    for rank, ok, info, calls in sorted(res):
        assert ok is True, "rank %d: %s" % (rank, info)
    return sorted(res)


@pytest.mark.parametrize("host_staged", [False, True])   # the halves re-spread device to device (gm_comm::pull_dev, HIP IPC) / through the host
@pytest.mark.parametrize("world,x_log,d_log,nbits,dist_min", [(2, 4, 2, 8, 2), (4, 5, 2, 8, 2), (2, 6, 3, 24, 2), (4, 7, 4, 32, 2),
                                                               (2, 10, 4, 16, 0), (4, 10, 8, 64, 0)])
def test_sharded_pushforward_matches_unsharded(world, x_log, d_log, nbits, dist_min, host_staged):
    res = _run(_claims_worker, world, x_log, d_log, nbits, dist_min, host_staged)
    for rank, ok, info, calls in res:
        assert calls > x_log      # the round sums (and the re-spread halves) really went through the communicator


def _claims_worker_q(rank, world, q, tag, x_log, d_log, nbits, dist_min):
    _claims_worker(rank, world, tag, x_log, d_log, nbits, dist_min, q)


@pytest.mark.parametrize("x_log,d_log,nbits,dist_min", [(6, 2, 32, 2), (8, 4, 64, 0)])
def test_sharded_pushforward_world_8(x_log, d_log, nbits, dist_min):
    """8 ranks x 2 windows as 4 processes x 2 rank threads (tests/rank_threads.py): the halves of every logup-tree level cross both
    kinds of peers -- threads of the same process (reached by address) and other processes (HIP IPC)"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from rank_threads import run_ranks
    res = run_ranks(_claims_worker_q, 8, ("w8-%d-%d" % (os.getpid(), x_log), x_log, d_log, nbits, dist_min), threads_per_proc=2)
    for rank, ok, info, calls in res:
        assert calls > x_log


def test_sharded_pushforward_when_one_rank_cannot_export_its_buffers():
    """gm_comm::pull_dev answers "unavailable" on EVERY rank when one of them cannot export (or open) an IPC mapping -- devices hidden
    from each other, IPC switched off -- and the argument stages that redistribution through the host instead: same result, no hang"""
    _run(_claims_worker, 4, 7, 4, 32, 2, "rank1-cannot-export")


def test_a_second_communicator_reuses_the_first_ones_ipc_mappings():
    """two communicators in a row in the same processes (what bench.py --gpus N does): the second finds every peer allocation still
    open -- HIP IPC mappings belong to the process -- and proves the same argument; closing and re-importing them was the cause of the
    intermittent wrong reads of round 4"""
    res = _run(_claims_worker, 4, 5, 2, 8, 2, host_staged="two-comms")
    for rank, ok, msg, _ in sorted(res):
        assert ok, "rank %d: %s" % (rank, msg)

