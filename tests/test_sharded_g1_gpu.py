"""The G1 side of a window-sharded run: with commitment_log_multiplicity > 0 one commitment matrix spans 2^clm windows,
i.e. several ranks (pushforward.rs:395-396, 431-456).  Every rank accumulates the outer buckets of ITS windows from the key slices it
holds (gm_msm_g1_outer_part) and the ranks exchange one group element per matrix and commitment (gm_g1_combine_parts): the combined
d_comm / c_comm must be those of the unsharded gm_msm_g1_outer, which tests/test_g1_gpu.py pins to the oracle.  world-2 / world-4
processes share the one GPU and exchange over gloo; clm = 4 with 20 windows gives a full matrix across ranks plus a partial one."""
import os
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, x_log, d_log, nbits, clm, q, transport="gloo"):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist = None
        if transport == "gloo":
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from gkr_msm_amd import codec, dist as gd, harness as H
        from pyref import field as F
        y_size = (nbits + d_log - 1) // d_log
        n, cm = 1 << x_log, 1 << clm
        d_pts = H.to_dev(codec.points_to_mont(F.random_points(n, 5)))
        sc = F.random_scalars(n, nbits, 6)
        sc[0] = 0
        d_sc = H.to_dev(codec.ints_to_limbs(sc))
        d_basis = H.g1_gen_points(n * cm, 0x5253)           # the whole key (every rank can make it here: the test is small)
        plan = H.MsmPlan(x_log, d_log, y_size)
        plan.run(d_pts, d_sc)
        n_mat = (y_size + cm - 1) >> clm
        _, _, _, d_comm, c_comm = H.msm_g1_outer(plan, d_basis, clm, n)
        # this rank: its windows, and ONLY the key slices they use, packed
        y0, y1 = gd.window_range(rank, world, y_size)
        need = sorted({y % cm for y in range(y0, y1)})
        slot = {s_: i for i, s_ in enumerate(need)}
        basis_h = H.to_host(d_basis).reshape(cm, n * 12)
        d_local = H.to_dev(basis_h[need].reshape(-1))
        plan_s = H.MsmPlan(x_log, d_log, y_size, y0, y1)
        plan_s.run(d_pts, d_sc)
        part = H.msm_g1_outer_part(plan_s, d_local, slot, clm, n)
        comm = gd.Comm(dist, rank, world) if dist is not None else gd.ShmComm("/gm-test-g1-%d" % port, rank, world)
        got_d = H.g1_combine_parts(comm, part["d_part"], n_mat, part["first_matrix"])
        got_c = H.g1_combine_parts(comm, part["c_part"], n_mat, part["first_matrix"])
        ok = got_d == d_comm and got_c == c_comm and part["n_matrices"] == ((y1 - 1) >> clm) - (y0 >> clm) + 1
        q.put((rank, ok, "slices held %s of %d; matrices %d..%d" % (need, cm, part["first_matrix"], part["first_matrix"] + part["n_matrices"] - 1)))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:  # report instead of hanging the parent
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc()))


@pytest.mark.parametrize("world,x_log,d_log,nbits,clm", [(2, 5, 2, 40, 4), (4, 4, 2, 40, 4), (2, 6, 3, 24, 2), (2, 5, 3, 12, 0)])
def test_sharded_outer_commitments_match_unsharded(world, x_log, d_log, nbits, clm):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, x_log, d_log, nbits, clm, q)) for r in range(world)]
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
    for rank, ok, info in sorted(res):
        assert ok is True, "rank %d: %s" % (rank, info)


def _worker_q(rank, world, q, port, x_log, d_log, nbits, clm):
    _worker(rank, world, port, x_log, d_log, nbits, clm, q, "shm")


def test_sharded_outer_commitments_world_8():
    """config E's structure: 32 windows, commitment_log_multiplicity 4, 8 ranks x 4 windows -- matrix 0 spans ranks 0-3, matrix 1
    ranks 4-7; each rank holds 4 of the 16 key slices (4 processes x 2 rank threads, the shared-memory communicator)"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from rank_threads import run_ranks
    run_ranks(_worker_q, 8, (36900 + os.getpid() % 2000, 4, 2, 64, 4), threads_per_proc=2)
