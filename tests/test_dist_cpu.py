"""CPU test of the N > 1 path: world_size-2 gloo.  Each rank owns half of the MSM windows, the window points are
all-gathered and recombined exactly as bench.py does on GPUs (gkr_msm_amd/dist.py + gm_msm_combine_host).  The per-rank
window points come from the C oracle here (there is no GPU in this container); on the GPU box the same plumbing is fed by
gm_msm_run (tests/test_msm_gpu.py::test_window_sharding_matches_full covers the device side of the partition)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, x_log, d_log, nbits, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi as O
    from gkr_msm_amd import codec, dist as gd, harness
    from pyref import field as F
    y_size = (nbits + d_log - 1) // d_log
    n = 1 << x_log
    pts = codec.points_to_mont(F.random_points(n, 5))
    sc = codec.ints_to_limbs(F.random_scalars(n, nbits, 6))
    y0, y1 = gd.window_range(rank, world, y_size)
    mine = O.msm(pts, sc, x_log, d_log, y_size, y0, y1, threads=1, want_aux=False)["window_cols"]
    local = torch.from_numpy(mine.view(np.int64).copy())
    raw = gd.gather_window_points(dist, local, world)
    got = harness.combine_host(raw, d_log)
    dist.barrier()
    if rank == 0:
        full = O.msm(pts, sc, x_log, d_log, y_size, threads=1, want_aux=False)["window_cols"]
        q.put((bool(np.array_equal(raw, full)), got))
    dist.destroy_process_group()


@pytest.mark.parametrize("x_log,d_log,nbits", [(6, 4, 32), (5, 8, 64)])
def test_window_sharded_msm_world2_gloo(x_log, d_log, nbits):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyref import field as F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x_log, d_log, nbits, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    res = None
    for _ in range(120):
        try:
            res = q.get(timeout=1)
            break
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    assert res is not None and all(p.exitcode == 0 for p in procs), "a rank failed"
    same, got = res
    assert same, "gathered window points differ from the unsharded run"
    n = 1 << x_log
    pts = F.random_points(n, 5)
    sc = F.random_scalars(n, nbits, 6)
    acc = (0, 1)
    for p_, s in zip(pts, sc):
        acc = F.te_add_affine(acc, F.te_mul_affine(p_, s))
    assert tuple(got) == acc


def test_window_range_partition():
    sys.path.insert(0, ROOT)
    from gkr_msm_amd import dist as gd
    assert [gd.window_range(r, 8, 32) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    with pytest.raises(ValueError):
        gd.window_range(0, 3, 32)


def _comm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    from gkr_msm_amd import codec, dist as gd, ffi
    from pyref import field as F
    comm = gd.Comm(dist, rank, world)
    rng = F.SplitMix64(100 + rank)
    vals = [rng.next_fr() for _ in range(3)]
    buf = codec.to_mont_limbs(vals)
    ffi.check(ffi.lib().gm_comm_sum_fr(C.byref(comm.c), buf.ctypes.data, 3))
    q.put((rank, vals, codec.from_mont_limbs(buf), comm.calls))
    dist.barrier()
    dist.destroy_process_group()


def test_comm_field_sum_world2_gloo():
    """the per-round exchange of the sharded prover: partial round sums of all ranks added mod p through the
    gm_comm all-gather callback -- host only, world_size 2, gloo"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyref import field as F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs)
    res.sort()
    want = [(a + b) % F.P for a, b in zip(res[0][1], res[1][1])]
    assert res[0][2] == want and res[1][2] == want
    assert res[0][3] == 1 and res[1][3] == 1
