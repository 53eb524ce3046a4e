"""The one-node gm_comm over POSIX shared memory (csrc/shm_comm.hip): host only, no GPU.  world_size 2 and 4: the
per-round exchange of the sharded prover (field sums of every rank's partial round sums), many calls in a row (slot reuse), a payload
larger than one slot (chunking: the once-per-proof bucket sums), and the bounded wait when a rank never shows up."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, name, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from gkr_msm_amd import codec, dist as gd
        from pyref import field as F
        comm = gd.ShmComm(name, rank, world)
        rng = F.SplitMix64(100 + rank)
        sums = []
        for it in range(300):   # a proof's worth of small exchanges: slot n & 1 is reused every other call
            vals = [rng.next_fr() for _ in range(3)]
            buf = codec.to_mont_limbs(vals)
            comm.sum_fr(buf)
            if it % 50 == 0 or it == 299:
                sums.append((it, codec.from_mont_limbs(buf)))
        # one payload above the slot size (256 KiB): 700 000 bytes per rank, rank-dependent content
        n = 700000
        buf = np.zeros(world * n, dtype=np.uint8)
        mine = (np.arange(n, dtype=np.uint64) * (rank + 3) + rank).astype(np.uint8)
        buf[rank * n:(rank + 1) * n] = mine
        rc = comm.c.all_gather(comm.c.ctx, buf.ctypes.data, n)
        ok_big = rc == 0
        for r in range(world):
            want = (np.arange(n, dtype=np.uint64) * (r + 3) + r).astype(np.uint8)
            ok_big = ok_big and bool(np.array_equal(buf[r * n:(r + 1) * n], want))
        q.put((rank, sums, ok_big, comm.calls))
        comm.close()
    except Exception as e:  # report instead of hanging the parent
        import traceback
        q.put((rank, repr(e) + traceback.format_exc(), False, 0))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shm_comm_field_sums_and_large_gather(world):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyref import field as F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/gm-test-%d-%d" % (os.getpid(), world)
    procs = [ctx.Process(target=_worker, args=(r, world, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    res.sort()
    for rank, sums, ok_big, calls in res:
        assert isinstance(sums, list), "rank %d: %s" % (rank, sums)
        assert ok_big, "rank %d: the chunked all-gather returned other bytes" % rank
        assert calls == 301
    # the expected sums, replayed from the ranks' generators
    rngs = [F.SplitMix64(100 + r) for r in range(world)]
    want = {}
    for it in range(300):
        vals = [[g.next_fr() for _ in range(3)] for g in rngs]
        want[it] = [sum(v[i] for v in vals) % F.P for i in range(3)]
    for rank, sums, _, _ in res:
        for it, got in sums:
            assert got == want[it], "rank %d, call %d" % (rank, it)
    assert not os.path.exists("/dev/shm" + name), "the name is removed once every rank has attached"


def test_shm_comm_ranks_as_threads_of_one_process():
    """8 ranks as threads of this process (one process driving every GPU of a node): every thread attaches under its own rank"""
    import queue
    import threading
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    q = queue.Queue()
    name = "/gm-test-thr-%d" % os.getpid()
    th = [threading.Thread(target=_worker, args=(r, 8, name, q)) for r in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    res = sorted(q.get_nowait() for _ in range(8))
    for rank, sums, ok_big, calls in res:
        assert isinstance(sums, list), "rank %d: %s" % (rank, sums)
        assert ok_big and calls == 301
    assert all(res[r][1] == res[0][1] for r in range(8))


def test_shm_comm_leaves_no_name_behind_when_a_rank_never_attaches():
    """advisor r03: rank 0 created the object, rank 1 never came -- the failure path must remove /dev/shm/<name> (O_EXCL would
    refuse the next job of that name otherwise)"""
    sys.path.insert(0, ROOT)
    from gkr_msm_amd import dist as gd, ffi
    name = "/gm-test-orphan-%d" % os.getpid()
    ffi.lib().gm_set_wait_timeout_ms(200)
    try:
        with pytest.raises(Exception) as e:
            gd.ShmComm(name, 0, 2)
        assert "attached" in str(e.value)
        assert not os.path.exists("/dev/shm" + name)
    finally:
        ffi.lib().gm_set_wait_timeout_ms(20000)


def _lonely(name, q):
    sys.path.insert(0, ROOT)
    from gkr_msm_amd import dist as gd, ffi
    ffi.lib().gm_set_wait_timeout_ms(300)
    try:
        gd.ShmComm(name, 1, 2)   # rank 0 never creates the object
        q.put("created")
    except Exception as e:
        q.put(str(e))


def test_shm_comm_wait_is_bounded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_lonely, args=("/gm-test-lonely-%d" % os.getpid(), q))
    p.start()
    msg = q.get(timeout=60)
    p.join(timeout=30)
    assert "did not appear" in msg, msg
