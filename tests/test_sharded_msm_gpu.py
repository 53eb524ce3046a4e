"""BASELINE.json configs[3]'s structure -- the MSM sharded by windows over 8 ranks, the window points all-gathered and recombined -- on
the one GPU of the box: 8 ranks as 4 processes x 2 rank threads (tests/rank_threads.py), 32 windows of 8 bits, 4 windows per rank; the
exchange is the library's shared-memory communicator (one all-gather of 27 x 4 field elements per rank, as the RCCL all-gather of
bench.py --gpus 8 carries).  Every rank must assemble the unsharded MSM's window points and final point; rank 0 also checks them
against the C oracle."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, q, tag, x_log, d_log, nbits):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ctypes as C
        from gkr_msm_amd import codec, dist as gd, ffi, harness as H
        from pyref import field as F
        L = ffi.lib()
        y_size = (nbits + d_log - 1) // d_log
        n = 1 << x_log
        pts = codec.points_to_mont(F.random_points(n, 31))
        sc = codec.ints_to_limbs(F.random_scalars(n, nbits, 32))
        d_pts, d_sc = H.to_dev(pts), H.to_dev(sc)
        full = H.MsmPlan(x_log, d_log, y_size)
        full.run(d_pts, d_sc)
        ref = full.window_points_raw()
        comm = gd.ShmComm("/gm-test-msm-%s" % tag, rank, world)
        y0, y1 = gd.window_range(rank, world, y_size)
        plan = H.MsmPlan(x_log, d_log, y_size, y0, y1)
        plan.run(d_pts, d_sc)
        mine = plan.window_points_raw()                     # (ncols, windows of this rank, 4)
        ncols, wpr = mine.shape[0], mine.shape[1]
        buf = np.zeros((world, ncols, wpr, 4), dtype=np.uint64)
        buf[rank] = mine
        rc = comm.c.all_gather(comm.c.ctx, buf.ctypes.data, ncols * wpr * 32)
        assert rc == 0
        raw = np.ascontiguousarray(np.transpose(buf, (1, 0, 2, 3)).reshape(ncols, world * wpr, 4))
        ok = bool(np.array_equal(raw, ref)) and H.combine_host(raw, d_log) == H.combine_host(ref, d_log)
        info = ""
        if rank == 0:
            import oracle_ffi as O
            o = O.msm(pts, sc, x_log, d_log, y_size, threads=4, want_aux=False)
            if not np.array_equal(raw, o["window_cols"]):
                ok, info = False, "assembled window points differ from the C oracle"
        q.put((rank, ok, info or ("" if ok else "assembled window points differ from the unsharded MSM")))
        comm.close()
    except Exception as e:
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc()))


@pytest.mark.parametrize("x_log,d_log,nbits", [(10, 8, 256), (7, 4, 128)])
def test_window_sharded_msm_over_eight_ranks(x_log, d_log, nbits):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from rank_threads import run_ranks
    run_ranks(_worker, 8, ("%d-%d" % (os.getpid(), x_log), x_log, d_log, nbits), threads_per_proc=2)
