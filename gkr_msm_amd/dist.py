"""Multi-GPU plumbing for the window-sharded MSM (harness level; one process per GPU).

The path shards by MSM window: rank g of G owns windows [g*y_size/G, (g+1)*y_size/G); operands are
replicated; the only exchange is the all-gather of the window points (3*(d+1) field elements per window) before the
host-side Horner recombination.  Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
import numpy as np


def window_range(rank, world, y_size):
    if y_size % world != 0:
        raise ValueError("windows (%d) must divide over %d ranks" % (y_size, world))
    wpr = y_size // world
    return rank * wpr, (rank + 1) * wpr


def gather_window_points(dist, local, world):
    """local: int64 tensor (ncols, wpr, 4) holding this rank's window points (bit pattern of u64 Montgomery limbs).
    Returns a numpy uint64 array (ncols, y_size, 4) with every rank's windows in window order."""
    import torch
    if world == 1:
        return local.cpu().numpy().view(np.uint64)
    ncols, wpr = local.shape[0], local.shape[1]
    # flat output (rank-major along dim 0): the layout both the nccl and the gloo backends accept
    gathered = torch.empty((world * ncols, wpr, 4), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous())
    raw = gathered.cpu().numpy().view(np.uint64).reshape(world, ncols, wpr, 4)
    return np.ascontiguousarray(np.transpose(raw, (1, 0, 2, 3)).reshape(ncols, world * wpr, 4))


class Comm:
    """gm_comm over torch.distributed: the one collective the sharded prover needs (an all-gather of a few bytes per round,
    of the bucket sums once per proof).  Works with "gloo" (CPU tensors; the CPU tests and the shared-GPU test) and with
    "nccl" (= RCCL: the payload is staged through a device tensor on `device`)."""

    def __init__(self, dist, rank, world, device=None):
        import ctypes as C
        import torch
        from . import ffi
        self.dist, self.rank, self.world, self.calls, self.bytes = dist, rank, world, 0, 0

        def _ag(ctx, buf, nbytes):
            try:
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(world * nbytes,))
                mine = torch.from_numpy(arr[rank * nbytes:(rank + 1) * nbytes].copy())
                if device is not None:
                    mine = mine.to(device)
                out = torch.empty(world * nbytes, dtype=torch.uint8, device=mine.device)
                dist.all_gather_into_tensor(out, mine)
                arr[:] = out.cpu().numpy()
                self.calls += 1
                self.bytes += nbytes
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("gm_comm all_gather failed: %r" % (e,), flush=True)
                return 9
        self._cb = ffi.ALL_GATHER_CB(_ag)  # keep the thunk alive
        self.c = ffi.GmComm(None, rank, world, self._cb)


class RcclComm:
    """The library's own RCCL communicator (gm_comm_rccl_*, csrc/rccl_comm.hip): ncclAllGather / ncclBroadcast inside
    libgkrmsm_hip.so, no Python on the data path.  torch.distributed (any backend) is only the side channel that carries the
    128-byte ncclUniqueId from rank 0 to the other ranks at start-up; pass `dist=None` with `unique_id=` to use another one."""

    def __init__(self, dist, rank, world, unique_id=None, bcast_device=None):
        import ctypes as C
        import torch
        from . import ffi, harness
        L = ffi.lib()
        self.L, self.rank, self.world = L, rank, world
        if unique_id is None:
            idb = np.zeros(128, dtype=np.uint8)
            if rank == 0:
                ffi.check(L.gm_comm_rccl_unique_id(idb.ctypes.data))
            if world > 1:
                t = torch.from_numpy(idb)
                if bcast_device is not None:
                    t = t.to(bcast_device)
                dist.broadcast(t, src=0)
                idb = t.cpu().numpy().copy()
            unique_id = idb
        self.unique_id = np.ascontiguousarray(unique_id, dtype=np.uint8)
        self.h = C.c_void_p()
        ffi.check(L.gm_comm_rccl_create(self.unique_id.ctypes.data, rank, world, C.byref(self.h), harness.cur_stream()))
        self.c = ffi.GmComm()
        ffi.check(L.gm_comm_rccl_as_comm(self.h, C.byref(self.c)))

    def close(self):
        if self.h:
            self.L.gm_comm_rccl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def calls(self):
        import ctypes as C
        n, b = C.c_uint64(), C.c_uint64()
        self.L.gm_comm_rccl_stats(self.h, C.byref(n), C.byref(b))
        return n.value

    def all_gather_dev(self, d_send_ptr, d_recv, nbytes):
        """d_send_ptr: raw device pointer (int / c_void_p) of nbytes; d_recv: tensor of world * nbytes bytes"""
        import ctypes as C
        from . import ffi, harness
        ffi.check(self.L.gm_comm_rccl_all_gather_dev(self.h, d_send_ptr, C.c_void_p(d_recv.data_ptr()), nbytes, harness.cur_stream()))

    def broadcast_dev(self, tensor, root=0):
        import ctypes as C
        from . import ffi, harness
        ffi.check(self.L.gm_comm_rccl_broadcast_dev(self.h, C.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size(), root,
                                                    harness.cur_stream()))

    def gather_window_points(self, plan):
        """all ranks' window points of a window-sharded gm_msm_plan -> numpy uint64 (ncols, y_size, 4), window order"""
        import ctypes as C
        import torch
        from . import ffi
        p, nc, cl = C.c_void_p(), C.c_uint64(), C.c_uint64()
        ffi.check(self.L.gm_msm_window_points(plan.h, C.byref(p), C.byref(nc), C.byref(cl)))
        ncols, wpr = nc.value, cl.value
        nbytes = ncols * wpr * 32
        recv = torch.empty(self.world * ncols * wpr * 4, dtype=torch.int64, device="cuda")
        self.all_gather_dev(p, recv, nbytes)
        raw = recv.cpu().numpy().view(np.uint64).reshape(self.world, ncols, wpr, 4)
        return np.ascontiguousarray(np.transpose(raw, (1, 0, 2, 3)).reshape(ncols, self.world * wpr, 4))


def shard_clock(reset=True):
    """gm_shard_clock of the calling thread as a dict (ms)"""
    import ctypes as C
    from . import ffi
    o = (C.c_double * 8)()
    ffi.check(ffi.lib().gm_shard_clock(1 if reset else 0, o))
    return dict(small_gather_ms=round(o[0] / 1e3, 2), small_gathers=int(o[1]), bulk_gather_ms=round(o[2] / 1e3, 2), bulk_gathers=int(o[3]),
                bulk_MB_per_rank=round(o[4] / 1e6, 2), pull_ms=round(o[5] / 1e3, 2), pulls=int(o[6]), pulled_MB=round(o[7] / 1e6, 2))


class ShmComm:
    """The library's one-node communicator over POSIX shared memory (gm_comm_shm_*, csrc/shm_comm.hip): the ranks' host threads
    exchange the per-round sums of a sharded proof directly -- no device collective, no Python on the path.  `name` ("/gm-...")
    must be the same on every rank and unique per job; creation is collective."""

    def __init__(self, name, rank, world):
        import ctypes as C
        from . import ffi
        L = ffi.lib()
        self.L, self.rank, self.world = L, rank, world
        self.h = C.c_void_p()
        ffi.check(L.gm_comm_shm_create(name.encode(), rank, world, C.byref(self.h)))
        self.c = ffi.GmComm()
        ffi.check(L.gm_comm_shm_as_comm(self.h, C.byref(self.c)))

    def close(self):
        if self.h:
            self.L.gm_comm_shm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def calls(self):
        import ctypes as C
        n, b = C.c_uint64(), C.c_uint64()
        self.L.gm_comm_shm_stats(self.h, C.byref(n), C.byref(b))
        return n.value

    def ipc_stats(self):
        """(opened, closed, held now): the HIP IPC mappings of peers' allocations behind pull_dev"""
        import ctypes as C
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.L.gm_comm_shm_ipc_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def sum_fr(self, vals):
        """self-test of the seam: field sums over the ranks of a list of Montgomery elements (numpy uint64 (n, 4)), in place"""
        from . import ffi
        ffi.check(self.L.gm_comm_sum_fr(self.c, vals.ctypes.data, vals.shape[0]))
        return vals
