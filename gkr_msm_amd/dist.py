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
