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
