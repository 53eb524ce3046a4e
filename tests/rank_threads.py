"""Running the ranks of a sharded test on ONE GPU box.  A box allows few processes on its card (six), so worlds above that run their
ranks as THREADS: `threads_per_proc` rank threads per process, each with a CUDA stream of its own (ranks that share a stream would
wait behind each other's pre-enqueued kernels), the process's streams spread over enough hardware queues.  That is also a deployment
form of its own -- one process driving several GPUs -- and what the library has to get right for it (no device-wide synchronisation
on a sharded path, no blocking wait for the co-residency budget) is what these runs pin."""
import os
import threading

import torch.multiprocessing as mp


def _group(target, ranks, world, q, args, env):
    for k, v in (env or {}).items():
        os.environ[k] = v
    if len(ranks) == 1:
        target(ranks[0], world, q, *args)
        return
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch

    def go(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                target(rank, world, q, *args)
                torch.cuda.current_stream().synchronize()
        except Exception as e:   # the target reports its own failures; this is for what escapes it
            import traceback
            q.put((rank, False, repr(e) + traceback.format_exc()))
    th = [threading.Thread(target=go, args=(r,)) for r in ranks]
    for t in th:
        t.start()
    for t in th:
        t.join()


def run_ranks(target, world, args=(), threads_per_proc=1, timeout=300, env=None):
    """target(rank, world, q, *args) puts exactly one tuple starting (rank, ok, info, ...) on q; -> the tuples sorted by rank"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    groups = [list(range(p, min(world, p + threads_per_proc))) for p in range(0, world, threads_per_proc)]
    assert len(groups) <= 6, "a GPU box allows six processes on its card"
    procs = [ctx.Process(target=_group, args=(target, g, world, q, args, env)) for g in groups]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert len(res) == world
    bad = ["rank %d: %s" % (t[0], str(t[2])[:1500]) for t in sorted(res, key=lambda t: t[0]) if t[1] is not True]
    assert not bad, "\n".join(bad)
    return sorted(res, key=lambda t: t[0])
