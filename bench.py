#!/usr/bin/env python3
"""bench.py -- headline benchmark: Pippenger MSM points/sec + sumcheck rounds/sec on MI355X (BASELINE.json metric).

A "step" is one full pass of the MSM hot path over one batch of synthetic input resident in HBM:
digits -> stable bucket scatter -> bucket sums (x_logsize levels of pairwise adds) -> bucket reduction ->
all-gather of the window points (N > 1, ncclAllGather inside the library) -> D2H of 27 x n_windows field elements -> host
recombination.

N = 1 : BASELINE.json configs[1]: x_logsize=20, d_logsize=8, nbits=256 (32 windows of 8 bits).
N > 1 : the path shards by MSM window: rank g owns windows [g*32/N, (g+1)*32/N); points/scalars are replicated
        (ncclBroadcast); the only exchange of a step is the all-gather of (d+1) points per window.  `value` is the metric's
        own configuration -- x_logsize=20 on N GPUs, total work fixed: "scaling": "strong" -- and the same line carries
        `weak` (x_logsize = 20 + log2 N: 2^25 bucket cells per rank, as at N = 1) and `config_d` (BASELINE.json configs[3]:
        x_logsize=24, windows sharded N ways), plus the window/row-sharded image-part prover's rounds/sec.

One JSON line on rank 0 (see README / DESIGN.md for the field meanings).  `roofline` is for the dominant MSM kernel (level-0
bucket add, k_add_level0) and `sumcheck.roofline` for the dominant round kernel of the prover, both timed with HIP events on
the launch stream inside the timed runs; `cpu_baseline` is the C oracle (oracle/gkrmsm_oracle*.c, OpenMP) on this host,
rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import math
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def _spawn_ranks_if_needed():
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU) and wait for them.
    Runs before torch / the HIP library are imported: the parent never touches a GPU and never execs; rank 0's JSON line goes to
    the inherited stdout; a failed rank fails the run (the others are ended after a grace period so nothing keeps a GPU)."""
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, GM_BENCH_SELF_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, deadline = 0, None
    live = list(procs)
    while live:
        for p_ in list(live):
            c = p_.poll()
            if c is not None:
                live.remove(p_)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    deadline = time.time() + 30.0      # a rank died: the others get 30 s to notice (collective time-outs), then are ended
        if deadline is not None and time.time() > deadline:
            for p_ in live:
                p_.terminate()
            for p_ in live:
                try:
                    p_.wait(10)
                except subprocess.TimeoutExpired:
                    p_.kill()
            break
        time.sleep(0.05)
    sys.exit(rc)


if __name__ == "__main__":
    _spawn_ranks_if_needed()
    if os.environ.get("GM_BENCH_SPAWN_CHECK"):
        # test hook of the launcher (tests/test_bench_spawn_cpu.py): a rank reports how it was started and leaves before anything
        # touches a GPU; "fail:<r>" makes rank r exit non-zero
        chk = os.environ["GM_BENCH_SPAWN_CHECK"]
        print("spawn-check rank %s of %s local %s port %s self_spawned %s" % (
            os.environ.get("RANK"), os.environ.get("WORLD_SIZE"), os.environ.get("LOCAL_RANK"), os.environ.get("MASTER_PORT"),
            os.environ.get("GM_BENCH_SELF_SPAWNED", "0")), flush=True)
        sys.exit(3 if chk == "fail:%s" % os.environ.get("RANK") else 0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gkr_msm_amd import codec, ffi, harness  # noqa: E402
from gkr_msm_amd import dist as gdist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FR_MUL_CEILING = 115e9     # measured: scripts/ubench/fr_mul_asm_test.hip, the 302-instruction 8 x 32-bit multiplier, all 256 CUs
FR9_MUL_CEILING = 175e9    # measured: scripts/ubench/fr9_rate_vs_occupancy.hip, the 205-instruction 9 x 29-bit multiplier in chains of dependent
                           # products at <= 6 waves per SIMD or two interleaved chains (153 G at 8 waves per SIMD with one chain, the figure
                           # quoted until the occupancy sweep); MSM level kernels, large-round kernels below
FR9_ROUND_PRIMS = ("PROJ_L1", "PROJ_L2", "PROJ_L3", "AFF_L1", "AFF_L2", "AFF_L3", "PT_BIT_CHOICE", "ADD_INVERSES", "LOGUP_LAYER")   # large-round kernels that compute in the 9 x 29 form (sumcheck.hip: lean9_has)


def round_kernel_ceiling(name):
    """multiplier ceiling of a large-round kernel by the form it computes in (GM_LEAN_FR9=0 puts all of them on 8 x 32)"""
    if os.environ.get("GM_LEAN_FR9", "1")[:1] != "0" and name.startswith("k_round_deg2_lean<") and any(
            name.split("<")[1].startswith(p_) for p_ in FR9_ROUND_PRIMS):
        return FR9_MUL_CEILING, "9x29"
    return FR_MUL_CEILING, "8x32"

P = codec.P
SEED = 0x474B524D534D      # "GKRMSM"

# HBM traffic per launch of the dominant kernels at config B from the committed PMC passes (profiles/r03/*_pmc_hbm.csv:
# (2 * FETCH_SIZE + WRITE_SIZE) * 1024 with the guide's gfx950 FETCH_SIZE correction); None for any other shape
PMC_FETCH_FACTOR_GATHER = 1    # 64-byte gathers: FETCH_SIZE reads the bytes exactly (scripts/ubench/fetch_size_gather.hip, profiles/r04)
PMC_TRAFFIC_MSM_B = None
PMC_TRAFFIC_GE1_B = None   # bytes per step of all k_add_level + k_add_tail launches together
PMC_TRAFFIC_SC_B = {}      # kernel name -> bytes per launch (profiles/r03/prover_pmc_per_launch.json, written by scripts/summarise_profiles.py)
try:
    _pj = [os.path.join(ROOT, "profiles", r_, "prover_pmc_per_launch.json") for r_ in ("r04", "r03", "r02")]
    with open([p_ for p_ in _pj if os.path.exists(p_)][0]) as _f:
        PMC_TRAFFIC_SC_B = json.load(_f)
    PMC_TRAFFIC_MSM_B = PMC_TRAFFIC_SC_B.get("k_add_level0")
    PMC_TRAFFIC_GE1_B = PMC_TRAFFIC_SC_B.get("k_add_levels_ge1_per_step")
except Exception:
    pass


def log(msg):
    print(msg, file=sys.stderr, flush=True)


def host_info():
    model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "usable_cpus": usable_cpus(), "cpu_model": model}


def usable_cpus():
    """hardware threads this process may actually use: the affinity mask, cut by a cgroup CPU quota when there is one"""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(math.ceil(int(q) / int(per)))))
    except Exception:
        pass
    return n


def make_scalars(n, nbits, seed=0x474B524D):
    rng = np.random.default_rng(seed)
    sc = rng.integers(0, 2 ** 64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)  # uniform below 2^252 (< Bandersnatch order): canonical bigints
    if nbits < 256:
        full = nbits // 64
        for limb in range(4):
            if limb > full:
                sc[:, limb] = 0
            elif limb == full:
                sc[:, limb] &= np.uint64((1 << (nbits % 64)) - 1)
    return sc


def claims_for(w_, y_log_, seed):
    outs, _ = w_.outputs()
    pr = np.random.default_rng(seed)
    r = [int.from_bytes(pr.bytes(64), "little") % P for _ in range(y_log_)]

    def ev(poly):
        cur = list(poly)
        for f in reversed(r):
            cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
        return cur[0]
    tape = [int.from_bytes(pr.bytes(16), "little") for _ in range(4000)]
    return r, [ev(o) for o in outs], tape


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--x-logsize", type=int, default=None)
    ap.add_argument("--d-logsize", type=int, default=8)
    ap.add_argument("--nbits", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU port (0 = every usable hardware thread; the MSM leg "
                    "cannot use more than its windows, as the reference's rayon loop over windows)")
    ap.add_argument("--no-sumcheck", action="store_true")
    ap.add_argument("--cpu-sumcheck-xlog", type=int, default=20, help="CPU prover sample (config B itself by default: ~25 s)")
    ap.add_argument("--cpu-faithful-xlog", type=int, default=17, help="sample of the 'reference-faithful' CPU variant (serial round loops)")
    ap.add_argument("--gen1-log-points", type=int, default=20, help="gen-1 gkr_msm_prove size (0 = skip)")
    ap.add_argument("--cpu-gen1-log-points", type=int, default=12)
    ap.add_argument("--g1-log-points", type=int, default=21, help="BLS12-381 G1 MSM size (KZG commit shape; 0 = skip)")
    ap.add_argument("--cpu-g1-log-points", type=int, default=19)
    ap.add_argument("--cpu-g1-outer-xlog", type=int, default=17)
    ap.add_argument("--concurrent-provers", type=int, default=3, help="N = 1: image-part proofs in flight from this many host threads (0 = skip)")
    ap.add_argument("--no-extra-shapes", action="store_true", help="N > 1: only the metric's x_logsize=20 line (skip weak / config D)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (bench.py starts its own ranks when no launcher did)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # GM_BENCH_BACKEND=gloo rehearses the N > 1 flow on a box with fewer GPUs than ranks (ranks then share devices and the
    # small exchanges go over gloo on host tensors); the real runs use RCCL over xGMI, one GPU per rank.
    backend = os.environ.get("GM_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    xdev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")  # where torch collective payloads live
    dist = None
    rcomm = None     # the library's own RCCL communicator (data path); torch.distributed only bootstraps it and syncs the clock
    if world > 1:
        import datetime
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=300))
            try:
                rcomm = gdist.RcclComm(dist, rank, world, bcast_device=xdev)
            except Exception as e:  # keep going over torch.distributed; the line says which transport was used
                log("rank %d: native RCCL communicator unavailable (%r); falling back to torch.distributed" % (rank, e))
                rcomm = None
            ok = torch.tensor([1 if rcomm is not None else 0], dtype=torch.int32, device=xdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                rcomm = None
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    if world == 1 and os.environ.get("GM_BENCH_RCCL_WORLD1") == "1":
        # rehearsal of the native-communicator code path on one GPU: a world-1 ncclAllGather / ncclBroadcast per step
        rcomm = gdist.RcclComm(None, 0, 1)
    transport = ("none" if rcomm is None else "rccl-native, world 1 (rehearsal)") if world == 1 else ("rccl-native (ncclAllGather / ncclBroadcast inside libgkrmsm_hip.so)" if rcomm is not None
                                            else "torch.distributed/%s callback" % backend)

    L = ffi.lib()
    d_log, nbits = args.d_logsize, args.nbits
    y_size = (nbits + d_log - 1) // d_log
    y0, y1 = gdist.window_range(rank, world, y_size)
    wpr = y1 - y0
    ncols = 3 * (d_log + 1)
    world_, rcomm_, y0_, y1_ = world, rcomm, y0, y1     # msm_leg(solo=True) shadows them with the one-GPU values

    def gather_floats(v):
        """one float per rank, in rank order, on every rank"""
        if dist is None:
            return [float(v)]
        t = torch.tensor([v], dtype=torch.float64, device=xdev)
        o = torch.empty(world_, dtype=torch.float64, device=xdev)
        dist.all_gather_into_tensor(o, t)
        return [float(x) for x in o.cpu()]

    # which GPUs the native communicator really spans: every rank contributes (rank, a digest of its device's UUID / PCI address)
    # through ncclAllGather itself
    rccl_seen = None
    if rcomm is not None:
        import hashlib
        pr_ = torch.cuda.get_device_properties(local_rank)
        ident = "%s|%s|%s|%s" % (getattr(pr_, "uuid", ""), getattr(pr_, "pci_domain_id", ""), getattr(pr_, "pci_bus_id", ""),
                                 getattr(pr_, "pci_device_id", ""))
        dig = int.from_bytes(hashlib.sha256(ident.encode()).digest()[:7], "little")
        mine_ = torch.tensor([rank, dig], dtype=torch.int64, device="cuda")
        all_ = torch.empty(2 * world, dtype=torch.int64, device="cuda")
        rcomm.all_gather_dev(C.c_void_p(mine_.data_ptr()), all_, 16)
        torch.cuda.synchronize()
        got_ = all_.cpu().numpy().reshape(world, 2)
        rccl_seen = {"ranks": int(len(set(int(v) for v in got_[:, 0]))), "distinct_devices": int(len(set(int(v) for v in got_[:, 1])))}

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_inputs(x_log):
        """synthetic operands, identical on every rank: generated on rank 0 and replicated with ncclBroadcast when the native
        communicator is up (the path's one link-bound transfer), generated in place otherwise (same seeds)"""
        n = 1 << x_log
        d_pts = harness.dev_empty(n * 8)
        sc = make_scalars(n, nbits)
        bcast_ms = None
        if rcomm is not None:
            if rank == 0:
                ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, SEED, harness.cur_stream()))
                d_sc = harness.to_dev(sc)
            else:
                d_sc = harness.dev_empty(n * 4)
            sync_all()
            t0 = time.perf_counter()
            rcomm.broadcast_dev(d_pts, 0)
            rcomm.broadcast_dev(d_sc, 0)
            sync_all()
            bcast_ms = (time.perf_counter() - t0) * 1e3
        else:
            ffi.check(L.gm_gen_points(C.c_void_p(d_pts.data_ptr()), n, SEED, harness.cur_stream()))
            d_sc = harness.to_dev(sc)
        return d_pts, d_sc, sc, bcast_ms

    def msm_leg(x_log, steps, warmup, keep=False, solo=False):
        """the timed MSM loop at one shape; returns the result dict (+ plan / operands when keep).  solo: every rank runs the WHOLE
        unsharded MSM on its own GPU, no exchange -- the N = 1 figure measured in the same run, on the same boxes"""
        n = 1 << x_log
        d_pts, d_sc, sc, bcast_ms = make_inputs(x_log)
        world, rcomm, y0, y1 = (1, None, 0, y_size) if solo else (world_, rcomm_, y0_, y1_)
        wpr = y1 - y0
        plan = harness.MsmPlan(x_log, d_log, y_size, y0, y1)

        def step():
            plan.run(d_pts, d_sc)
            if world > 1 and rcomm is not None:
                raw = rcomm.gather_window_points(plan)
            elif world > 1:
                p, nc, cl = C.c_void_p(), C.c_uint64(), C.c_uint64()
                ffi.check(L.gm_msm_window_points(plan.h, C.byref(p), C.byref(nc), C.byref(cl)))
                mine = torch.empty((ncols, wpr, 4), dtype=torch.int64, device="cuda")
                ffi.check(L.gm_memcpy_d2d(C.c_void_p(mine.data_ptr()), p, ncols * wpr * 32, harness.cur_stream()))
                raw = gdist.gather_window_points(dist, mine.to(xdev), world)
            else:
                raw = plan.window_points_raw()
            return harness.combine_host(raw, d_log), raw

        for _ in range(warmup):
            step()
        ffi.check(L.gm_msm_profile(plan.h, 1))
        dom_ms = []
        prof = (C.c_float * 7)()
        # Software pipeline over two plans on two streams: while the GPU runs step i, the host reads back and recombines the
        # window points of step i - 1 (27 KB D2H + ~270 host curve operations, ~0.2 ms that would otherwise idle the GPU; with
        # N ranks also the all-gather of the window points, enqueued on the step's own stream).  Every step still produces its
        # final group element inside the timed region, on every rank.
        # Depth: 2 steps in flight when a step is milliseconds of large kernels; 4 when a rank's share is small (x_logsize 20 over
        # 8 ranks: ~25 launches of 10-200 us, a 0.93 ms dependency chain for 0.5 ms of work -- measured on one GPU with
        # scripts/quick_rank_share_time.py: 0.97 ms unpipelined, 0.66 ms at depth 2, 0.55 ms at depth 4).
        # (round 4, same script, with levels 0 + 1 fused: 8 ranks' share 0.472 ms at depth 4, 0.402 at 6, 0.391 at 8 against 2.86 / 8 = 0.358
        # for perfect scaling of the one-GPU step; 4 ranks' share 0.776 / 0.719 / 0.720; 2 ranks' 1.50 / 1.38 / 1.37)
        depth = int(os.environ.get("GM_BENCH_DEPTH", "0")) or (2 if world == 1 or wpr * n > (1 << 24) else (8 if wpr * n <= (1 << 22) else 6))
        plans = [plan] + [harness.MsmPlan(x_log, d_log, y_size, y0, y1) for _ in range(depth - 1)]
        streams = [torch.cuda.Stream() for _ in range(depth)]
        recv = [torch.empty(world * ncols * wpr * 4, dtype=torch.int64, device="cuda") for _ in range(depth)] if rcomm is not None else None

        # RCCL sees ONE stream: every all-gather is enqueued on comm_stream, ordered after its step's kernels and before that plan's
        # next run by events (collectives of one communicator issued from several streams are legal but serialised inside RCCL in
        # ways this box cannot rehearse with more than one rank)
        comm_stream = torch.cuda.Stream() if rcomm is not None else None
        gathered = [None] * depth

        def launch(j):
            k = j % depth
            with torch.cuda.stream(streams[k]):
                if gathered[k] is not None:
                    streams[k].wait_event(gathered[k])      # the previous gather of this plan's window points has read them
                plans[k].run(d_pts, d_sc)
                if rcomm is not None:
                    p, nc, cl = C.c_void_p(), C.c_uint64(), C.c_uint64()
                    ffi.check(L.gm_msm_window_points(plans[k].h, C.byref(p), C.byref(nc), C.byref(cl)))
                    ran = torch.cuda.Event()
                    ran.record(streams[k])
            if rcomm is not None:
                comm_stream.wait_event(ran)
                with torch.cuda.stream(comm_stream):
                    rcomm.all_gather_dev(p, recv[k], ncols * wpr * 32)      # ncclAllGather, asynchronous
                    gathered[k] = torch.cuda.Event()
                    gathered[k].record(comm_stream)

        def finish(j):
            with torch.cuda.stream(streams[j % depth]):
                if rcomm is not None:
                    gathered[j % depth].synchronize()                                                  # step j's kernels and its gather
                    g = recv[j % depth].cpu().numpy().view(np.uint64).reshape(world, ncols, wpr, 4)
                    raw_ = np.ascontiguousarray(np.transpose(g, (1, 0, 2, 3)).reshape(ncols, world * wpr, 4))
                elif world > 1:
                    p, nc, cl = C.c_void_p(), C.c_uint64(), C.c_uint64()
                    ffi.check(L.gm_msm_window_points(plans[j % depth].h, C.byref(p), C.byref(nc), C.byref(cl)))
                    mine = torch.empty((ncols, wpr, 4), dtype=torch.int64, device="cuda")
                    ffi.check(L.gm_memcpy_d2d(C.c_void_p(mine.data_ptr()), p, ncols * wpr * 32, harness.cur_stream()))
                    raw_ = gdist.gather_window_points(dist, mine.to(xdev), world)
                else:
                    raw_ = plans[j % depth].window_points_raw()
                ffi.check(L.gm_msm_profile_read(plans[j % depth].h, prof, 7))
            dom_ms.append(prof[4])
            return harness.combine_host(raw_, d_log), raw_
        for j in range(depth):
            launch(j)
            finish(j)
        for pl in plans[1:]:
            ffi.check(L.gm_msm_profile(pl.h, 1))
        sync_all()
        t0 = time.perf_counter()
        for j in range(steps):
            launch(j)
            if j >= depth - 1:
                result, raw = finish(j - depth + 1)
        for j in range(max(steps - depth + 1, 0), steps):
            result, raw = finish(j)
        sync_all()
        dt_local = time.perf_counter() - t0
        dt = max_over_ranks(dt_local)
        per_rank_ms = [round(v / steps * 1e3, 4) for v in gather_floats(dt_local)]
        for pl in plans[1:]:
            pl.close()
        del plans, recv
        # stage breakdown (one extra, untimed pass)
        ffi.check(L.gm_msm_profile(plan.h, 2))
        step()
        ffi.check(L.gm_msm_profile_read(plan.h, prof, 7))
        # ("add_level0" / "add_levels_ge1" keep their names: with levels 0 and 1 fused -- the default, see roofline.kernel -- the first
        # is the fused launch and the second the levels from 2 on)
        stages = dict(zip(["digits", "histogram", "chunk_scan_offsets", "scatter", "add_level0", "add_levels_ge1", "triangle"],
                          [round(float(v), 4) for v in prof]))
        ffi.check(L.gm_msm_profile(plan.h, 0))
        ms_per_step = dt / steps * 1e3
        dom = float(np.mean(dom_ms)) if dom_ms and dom_ms[0] > 0 else None
        # algorithmic bytes of one k_add_level0 launch: per output cell 2 gathered affine points (2 x 64 B), 2 cell indices
        # (2 x 4 B), one projective point written (96 B); cells = windows * N / 2.  Field multiplications: 8 per level-0 add,
        # 12 per add above (shared sub-products; the layer-by-layer evaluation needs 9 / 13).
        cells0 = wpr * n // 2
        fr_mul_step = cells0 * 8 + (cells0 - (wpr << d_log)) * 12 if cells0 > (wpr << d_log) else cells0 * 8
        fz = C.c_int32(0)
        ffi.check(L.gm_msm_run_info(plan.h, C.byref(fz)))
        fused = bool(fz.value)
        try:
            cells = (C.c_uint64 * (x_log + 1))()
            ffi.check(L.gm_msm_level_cells(plan.h, cells, x_log + 1, harness.cur_stream()))
            adds = [int(cells[l]) // 2 for l in range(1, x_log)]     # adds[0] = additions of level 1, ...
        except Exception:
            adds = []
        if fused and adds:
            # k_add_level01: bintree levels 0 and 1 in one launch.  Algorithmic bytes per launch (SURVEY 8(d) units): every level-0 addition
            # gathers two affine points (2 x 64 B) and two cell indices (2 x 4 B); the level-0 cells never exist; every level-1 cell is one
            # projective point written (96 B; 108 as stored, in the 9 x 29 cell form).  8 field multiplications per level-0 addition, 12 per
            # level-1 addition.
            dom_name = "k_add_level01 (bintree levels 0 + 1 fused)"
            alg_bytes = cells0 * (128 + 8) + int(cells[2]) * 96
            fr_mul0 = cells0 * 8 + adds[0] * 12
            first_ge = 2
            traffic_dom = PMC_TRAFFIC_SC_B.get("k_add_level01")
            traffic_rest = PMC_TRAFFIC_SC_B.get("k_add_levels_ge2_per_step")
        else:
            # algorithmic bytes of one k_add_level0 launch: per output cell 2 gathered affine points (2 x 64 B), 2 cell indices
            # (2 x 4 B), one projective point written (96 B); cells = windows * N / 2.  Field multiplications: 8 per level-0 add,
            # 12 per add above (shared sub-products; the layer-by-layer evaluation needs 9 / 13).
            dom_name = "k_add_level0"
            alg_bytes = cells0 * (128 + 8 + 96)
            fr_mul0 = cells0 * 8
            first_ge = 1
            traffic_dom = PMC_TRAFFIC_MSM_B
            traffic_rest = PMC_TRAFFIC_GE1_B
        roofline = None
        if dom:
            ach = alg_bytes / (dom * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": traffic_dom if (x_log, d_log, nbits, wpr) == (20, 8, 256, 32) else None,
                        "traffic_note": "PMC bytes per launch = (FETCH_SIZE x %s + WRITE_SIZE) x 1024: this kernel's reads are 64-byte gathers, for which "
                                        "FETCH_SIZE reads the bytes exactly (profiles/r04/fetch_size_gather_factors.json: factor 1.00-1.06; the x 2 of "
                                        "the guide is for 16 B / lane coalesced streams)" % PMC_FETCH_FACTOR_GATHER,
                        "avg_launch_ms": round(dom, 4), "algorithmic_bytes_per_launch": alg_bytes, "fr_mul_per_launch": fr_mul0,
                        "fr_mul_per_s": round(fr_mul0 / (dom * 1e-3), 1),
                        "valu_frac_of_measured_ceiling": round(fr_mul0 / (dom * 1e-3) / FR9_MUL_CEILING, 3)}
            # In the timed loop the kernel shares the chip with the late, latency-bound levels of the previous step (other
            # stream): that overlap shortens the step and lengthens this launch.  The same launch with the chip to itself
            # (the untimed stage-breakdown pass below):
            alone = stages["add_level0"]
            if alone > 0:
                roofline["unoverlapped"] = {"launch_ms": alone, "achieved": round(alg_bytes / (alone * 1e-3) / 1e9, 1),
                                            "frac": round(alg_bytes / (alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                            "fr_mul_per_s": round(fr_mul0 / (alone * 1e-3), 1),
                                            "valu_frac_of_measured_ceiling": round(fr_mul0 / (alone * 1e-3) / FR9_MUL_CEILING, 3)}
        # the level kernels above (k_add_level on the flat levels, k_add_tail on the late ones), summed over their launches of one step.
        # Algorithmic bytes per addition as SURVEY 8(d): two projective points read (2 x 96 B), one written (96 B); 12 field
        # multiplications.  Exact pair counts from the row layouts of the run.
        roofline_ge1 = None
        try:
            rest = adds[first_ge - 1:]
            t_ge1 = stages["add_levels_ge1"]
            if t_ge1 > 0 and rest:
                tot_adds = sum(rest)
                b_ge1 = tot_adds * 288
                m_ge1 = tot_adds * 12
                ach1 = b_ge1 / (t_ge1 * 1e-3) / 1e9
                roofline_ge1 = {"bound": "hbm", "kernel": "k_add_level (levels %d..) + k_add_tail (the late levels, to %d)" % (first_ge, x_log - 1),
                                "achieved": round(ach1, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach1 / HBM_PEAK_GBS, 4),
                                "traffic": traffic_rest if (x_log, d_log, nbits, wpr) == (20, 8, 256, 32) else None,
                                "ms_per_step": round(t_ge1, 4), "additions_per_step": tot_adds, "algorithmic_bytes_per_step": b_ge1,
                                "fr_mul_per_step": m_ge1, "fr_mul_per_s": round(m_ge1 / (t_ge1 * 1e-3), 1),
                                "valu_frac_of_measured_ceiling": round(m_ge1 / (t_ge1 * 1e-3) / FR9_MUL_CEILING, 3),
                                "largest_levels_additions": rest[:4],
                                "note": "unoverlapped (the untimed stage-breakdown pass); flat launches up to level x - d - 3, one k_add_tail for the rest"}
        except Exception as e:
            roofline_ge1 = {"error": repr(e)[:200]}
        res = {"x_logsize": x_log, "value": round(n * steps / dt, 1), "ms_per_step": round(ms_per_step, 4), "roofline_levels_ge1": roofline_ge1,
               "per_rank_ms_per_step": per_rank_ms, "pipeline_depth": depth, "roofline": roofline,
               "stage_ms": stages, "result_x": hex(result[0]),
               # SURVEY 8(d)'s whole-MSM unit: 96 B of compulsory HBM traffic per point (64 B point + 32 B scalar)
               "whole_msm": {"algorithmic_bytes_per_point": 96, "achieved_GBps": round(96 * n / (ms_per_step * 1e-3) / 1e9, 2),
                             "frac_of_hbm_peak": round(96 * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "fr_mul_per_step_per_gpu": fr_mul_step,
                             "fr_mul_per_s_per_gpu": round(fr_mul_step / (ms_per_step * 1e-3), 1),
                             "valu_frac_of_measured_ceiling": round(fr_mul_step / (ms_per_step * 1e-3) / FR9_MUL_CEILING, 3),
                             "bound": "integer VALU (8-12 Fr multiplications per bucket add; MFMA not applicable)"}}
        if bcast_ms is not None:
            res["operand_broadcast_ms"] = round(bcast_ms, 2)
            res["operand_bytes"] = n * 96
        if keep:
            return res, plan, d_pts, d_sc, sc, raw
        plan.close()
        del d_pts, d_sc
        L.gm_release_cached_memory()
        torch.cuda.empty_cache()
        return res

    x_main = args.x_logsize if args.x_logsize is not None else 20
    main_res, plan, d_pts, d_sc, sc, raw = msm_leg(x_main, args.steps, args.warmup, keep=True)
    x_log, n = x_main, 1 << x_main

    out = {
        "metric": "msm_points_per_sec", "value": main_res["value"], "unit": "points/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
        "dtype": "u32 limbs, u64 column accumulators (BLS12-381 Fr in Montgomery form: 9 x 29-bit limbs in registers, 8 x 32 in memory)", "data": "synthetic",
        "config": {"workload": "pippenger_msm x_logsize=%d d_logsize=%d nbits=%d (bandersnatch, %d windows)" % (
            x_log, d_log, nbits, y_size), "x_logsize": x_log, "d_logsize": d_log, "nbits": nbits,
            "windows_per_gpu": wpr, "sharding": "windows" if world > 1 else "none", "transport": transport},
        "roofline": main_res["roofline"], "roofline_levels_ge1": main_res["roofline_levels_ge1"], "stage_ms": main_res["stage_ms"],
        "result_x": main_res["result_x"],
        "whole_msm": main_res["whole_msm"],
    }
    for k in ("operand_broadcast_ms", "operand_bytes", "per_rank_ms_per_step", "pipeline_depth"):
        if k in main_res:
            out[k] = main_res[k]
    if world == 1 and rccl_seen:
        out["rccl_ranks_seen"], out["rccl_distinct_devices"] = rccl_seen["ranks"], rccl_seen["distinct_devices"]   # world-1 rehearsal
    if world > 1:
        out["rccl_ranks_seen"] = rccl_seen["ranks"] if rccl_seen else 0
        out["rccl_distinct_devices"] = rccl_seen["distinct_devices"] if rccl_seen else 0
        # the N = 1 figure of the SAME run: every rank runs the whole unsharded MSM on its own GPU (replicas, no exchange)
        try:
            solo = msm_leg(x_main, args.steps, args.warmup, solo=True)
            out["n1_same_run"] = {"value": solo["value"], "ms_per_step": solo["ms_per_step"], "per_rank_ms_per_step": solo["per_rank_ms_per_step"],
                                  "note": "x_logsize=%d unsharded on every GPU at once; value = one GPU's points/s (slowest rank)" % x_main}
            out["per_gpu_efficiency"] = {"value_strong": round(out["value"] / (world * solo["value"]), 4)}
        except Exception as e:
            out["n1_same_run"] = {"error": repr(e)[:300]}

    # ---- second headline: sumcheck rounds/sec of the image-part prover (triangle + bintree GKR) at the same config
    if world == 1 and not args.no_sumcheck:
        y_log = (y_size - 1).bit_length()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        w = harness.PipWitness(plan, d_pts, y_log)
        torch.cuda.synchronize()
        wit_cold_ms = (time.perf_counter() - t1) * 1e3
        w.close()                                      # the trace buffers go back to the library's pool ...
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        w = harness.PipWitness(plan, d_pts, y_log)     # ... and the second build reuses them (steady state)
        torch.cuda.synchronize()
        wit_ms = (time.perf_counter() - t1) * 1e3
        r_pt, r_evs, tape = claims_for(w, y_log, 7)
        w.prove_image_part(r_pt, r_evs, tape)          # warmup
        reps = 3
        harness.sc_profile(1)                          # HIP events around the large round kernels, inside the timed proofs
        torch.cuda.synchronize()
        prove_dt = 0.0
        for _ in range(reps):
            res = w.prove_image_part(r_pt, r_evs, tape)
            prove_dt += res["call_s"] / reps       # the library call (gm_pip_prove_image_part), without the Python big-int conversions
        rows, _, _ = harness.sc_profile_read()
        harness.sc_profile(2)                          # one extra, untimed proof: algorithmic bytes of EVERY round and fold
        w.prove_image_part(r_pt, r_evs, tape)
        rows2, other_bytes, fold_bytes = harness.sc_profile_read()
        harness.sc_profile(0)
        # The headline runs under a LIVE Fiat-Shamir transcript, as the reference's round loop does (sumcheck.rs:101-123 absorbs every
        # round polynomial and squeezes every challenge, proof_transcript.rs:37-57): the library's merlin clone (gm_merlin_transcript)
        # as the gm_transcript of gm_pip_prove_image_part_tr -- hashing on the round's critical path, no Python in the loop.  The
        # challenge-tape form (pre-drawn challenges, messages returned) is what the parity tests drive; its time is reported beside.
        mt = harness.MerlinTranscript(b"bench-image-part")
        harness.prove_image_part_tr(w, r_pt, r_evs, mt)    # warm
        mt.close()
        merlin_dt = 0.0
        for _ in range(reps):
            mt = harness.MerlinTranscript(b"bench-image-part")
            rm = harness.prove_image_part_tr(w, r_pt, r_evs, mt)
            merlin_dt += rm["call_s"] / reps
            proof_bytes = len(mt.proof())
            mt.close()
        assert rm["rounds"] == res["rounds"]
        out["sumcheck"] = {"metric": "sumcheck_rounds_per_sec", "value": round(res["rounds"] / merlin_dt, 1),
                           "transcript": "merlin (STROBE-128 / Keccak-f[1600], gm_merlin_transcript) inside the round loop: %d proof bytes" % proof_bytes,
                           "rounds": res["rounds"], "prove_ms": round(merlin_dt * 1e3, 2),
                           "challenge_tape_form": {"value": round(res["rounds"] / prove_dt, 1), "prove_ms": round(prove_dt * 1e3, 2),
                                                   "note": "pre-drawn challenges, no hashing on the path (the form the parity tests and the "
                                                           "kernel profiles below use)"},
                           "transcript_us_per_round": round((merlin_dt - prove_dt) * 1e6 / res["rounds"], 2),
                           "witness_build_ms": round(wit_ms, 2), "witness_first_build_ms_incl_allocation": round(wit_cold_ms, 2),
                           "witness_trace_GiB": round(L.gm_pip_witness_bytes(w.h) / 2 ** 30, 2),
                           "workload": "prove image part (triangle + bintree GKR) x_logsize=%d d_logsize=%d nbits=%d" % (
                               x_log, d_log, nbits)}
        if rows:
            kern = []
            for r_ in sorted(rows, key=lambda r_: -r_["total_ms"]):
                per = r_["launches"]
                gbps = r_["alg_bytes"] / (r_["total_ms"] * 1e-3) / 1e9 if r_["total_ms"] > 0 else 0.0
                kern.append({"kernel": r_["kernel"], "launches_per_proof": per // reps, "k_cols": r_["k_cols"],
                             "total_ms_per_proof": round(r_["total_ms"] / reps, 3), "avg_launch_ms": round(r_["total_ms"] / per, 4),
                             "algorithmic_GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBS, 4),
                             "fr_mul_per_s": round(r_["fr_mul"] / (r_["total_ms"] * 1e-3), 1),
                             "valu_frac_of_measured_ceiling": round(r_["fr_mul"] / (r_["total_ms"] * 1e-3) / round_kernel_ceiling(r_["kernel"])[0], 3),
                             "field_form": round_kernel_ceiling(r_["kernel"])[1]})
            dom_r = max(rows, key=lambda r_: r_["total_ms"])
            ach = dom_r["alg_bytes"] / (dom_r["total_ms"] * 1e-3) / 1e9
            out["sumcheck"]["roofline"] = {
                "bound": "hbm", "kernel": dom_r["kernel"], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": PMC_TRAFFIC_SC_B.get(dom_r["kernel"]) if (x_log, d_log, nbits) == (20, 8, 256) else None,
                "avg_launch_ms": round(dom_r["total_ms"] / dom_r["launches"], 4),
                "algorithmic_bytes_per_launch": int(dom_r["alg_bytes"] / dom_r["launches"]),
                "algorithmic_bytes_per_pair": int(64 * dom_r["k_cols"] + 32), "launches_per_proof": dom_r["launches"] // reps,
                "fr_mul_per_s": round(dom_r["fr_mul"] / (dom_r["total_ms"] * 1e-3), 1),
                "valu_frac_of_measured_ceiling": round(dom_r["fr_mul"] / (dom_r["total_ms"] * 1e-3) / round_kernel_ceiling(dom_r["kernel"])[0], 3),
                "field_form": round_kernel_ceiling(dom_r["kernel"])[1],
                "note": "average over the large (> 2^14 pairs) launches of this kernel in the timed proofs, HIP events on the launch stream"}
            if dom_r.get("max_ms_pairs", 0) > 0 and dom_r["max_ms"] > 0:
                # the largest launch on its own (the average above is dominated by the 2^14-2^17-pair launches and their ~15 us of
                # launch + cross-block reduction each)
                bpp = dom_r["alg_bytes"] / dom_r["pairs"]
                mpp = dom_r["fr_mul"] / dom_r["pairs"]
                lg_b = bpp * dom_r["max_ms_pairs"] / (dom_r["max_ms"] * 1e-3) / 1e9
                lg_m = mpp * dom_r["max_ms_pairs"] / (dom_r["max_ms"] * 1e-3)
                out["sumcheck"]["roofline"]["largest_launch"] = {
                    "pairs": int(dom_r["max_ms_pairs"]), "ms": round(dom_r["max_ms"], 4), "achieved": round(lg_b, 1),
                    "frac": round(lg_b / HBM_PEAK_GBS, 4), "fr_mul_per_s": round(lg_m, 1),
                    "valu_frac_of_measured_ceiling": round(lg_m / round_kernel_ceiling(dom_r["kernel"])[0], 3)}
            out["sumcheck"]["large_round_kernels"] = kern
            big_bytes = sum(r_["alg_bytes"] for r_ in rows2)
            tot = big_bytes + other_bytes + fold_bytes
            # SURVEY 8(d)'s unit: a FUSED round + fold reads k L F and writes k (L / 2) F = 48 k L bytes per round -- the round's reads
            # once (+ its eq weights) and only the fold's writes (a third of the fold's own 96 B per cell).  This build does not fuse
            # them (measured: 3 % slower, the round kernels are VALU-bound), so its own traffic is the larger figure below it.
            b8d = big_bytes + other_bytes + fold_bytes / 3.0
            out["sumcheck"]["whole_prover"] = {
                "survey_8d_bytes_48kL": int(b8d), "survey_8d_GBps": round(b8d / merlin_dt / 1e9, 1),
                "survey_8d_frac_of_hbm_peak": round(b8d / merlin_dt / 1e9 / HBM_PEAK_GBS, 4),
                "algorithmic_bytes": int(tot), "large_round_bytes": int(big_bytes), "other_round_bytes": int(other_bytes),
                "fold_bytes": int(fold_bytes), "bytes_weighted_GBps": round(tot / merlin_dt / 1e9, 1),
                "frac_of_hbm_peak": round(tot / merlin_dt / 1e9 / HBM_PEAK_GBS, 4),
                "large_kernels_ms_per_proof": round(sum(r_["total_ms"] for r_ in rows) / reps, 2),
                "note": "64 k B per pair read by a round kernel (+32 eq weight), 96 B per cell and column moved by a fold; small sparse "
                        "rounds counted at their capacity bound"}
        # "prove pushforward" chained on the image part's final claims (pippenger.rs:147-160): logup main phase + combined sumcheck
        pr = np.random.default_rng(9)
        pf_tape = [int.from_bytes(pr.bytes(64), "little") % P for _ in range(4)] + [int.from_bytes(pr.bytes(16), "little")
                                                                                     for _ in range(1200)]
        harness.pushforward_prove(plan, d_pts, y_log, res["point"], res["evs"], pf_tape)     # warmup (pool growth)
        torch.cuda.synchronize()
        pf = harness.pushforward_prove(plan, d_pts, y_log, res["point"], res["evs"], pf_tape)
        pf_dt = pf["call_s"]
        mt = harness.MerlinTranscript(b"bench-pushforward")
        pfm = harness.pushforward_prove_tr(plan, d_pts, y_log, res["point"], res["evs"], mt)
        mt.close()
        pfm_dt = pfm["call_s"]
        assert pfm["rounds"] == pf["rounds"]
        out["sumcheck"]["pushforward"] = {"workload": "prove pushforward (columns + logup main phase + combined sumcheck)",
                                          "ms": round(pfm_dt * 1e3, 2), "rounds": pf["rounds"],
                                          "rounds_per_sec": round(pf["rounds"] / pfm_dt, 1), "transcript": "merlin",
                                          "challenge_tape_form_ms": round(pf_dt * 1e3, 2)}
        out["sumcheck"]["gen2_fr_prover_total"] = {"rounds": res["rounds"] + pf["rounds"],
                                                   "ms": round(merlin_dt * 1e3 + pfm_dt * 1e3, 2), "transcript": "merlin",
                                                   "rounds_per_sec": round((res["rounds"] + pf["rounds"]) / (merlin_dt + pfm_dt), 1)}
        w.close()
        del w
        L.gm_release_cached_memory()

        # ---- the same proof from several host threads at once (a proving service's mode: one thread = one stream, one witness).
        # Half of a proof's wall time is latency-bound small rounds that leave the chip idle; independent proofs fill it.  The
        # headline above stays the single proof; this is whole-device throughput.
        if args.concurrent_provers > 1:
            import threading
            T, reps_c = args.concurrent_provers, 4
            bar = threading.Barrier(T + 1)
            box = {"ok": True}

            def prover_thread(k):
                try:
                    st = torch.cuda.Stream()
                    with torch.cuda.stream(st):
                        pl = harness.MsmPlan(x_log, d_log, y_size)
                        pl.run(d_pts, d_sc)
                        wk = harness.PipWitness(pl, d_pts, y_log)
                        first = wk.prove_image_part(r_pt, r_evs, tape)     # warm
                        bar.wait()
                        for _ in range(reps_c):
                            rr = wk.prove_image_part(r_pt, r_evs, tape)
                            if rr["msgs"] != first["msgs"] or first["msgs"] != res["msgs"]:
                                box["ok"] = False
                        bar.wait()
                        wk.close()
                        pl.close()
                except threading.BrokenBarrierError:
                    box["ok"] = False
                except Exception as e:
                    box["ok"] = False
                    box.setdefault("error", repr(e)[:300])
                    bar.abort()
            ths = [threading.Thread(target=prover_thread, args=(k,)) for k in range(T)]
            for th in ths:
                th.start()
            try:
                bar.wait()
                t0 = time.perf_counter()
                bar.wait()
                wall = time.perf_counter() - t0
            except threading.BrokenBarrierError:
                wall = None
            for th in ths:
                th.join()
            out["sumcheck"]["concurrent_provers"] = (
                {"threads": T, "proofs": T * reps_c, "wall_ms": round(wall * 1e3, 1),
                 "rounds_per_sec": round(T * reps_c * res["rounds"] / wall, 1),
                 "proofs_identical_to_the_single_threaded_one": bool(box["ok"]),
                 "note": "whole-device throughput, %d independent image-part proofs in flight (one host thread, stream and witness "
                         "each); the sumcheck headline is the single proof" % T}
                if wall is not None and box["ok"] else {"threads": T, "error": box.get("error", "proofs differ")})
            L.gm_release_cached_memory()
            torch.cuda.empty_cache()

    # ---- N > 1: the other named shapes in the same line (weak: fixed work per GPU; config D: x_logsize = 24)
    if world > 1 and not args.no_extra_shapes:
        plan.close()
        del d_pts, d_sc
        L.gm_release_cached_memory()
        torch.cuda.empty_cache()
        plan = None
        for key, xl in (("weak", 20 + int(round(math.log2(world)))), ("config_d", 24)):
            try:
                steps_x = max(3, args.steps // (4 if xl >= 24 else 1))
                r_ = msm_leg(xl, steps_x, min(args.warmup, 2))
                r_["steps"] = steps_x
                r_["scaling"] = "weak" if key == "weak" else "BASELINE.json configs[3]"
                r_["windows_per_gpu"] = wpr
                if key == "weak" and "n1_same_run" in out and "value" in out["n1_same_run"]:
                    out.setdefault("per_gpu_efficiency", {})["weak"] = round(r_["value"] / (world * out["n1_same_run"]["value"]), 4)
                if key == "config_d":
                    solo_d = msm_leg(xl, 3, 1, solo=True)
                    r_["n1_same_run"] = {"value": solo_d["value"], "ms_per_step": solo_d["ms_per_step"]}
                    out.setdefault("per_gpu_efficiency", {})["config_d"] = round(r_["value"] / (world * solo_d["value"]), 4)
                out[key] = r_
            except Exception as e:
                out[key] = {"error": repr(e)[:300]}

    # ---- N > 1: the same prover sharded by windows / bucket rows: per-round all-gather of the partial sums
    # It runs LAST and under a watchdog: it has only been rehearsed with ranks sharing the builder's one-GPU box; if it stalls on a
    # real node the MSM line -- the metric -- must still come out.
    if world > 1 and not args.no_sumcheck:
        import threading

        def _emit_without_prover():
            # The timer thread races the main thread, which may be adding keys to `out`: serialise a snapshot, retrying if the dict
            # changed under the copy, and ALWAYS leave through os._exit -- a watchdog that dies of an exception leaves the hang in place.
            # Exit code 3 (not 0): a requested leg did not finish; the line is printed first, so the launcher still forwards it.
            line = None
            try:
                for _ in range(20):
                    try:
                        snap = json.loads(json.dumps(dict(out), default=repr))
                        snap.setdefault("sumcheck", {})
                        if not isinstance(snap["sumcheck"], dict):
                            snap["sumcheck"] = {"partial": snap["sumcheck"]}
                        snap["sumcheck"]["error"] = "the sharded prover leg did not finish within its time box; line emitted by the watchdog"
                        snap["legs_unfinished"] = ["sharded_prover"]
                        line = json.dumps(snap)
                        break
                    except RuntimeError:
                        time.sleep(0.01)
                if line is None:
                    line = json.dumps({"metric": out.get("metric"), "value": out.get("value"), "unit": out.get("unit"),
                                       "n_gpus": out.get("n_gpus"), "legs_unfinished": ["sharded_prover"],
                                       "error": "watchdog could not snapshot the full line"}, default=repr)
                if rank == 0:
                    print(line, flush=True)
            finally:
                os._exit(3)
        watchdog = threading.Timer(float(os.environ.get("GM_BENCH_PROVER_TIMEBOX_S", "240")), _emit_without_prover)
        watchdog.daemon = True
        watchdog.start()
        try:
            L.gm_release_cached_memory()     # the extra shapes (x_logsize 24) leave tens of GiB in the library's and torch's caches
            torch.cuda.empty_cache()
            # GM_BENCH_OWN_STREAM=1: the sharded provers on a stream of their own instead of the default one (an experiment of round 4's
            # end, DESIGN section 6 "An oversubscribed device": one more hardware queue per process made that condition worse)
            torch.cuda.synchronize()
            if os.environ.get("GM_BENCH_OWN_STREAM") == "1":
                torch.cuda.set_stream(torch.cuda.Stream())
            if plan is None:     # the extra shapes released the x_logsize-20 plan and operands
                d_pts, d_sc, sc, _ = make_inputs(x_main)
                plan = harness.MsmPlan(x_log, d_log, y_size, y0, y1)
                plan.run(d_pts, d_sc)
                torch.cuda.synchronize()
            y_log = (y_size - 1).bit_length()
            # The per-round sums (<= 96 B per rank) are wanted on the HOST -- they go into the transcript -- so the ranks of the node
            # exchange them between their host threads through shared memory (gm_comm_shm_*): the device side of a sharded round is
            # then the unsharded one, pre-enqueued folds and the persistent stage kernel included.  north_star says "RCCL reduce over
            # xGMI": the device-side exchange (ncclAllGather + a one-wave sum per round, gm_comm_rccl_as_comm) is measured in the SAME
            # run, after the shared-memory one, so that a real node decides between them with data.  GM_BENCH_PROVER_COMM=shm|rccl|torch
            # restricts the leg to one of them.
            only = os.environ.get("GM_BENCH_PROVER_COMM", "")
            order = [only] if only else (["shm"] + (["rccl"] if rcomm is not None else []))
            exchanges = {}
            out["sumcheck"] = {"metric": "sumcheck_rounds_per_sec", "exchanges": exchanges,
                               "workload": "prove image part (triangle + bintree GKR) x_logsize=%d d_logsize=%d nbits=%d" % (x_log, d_log, nbits)}
            state = {"r_pt": None, "r_evs": None, "tape": None, "best": None}

            def run_exchange(which):
                r_pt, r_evs, tape, best = state["r_pt"], state["r_evs"], state["tape"], state["best"]
                ex = exchanges.setdefault(which, {})
                try:
                    if which == "shm":
                        tok = torch.tensor([int.from_bytes(os.urandom(6), "little")], dtype=torch.int64, device=xdev if backend == "nccl" else "cpu")
                        dist.broadcast(tok, src=0)
                        comm = gdist.ShmComm("/gm-bench-%x" % int(tok.item()), rank, world)
                        ex["transport"] = "host shared memory between the ranks' host threads (gm_comm_shm); bulk moves by HIP IPC pulls"
                    elif which == "rccl" and rcomm is not None:
                        comm = rcomm
                        ex["transport"] = transport + ": ncclAllGather of the round sums on the device + a one-wave sum per round"
                    else:
                        comm = gdist.Comm(dist, rank, world, device=xdev if backend == "nccl" else None)
                        ex["transport"] = "torch.distributed (%s)" % backend
                    stage0 = (C.c_uint64(), C.c_uint64())
                    L.gm_sc_stage_counts(C.byref(stage0[0]), C.byref(stage0[1]))
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    w = harness.PipWitness(plan, d_pts, y_log, comm=comm)
                    torch.cuda.synchronize()
                    wit_ms = (time.perf_counter() - t1) * 1e3
                    if r_pt is None:
                        r_pt, r_evs, tape = claims_for(w, y_log, 7)
                    w.prove_image_part(r_pt, r_evs, tape)      # warmup
                    calls0 = comm.calls
                    sync_all()
                    gdist.shard_clock()
                    res = w.prove_image_part(r_pt, r_evs, tape)
                    clock = gdist.shard_clock()
                    p_dt = max_over_ranks(res["call_s"])
                    sync_all()
                    stage1 = (C.c_uint64(), C.c_uint64())
                    L.gm_sc_stage_counts(C.byref(stage1[0]), C.byref(stage1[1]))
                    ex.update({"value": round(res["rounds"] / p_dt, 1), "rounds": res["rounds"], "prove_ms": round(p_dt * 1e3, 2),
                               "witness_build_ms": round(wit_ms, 2), "exchanges_per_proof": comm.calls - calls0,
                               "prover_exchange_ms_per_round": round(clock["small_gather_ms"] / max(res["rounds"], 1), 5),
                               "time_inside_the_communicator_rank0": clock,
                               "stage_kernel_launches_per_proof": (stage1[0].value - stage0[0].value) // 2,
                               "stage_kernel_launches_left_early": stage1[1].value - stage0[1].value})
                    # the pushforward argument on the same sharding (gm_pushforward_prove_sharded), chained after the image part as in
                    # Pippenger::prove; its tree halves move device to device through gm_comm::pull_dev when the communicator has it
                    try:
                        g_rng = np.random.default_rng(23)
                        pf_tape = [int.from_bytes(g_rng.bytes(64), "little") % P for _ in range(4)] + [int.from_bytes(g_rng.bytes(16), "little") for _ in range(3000)]
                        harness.pushforward_prove(plan, d_pts, y_log, res["point"], res["evs"], pf_tape, comm=comm)      # warmup
                        sync_all()
                        gdist.shard_clock()
                        pf = harness.pushforward_prove(plan, d_pts, y_log, res["point"], res["evs"], pf_tape, comm=comm)
                        pf_dt = max_over_ranks(pf["call_s"])
                        ex["pushforward_sharded"] = {"ms": round(pf_dt * 1e3, 2), "rounds": pf["rounds"], "rounds_per_sec": round(pf["rounds"] / pf_dt, 1),
                                                     "time_inside_the_communicator_rank0": gdist.shard_clock(),
                                                     "redistribution": "gm_comm::pull_dev (HIP IPC, device to device)" if which == "shm" else "host all-gather"}
                    except Exception as e:
                        ex["pushforward_sharded"] = {"error": repr(e)[:300]}
                    w.close()
                    del w
                    if best is None or p_dt < best[0]:
                        best = (p_dt, which, res)
                    if which == "shm":
                        comm.close()
                except Exception as e:
                    ex["error"] = repr(e)[:300]
                state.update(r_pt=r_pt, r_evs=r_evs, tape=tape, best=best)

            # the shared-memory exchange first (rehearsed at world 2, 4, 8); the device-side RCCL exchange -- which has never run with more
            # than one rank anywhere -- LAST, after every other leg, so that a stall there (the watchdog ends the run) costs nothing else
            run_exchange(order[0])
            best = state["best"]
            r_pt, r_evs, tape = state["r_pt"], state["r_evs"], state["tape"]
            if best is None:
                raise RuntimeError("no exchange finished: %s" % {k: v.get("error") for k, v in exchanges.items()})
            p_dt, which, res = best
            out["sumcheck"].update({"value": exchanges[which]["value"], "rounds": res["rounds"], "prove_ms": exchanges[which]["prove_ms"],
                                    "transport": "%s (the faster of %s in this run)" % (exchanges[which]["transport"], order),
                                    "witness_build_ms": exchanges[which]["witness_build_ms"],
                                    "sharding": "bucket rows of %d windows per rank; %d exchanges of <= 96 B per rank per proof" % (
                                        wpr, exchanges[which]["exchanges_per_proof"])})
            # the unsharded prover on every GPU at once, same run: what one GPU does alone
            plan_u = harness.MsmPlan(x_log, d_log, y_size)
            plan_u.run(d_pts, d_sc)
            wu = harness.PipWitness(plan_u, d_pts, y_log)
            wu.prove_image_part(r_pt, r_evs, tape)
            sync_all()
            ru = wu.prove_image_part(r_pt, r_evs, tape)
            u_dt = max_over_ranks(ru["call_s"])
            same = ru["msgs"] == res["msgs"] and ru["evs"] == res["evs"]
            out["sumcheck"]["unsharded_same_run"] = {"prove_ms": round(u_dt * 1e3, 2), "rounds_per_sec": round(ru["rounds"] / u_dt, 1),
                                                     "sharded_messages_identical": bool(same)}
            out.setdefault("per_gpu_efficiency", {})["sharded_prover"] = round(u_dt / p_dt / world, 4)
            wu.close()
            plan_u.close()
            del wu, plan_u
            assert same, "sharded prover messages differ from the unsharded ones"
            # ---- the WHOLE gen-2 proof sharded (gm_pippenger_wg_create_sharded + gm_pippenger_prove_tr; BASELINE.json configs[4]'s code
            # path at this run's x_logsize, commitment_log_multiplicity 0): partial G1 commitments over the key ranges a rank holds, the
            # opening on slices; under the merlin transcript, checked against the unsharded proof's pairing pair
            if not os.environ.get("GM_BENCH_NO_SHARDED_FULL") and (world & (world - 1)) == 0:
                try:
                    tok = torch.tensor([int.from_bytes(os.urandom(6), "little")], dtype=torch.int64, device=xdev if backend == "nccl" else "cpu")
                    dist.broadcast(tok, src=0)
                    comm = gdist.ShmComm("/gm-bench-full-%x" % int(tok.item()), rank, world)
                    nv_f = x_log
                    n_key = (2 << nv_f) - 1
                    tau_f = int.from_bytes(np.random.default_rng(23).bytes(32), "little") % P
                    d_basis = harness.g1_mock_srs(tau_f, n_key, codec.G1_GEN)       # every rank makes the mock SRS itself (200 MB)
                    key = harness.KeyView.minimal(d_basis, x_log, d_log, y_log, 0, rank, world)
                    first, cnt = harness.knuckles_slice_of(nv_f, rank, world)
                    d_inv_s = harness.knuckles_setup_range(2, nv_f, first, cnt)
                    times = []
                    for it in range(2):
                        sync_all()
                        t1 = time.perf_counter()
                        wgs = harness.PippengerWGSharded(plan, d_pts, y_log, 0, key, comm)
                        torch.cuda.synchronize()
                        t_w = max_over_ranks(time.perf_counter() - t1)
                        mt = harness.MerlinTranscript(b"bench-full-sharded")
                        gdist.shard_clock()
                        fm = harness.pippenger_prove_tr(wgs, r_pt, r_evs, d_inv_s, 2, mt)
                        clock_f = gdist.shard_clock()
                        mt.close()
                        t_p = max_over_ranks(fm["call_s"])
                        spans_f = harness.pippenger_last_spans()
                        wgs.close()
                        times.append((t_w, t_p))
                    out["full_gen2_prover_sharded"] = {
                        "workload": "gm_pippenger_wg_create_sharded + gm_pippenger_prove_tr (merlin), x_logsize=%d clm=0 over %d ranks" % (x_log, world),
                        "witness_and_commitments_ms": round(times[-1][0] * 1e3, 2), "prove_ms": round(times[-1][1] * 1e3, 2),
                        "first_call_ms": {"witness_and_commitments": round(times[0][0] * 1e3, 2), "prove": round(times[0][1] * 1e3, 2)},
                        "spans_ms_rank0": spans_f, "time_inside_the_communicator_rank0": clock_f,
                        "key_points_resident_per_rank": int(sum(key.count)), "key_points_total": n_key,
                        "pair_satisfies_A_eq_tau_B": None}
                    from gkr_msm_amd import verifier as VF
                    h0, h1 = VF.kzg_mock_vk(tau_f)
                    out["full_gen2_prover_sharded"]["pair_satisfies_A_eq_tau_B"] = bool(VF.kzg_verify_pair(fm["pair"], h0, h1))
                    comm.close()
                    del d_basis, key
                except Exception as e:
                    out["full_gen2_prover_sharded"] = {"error": repr(e)[:400]}
            for which_late in order[1:]:
                run_exchange(which_late)
                if state["best"] is not None and state["best"][1] == which_late:      # the late exchange was the faster one: it is the headline
                    p_dt, which, res = state["best"]
                    out["sumcheck"].update({"value": exchanges[which]["value"], "prove_ms": exchanges[which]["prove_ms"],
                                            "transport": "%s (the faster of %s in this run)" % (exchanges[which]["transport"], order),
                                            "witness_build_ms": exchanges[which]["witness_build_ms"]})
                    out.setdefault("per_gpu_efficiency", {})["sharded_prover"] = round(u_dt / p_dt / world, 4)
        except Exception as e:  # keep the MSM line even if the sharded prover leg fails on this node
            out.setdefault("sumcheck", {})["error"] = repr(e)[:300]
        watchdog.cancel()
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())

    # ---- gen-1 prover (gkr_msm_simple.rs gkr_msm_prove, Fr part): BASELINE.json configs[2]
    if world == 1 and not args.no_sumcheck and args.gen1_log_points > 0:
        plan.close()
        plan = None
        L.gm_release_cached_memory()
        torch.cuda.empty_cache()
        lp, lb = args.gen1_log_points, 8
        g_rng = np.random.default_rng(11)
        d_bits = torch.from_numpy(g_rng.integers(0, 2, size=(1 << (lp + lb)), dtype=np.uint8)).cuda()
        g_tape = [int.from_bytes(g_rng.bytes(64), "little") % P for _ in range(6000)]
        d_pts_g = d_pts[: (1 << lp) * 8] if lp <= x_log else None
        if d_pts_g is None:
            d_pts_g = harness.dev_empty((1 << lp) * 8)
            ffi.check(L.gm_gen_points(C.c_void_p(d_pts_g.data_ptr()), 1 << lp, SEED, harness.cur_stream()))
        # first call: the library's device-memory pool grows by the trace (the driver hands out recycled HBM at ~40 GiB/s);
        # steady state = the second call, as for a prover that proves more than once
        try:
            # the column commitments gkr_msm_prove writes first (gkr_msm_simple.rs:117-151): 2^7 bit columns through binary_msm
            # (gamma = 4, the reference bench's value) + the point column; the CommitmentKey (bases, binary tables) is setup
            lcols, gamma = 7, 4
            col_size = 1 << (lp + lb - lcols)
            d_cbases = harness.g1_gen_points(col_size, 0x434B)
            d_ctables = harness.g1_prepare_bases(d_cbases, col_size, gamma)
            hb = np.zeros((1 << lcols, 12), dtype=np.uint64)
            hp = np.zeros(12, dtype=np.uint64)

            def commit():
                ffi.check(L.gm_gkr_msm_commit(C.c_void_p(d_pts_g.data_ptr()), C.c_void_p(d_bits.data_ptr()), lp, lb, lcols,
                                              C.c_void_p(d_cbases.data_ptr()), C.c_void_p(d_ctables.data_ptr()), gamma, hb.ctypes.data,
                                              hp.ctypes.data, harness.cur_stream()))
            commit()
            torch.cuda.synchronize()
            g_cold = harness.gkr_msm_prove(d_pts_g, d_bits, lp, lb, g_tape, msgs_cap=1 << 16)["call_s"]
            torch.cuda.synchronize()
            # the same first call with the memory RESERVED at set-up time (gm_reserve): the pool is emptied back to the driver, the reserve
            # taken (timed: this is where the driver's allocation rate is paid -- once, next to the SRS load), and the next call is
            # again the first one to need its ~120 GiB of trace and workspace
            g_after_reserve = reserve_s = None
            if lp == 20 and not os.environ.get("GM_BENCH_NO_RESERVE"):
                L.gm_release_cached_memory()
                torch.cuda.empty_cache()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rc_res = L.gm_reserve(C.c_uint64(150 << 30))
                torch.cuda.synchronize()
                reserve_s = time.perf_counter() - t1
                if rc_res == 0:
                    g_after_reserve = harness.gkr_msm_prove(d_pts_g, d_bits, lp, lb, g_tape, msgs_cap=1 << 16)["call_s"]
                    torch.cuda.synchronize()
            t1 = time.perf_counter()
            commit()
            torch.cuda.synchronize()
            c_dt = time.perf_counter() - t1
            harness.sc_profile(1)
            g1 = harness.gkr_msm_prove(d_pts_g, d_bits, lp, lb, g_tape, msgs_cap=1 << 16)
            g_rows, _, _ = harness.sc_profile_read()
            harness.sc_profile(0)
            g_tape_dt = g1["call_s"]
            # the timed figure: under the live merlin transcript (gm_gkr_msm_prove_tr), as the reference's prover runs
            mt = harness.MerlinTranscript(b"bench-gen1")
            g1m = harness.gkr_msm_prove_tr(d_pts_g, d_bits, lp, lb, mt)
            mt.close()
            g_dt = g1m["call_s"]
            assert g1m["rounds"] == g1["rounds"]
            harness.sc_profile(2)                      # one more proof: algorithmic bytes of every round and fold
            harness.gkr_msm_prove(d_pts_g, d_bits, lp, lb, g_tape, msgs_cap=1 << 16)
            g_rows2, g_other, g_fold = harness.sc_profile_read()
            harness.sc_profile(0)
            out["gen1"] = {"workload": "gkr_msm_prove log_num_points=%d log_num_scalar_bits=%d: column commitments (2^%d bit columns, "
                                       "binary_msm gamma=%d, + the point column) + witness + prover" % (lp, lb, lcols, gamma),
                           "transcript": "merlin (gm_gkr_msm_prove_tr)",
                           "total_ms": round((g_dt + c_dt) * 1e3, 2), "commit_ms": round(c_dt * 1e3, 2), "gkr_ms": round(g_dt * 1e3, 2),
                           "gkr_ms_challenge_tape_form": round(g_tape_dt * 1e3, 2),
                           "first_call_ms_incl_allocation": round(g_cold * 1e3, 2),
                           "first_call_ms_after_gm_reserve": round(g_after_reserve * 1e3, 2) if g_after_reserve else None,
                           "gm_reserve_150GiB_ms": round(reserve_s * 1e3, 1) if reserve_s else None,
                           "witness_ms": round(g1["witness_ms"], 2), "rounds": g1["rounds"],
                           "rounds_per_sec": round(g1["rounds"] / max(g_dt - g1["witness_ms"] * 1e-3, 1e-9), 1),
                           "points_per_sec": round((1 << lp) / (g_dt + c_dt), 1)}
            # roofline of the path north_star names literally (gkr_msm_simple.rs:200-317): the dominant dense round kernel (HIP
            # events inside the proof), the rounds + folds of the whole proof by bytes, and the witness maps (map_over_poly,
            # utils.rs:70-86: (k + m) F per element and layer) over the witness build's HIP-event time
            n0g = 1 << (lp + lb)
            wit_bytes = 3 * 32 * n0g + (3 + 2) * 32 * n0g + (2 * n0g + 4 * (n0g // 2)) * 32      # base, pt_bit_choice, split(2)
            ng = n0g // 2
            wit_bytes += ((4 + 3) + (3 + 3) + (3 + 3)) * 32 * ng                                   # affine l1, l2, l3
            for _ in range(lp - 1):
                wit_bytes += (3 * ng + 6 * (ng // 2)) * 32                                         # split(3)
                ng //= 2
                wit_bytes += ((6 + 4) + (4 + 4) + (4 + 3)) * 32 * ng                               # projective l1, l2, l3
            if g_rows:
                dg = max(g_rows, key=lambda r_: r_["total_ms"])
                achg = dg["alg_bytes"] / (dg["total_ms"] * 1e-3) / 1e9
                bppg, mppg = dg["alg_bytes"] / dg["pairs"], dg["fr_mul"] / dg["pairs"]
                out["gen1"]["roofline"] = {
                    "bound": "hbm", "kernel": dg["kernel"], "achieved": round(achg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achg / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(dg["total_ms"] / dg["launches"], 4),
                    "launches_per_proof": dg["launches"], "algorithmic_bytes_per_pair": int(bppg),
                    "fr_mul_per_s": round(dg["fr_mul"] / (dg["total_ms"] * 1e-3), 1),
                    "valu_frac_of_measured_ceiling": round(dg["fr_mul"] / (dg["total_ms"] * 1e-3) / round_kernel_ceiling(dg["kernel"])[0], 3),
                    "largest_launch": {"pairs": int(dg["max_ms_pairs"]), "ms": round(dg["max_ms"], 4),
                                       "achieved": round(bppg * dg["max_ms_pairs"] / (dg["max_ms"] * 1e-3) / 1e9, 1),
                                       "frac": round(bppg * dg["max_ms_pairs"] / (dg["max_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "valu_frac_of_measured_ceiling": round(mppg * dg["max_ms_pairs"] / (dg["max_ms"] * 1e-3) /
                                                                              round_kernel_ceiling(dg["kernel"])[0], 3)} if dg["max_ms"] > 0 else None,
                    "large_round_kernels": [{"kernel": r_["kernel"], "launches": r_["launches"], "total_ms": round(r_["total_ms"], 3),
                                             "algorithmic_GBps": round(r_["alg_bytes"] / (r_["total_ms"] * 1e-3) / 1e9, 1),
                                             "valu_frac_of_measured_ceiling": round(r_["fr_mul"] / (r_["total_ms"] * 1e-3) /
                                                                                    round_kernel_ceiling(r_["kernel"])[0], 3)}
                                            for r_ in sorted(g_rows, key=lambda r_: -r_["total_ms"])[:6]],
                    "note": "dominant dense round kernel of the gen-1 proof (HIP events on the launch stream, challenge-tape proof)"}
                g_big = sum(r_["alg_bytes"] for r_ in g_rows2)
                prove_only = max(g_dt - g1["witness_ms"] * 1e-3, 1e-9)
                out["gen1"]["whole_prover"] = {
                    "survey_8d_bytes_48kL": int(g_big + g_other + g_fold / 3.0),
                    "survey_8d_frac_of_hbm_peak": round((g_big + g_other + g_fold / 3.0) / prove_only / 1e9 / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes": int(g_big + g_other + g_fold), "fold_bytes": int(g_fold),
                    "bytes_weighted_GBps": round((g_big + g_other + g_fold) / prove_only / 1e9, 1),
                    "frac_of_hbm_peak": round((g_big + g_other + g_fold) / prove_only / 1e9 / HBM_PEAK_GBS, 4),
                    "large_kernels_ms": round(sum(r_["total_ms"] for r_ in g_rows), 2), "prove_ms_without_witness": round(prove_only * 1e3, 2)}
                out["gen1"]["witness_maps"] = {
                    "algorithmic_bytes": int(wit_bytes), "ms": round(g1["witness_ms"], 2),
                    "GBps": round(wit_bytes / (g1["witness_ms"] * 1e-3) / 1e9, 1),
                    "frac_of_hbm_peak": round(wit_bytes / (g1["witness_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "k_dense_map / k_dense_map_split launches of the forward pass: (k + m) x 32 B per element and layer"}
            del d_cbases, d_ctables
        except Exception as e:
            out["gen1"] = {"error": repr(e)[:300]}
        del d_bits
        L.gm_release_cached_memory()
        L.gm_unreserve()
        torch.cuda.empty_cache()

    # ---- BLS12-381 G1 side: KZG-commit-shaped MSM and the outer buckets of PushForwardState::new
    if world == 1 and args.g1_log_points > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        ng = 1 << args.g1_log_points
        d_srs = harness.g1_gen_points(ng, 0x53525331)
        g_sc = np.random.default_rng(21).integers(0, 2 ** 64, size=(ng, 4), dtype=np.uint64)
        g_sc[:, 3] &= np.uint64((1 << 62) - 1)            # < 2^254 < r: canonical Fr bigints
        d_gsc = harness.to_dev(g_sc)
        harness.g1_msm(d_srs, d_gsc, ng)                   # warmup (scratch allocation)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            g_res = harness.g1_msm(d_srs, d_gsc, ng)
        g_dt = (time.perf_counter() - t1) / reps
        out["g1"] = {"msm": {"workload": "G1 MSM (KzgProvingKey::commit shape) 2^%d affine bases x 255-bit scalars" % args.g1_log_points,
                             "ms": round(g_dt * 1e3, 3), "points_per_sec": round(ng / g_dt, 1)}}
        # the same MSM over a registered proving key (fixed-base tables: 16 windows of precomputed multiples, one bucket set)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        harness.g1_fixed_base_register(d_srs, ng)
        torch.cuda.synchronize()
        fb_setup = time.perf_counter() - t1
        assert harness.g1_msm(d_srs, d_gsc, ng) == g_res, "fixed-base MSM differs from the per-window MSM"
        t1 = time.perf_counter()
        for _ in range(reps):
            harness.g1_msm(d_srs, d_gsc, ng)
        fb_dt = (time.perf_counter() - t1) / reps
        harness.g1_fixed_base_release(d_srs)
        out["g1"]["msm"]["fixed_base"] = {"ms": round(fb_dt * 1e3, 3), "points_per_sec": round(ng / fb_dt, 1),
                                          "table_GiB": round(ng * 16 * 96 / 2 ** 30, 2), "table_build_ms": round(fb_setup * 1e3, 1),
                                          "parity": "same group element as the per-window MSM"}
        if not args.no_sumcheck:
            # outer buckets of the bench shape: re-run the bucketing (the plan was closed by the gen-1 leg)
            plan_o = harness.MsmPlan(x_log, d_log, y_size)
            plan_o.run(d_pts, d_sc)
            cap = max(n >> 3, 64)
            harness.msm_g1_outer(plan_o, d_srs[: 12 * n], 0, cap)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            o_res = harness.msm_g1_outer(plan_o, d_srs[: 12 * n], 0, cap)
            o_dt = time.perf_counter() - t1
            out["g1"]["outer_buckets"] = {
                "workload": "d_outer + c_outer accumulation + c/d commitments, x_logsize=%d d_logsize=%d y_size=%d clm=0" % (
                    x_log, d_log, y_size), "ms": round(o_dt * 1e3, 2), "g1_adds": 2 * n * y_size,
                "g1_adds_per_sec": round(2 * n * y_size / o_dt, 1)}
            del o_res
            plan_o.close()
        if not args.no_cpu_baseline:
            import oracle_ffi as O
            threads = args.cpu_threads or usable_cpus()
            lg2 = min(args.cpu_g1_log_points, args.g1_log_points)
            n2 = 1 << lg2
            srs_h = harness.to_host(d_srs[: 12 * n2]).reshape(n2, 12)
            t1 = time.perf_counter()
            c_res = O.g1_msm_affine(srs_h, g_sc[:n2], threads)
            c_dt = time.perf_counter() - t1
            g_same = harness.g1_msm(d_srs, d_gsc, n2)
            t1 = time.perf_counter()
            g_same = harness.g1_msm(d_srs, d_gsc, n2)
            gs_dt = time.perf_counter() - t1
            assert codec.g1_aff_from_limbs(c_res)[0] == g_same, "GPU G1 MSM differs from the CPU oracle"
            out["g1"]["msm"]["cpu_baseline"] = {
                "value": round(n2 / c_dt, 1), "unit": "points/s", "cores": threads, "kind": "port",
                "sample": "msm_bigint_wnaf_nonaff, first 2^%d bases (GPU figure above is at 2^%d), %.2f s" % (lg2, args.g1_log_points, c_dt),
                "gpu_same_sample_points_per_sec": round(n2 / gs_dt, 1), "parity": "same group element"}
            if "outer_buckets" in out["g1"]:
                xo = min(args.cpu_g1_outer_xlog, x_log)
                no = 1 << xo
                plan_s = harness.MsmPlan(xo, d_log, y_size)
                plan_s.run(harness.to_dev(harness.to_host(d_pts).reshape(n, 8)[:no]), harness.to_dev(sc[:no]))
                dg, ct, _ = plan_s.digits_counter_rowlen()
                t1 = time.perf_counter()
                c_o = O.g1_pushforward_outer(dg, ct, srs_h[:no] if no <= n2 else harness.to_host(d_srs[: 12 * no]).reshape(no, 12),
                                             xo, d_log, y_size, 0, threads)
                co_dt = time.perf_counter() - t1
                harness.msm_g1_outer(plan_s, d_srs[: 12 * no], 0, no)
                t1 = time.perf_counter()
                g_o = harness.msm_g1_outer(plan_s, d_srs[: 12 * no], 0, no)
                go_dt = time.perf_counter() - t1
                same = (g_o[3] == codec.g1_aff_from_limbs(c_o["d_comm"]) and g_o[4] == codec.g1_aff_from_limbs(c_o["c_comm"])
                        and harness.g1_read_jac(g_o[0]) == codec.g1_jac_from_limbs(c_o["d_outer"]))
                assert same, "GPU outer buckets differ from the CPU oracle"
                out["g1"]["outer_buckets"]["cpu_baseline"] = {
                    "value": round(2 * no * y_size / co_dt, 1), "unit": "g1_adds/s", "cores": threads, "kind": "port",
                    "sample": "same accumulation at x_logsize=%d (GPU figure above is at %d): %.2f s" % (xo, x_log, co_dt),
                    "gpu_same_sample_g1_adds_per_sec": round(2 * no * y_size / go_dt, 1),
                    "parity": "same group elements (d_outer buckets, c_comm, d_comm)"}
                plan_s.close()
        # ---- the whole gen-2 prover at the bench shape: PippengerWG::new + Pippenger::prove (pippenger.rs:37-70, 118-290)
        if not args.no_sumcheck and args.g1_log_points >= x_log + 1:
            y_log = (y_size - 1).bit_length()
            plan_f = harness.MsmPlan(x_log, d_log, y_size)
            d_inv = harness.knuckles_setup(2, x_log)
            # a real (mock-setup) SRS: powers of a known tau, so that the proof's pairing equation can be checked in G1 below
            from gkr_msm_amd import verifier as VF
            tau_f = int.from_bytes(np.random.default_rng(23).bytes(32), "little") % P
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            d_basis = harness.g1_mock_srs(tau_f, (2 << x_log) - 1, codec.G1_GEN)
            torch.cuda.synchronize()
            srs_s = time.perf_counter() - t1
            t1 = time.perf_counter()
            harness.g1_fixed_base_register(d_basis, (2 << x_log) - 1)   # part of the proving-key setup, like the SRS itself
            torch.cuda.synchronize()
            fbk_s = time.perf_counter() - t1
            full = None
            for it in range(2):                               # second pass = warm memory pool
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                plan_f.run(d_pts, d_sc)
                torch.cuda.synchronize()
                t_b = time.perf_counter() - t1
                t1 = time.perf_counter()
                wgf = harness.PippengerWG(plan_f, d_pts, y_log, 0, d_basis)
                torch.cuda.synchronize()
                t_w = time.perf_counter() - t1
                outs = wgf.dense_output()
                pr2 = np.random.default_rng(17)
                r_f = [int.from_bytes(pr2.bytes(64), "little") % P for _ in range(y_log)]

                def ev_f(poly):
                    cur = list(poly)
                    for f in reversed(r_f):
                        cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
                    return cur[0]
                evs_f = [ev_f(o) for o in outs]
                tape_f = [int.from_bytes(pr2.bytes(16), "little") for _ in range(6000)]
                t1 = time.perf_counter()
                full = wgf.prove(r_f, evs_f, d_inv, 2, tape_f)
                t_p_tape = time.perf_counter() - t1
                spans_tape = harness.pippenger_last_spans()
                # the timed figure: the same proof under the live merlin transcript (gm_pippenger_prove_tr)
                mt = harness.MerlinTranscript(b"bench-full")
                fm = harness.pippenger_prove_tr(wgf, r_f, evs_f, d_inv, 2, mt)
                proof_f = bytes(mt.proof())
                proof_len_f = len(proof_f)
                mt.close()
                t_p = fm["call_s"]
                spans = harness.pippenger_last_spans()
                if it == 0:
                    t_first = (t_b, t_w, t_p_tape)
                wgf.close()
            # the library's verifier (host, Pippenger::verify) on the proof just made, then the real pairing check
            t1 = time.perf_counter()
            ver = VF.pippenger_verify(x_log, d_log, y_size, y_log, 0, r_f, evs_f, codec.G1_GEN, 2, full["msgs"], full["points"],
                                      tape_f[: full["tape_used"]])
            t_v = time.perf_counter() - t1
            assert ver["pair"] == full["pair"], "verifier and prover disagree on the deferred pairing pair"
            t1 = time.perf_counter()
            h0, h1 = VF.kzg_mock_vk(tau_f)
            assert VF.kzg_verify_pair(ver["pair"], h0, h1), "the full-size proof does not satisfy the pairing equation"
            t_pair = time.perf_counter() - t1
            out["full_gen2_prover"] = {
                "workload": "PippengerWG::new + Pippenger::prove, x_logsize=%d d_logsize=%d nbits=%d clm=0" % (x_log, d_log, nbits),
                "bucketing_and_msm_ms": round(t_b * 1e3, 2), "witness_and_commitments_ms": round(t_w * 1e3, 2),
                "prove_ms": round(t_p * 1e3, 2), "transcript": "merlin (gm_pippenger_prove_tr), %d proof bytes" % proof_len_f,
                "prove_ms_challenge_tape_form": round(t_p_tape * 1e3, 2),
                "spans_ms": dict(spans, compute_buckets_and_commit_phase_1_ms=round((t_b + t_w) * 1e3, 2),
                                 note="the reference's tracing spans (pippenger.rs:121-159); open = opening witnesses + MultiOpenReduction + Knuckles"),
                "spans_ms_challenge_tape_form": spans_tape,
                "first_call_ms_incl_allocation": {"bucketing_and_msm": round(t_first[0] * 1e3, 2), "witness_and_commitments": round(t_first[1] * 1e3, 2),
                                                  "prove": round(t_first[2] * 1e3, 2)},
                "total_ms": round((t_b + t_w + t_p) * 1e3, 2), "sumcheck_rounds": full["rounds"],
                "transcript_scalars": len(full["msgs"]), "transcript_points": len(full["points"]),
                "proofs_per_sec": round(1.0 / (t_b + t_w + t_p), 3),
                "srs": "KzgProvingKey::mock_setup, 2^%d - 1 powers of tau generated on the GPU in %.2f s; fixed-base tables of the "
                       "key (3.2 GB) in %.2f s" % (x_log + 1, srs_s, fbk_s),
                "verified": "accepted by gm_pippenger_verify (host, %.0f ms incl. marshalling) and e(A, [1]_2) == e(B, [tau]_2) "
                            "(gm_kzg_verify_pair, %.0f ms incl. the mock verifying key)" % (t_v * 1e3, t_pair * 1e3)}
            # ---- whole proofs from several host threads at once (a proving service's mode; the G1 engine takes one call at a time
            # per device, so what overlaps is one proof's G1 work with the others' latency-bound sumcheck rounds)
            if args.concurrent_provers > 1:
                import threading
                T, reps_c = args.concurrent_provers, 2
                bar = threading.Barrier(T + 1)
                box = {"ok": True}

                def full_thread(k):
                    try:
                        with torch.cuda.stream(torch.cuda.Stream()):
                            pl = harness.MsmPlan(x_log, d_log, y_size)

                            def one():
                                pl.run(d_pts, d_sc)
                                wk = harness.PippengerWG(pl, d_pts, y_log, 0, d_basis)
                                mk = harness.MerlinTranscript(b"bench-full")
                                rk = harness.pippenger_prove_tr(wk, r_f, evs_f, d_inv, 2, mk)
                                same = rk["pair"] == fm["pair"] and bytes(mk.proof()) == proof_f
                                mk.close()
                                wk.close()
                                return same
                            one()                                  # warm this thread's share of the memory pool
                            bar.wait()
                            for _ in range(reps_c):
                                if not one():
                                    box["ok"] = False
                            torch.cuda.current_stream().synchronize()
                            bar.wait()
                            pl.close()
                    except threading.BrokenBarrierError:
                        box["ok"] = False
                    except Exception as e:
                        box["ok"] = False
                        box.setdefault("error", repr(e)[:300])
                        bar.abort()
                ths = [threading.Thread(target=full_thread, args=(k,)) for k in range(T)]
                for th in ths:
                    th.start()
                try:
                    bar.wait()
                    t1 = time.perf_counter()
                    bar.wait()
                    wall_c = time.perf_counter() - t1
                except threading.BrokenBarrierError:
                    wall_c = None
                for th in ths:
                    th.join()
                out["full_gen2_prover"]["concurrent_provers"] = (
                    {"threads": T, "proofs": T * reps_c, "wall_ms": round(wall_c * 1e3, 1),
                     "proofs_per_sec": round(T * reps_c / wall_c, 3), "ms_per_proof": round(wall_c * 1e3 / (T * reps_c), 2),
                     "proofs_identical_to_the_single_threaded_one": True,
                     "note": "whole-device throughput: %d independent whole proofs (PippengerWG::new + prove under merlin) in flight, one "
                             "host thread, stream, plan and witness each, one proving key; the figures above are the single proof" % T}
                    if wall_c is not None and box["ok"] else {"threads": T, "error": box.get("error", "proofs differ")})
            harness.g1_fixed_base_release(d_basis)
            plan_f.close()
        del d_srs, d_gsc
        ffi.check(L.gm_g1_release_scratch())

    # ---- CPU baseline + in-run parity (rank 0, N = 1)
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_ffi as O
        L.gm_release_cached_memory()
        torch.cuda.empty_cache()
        threads = args.cpu_threads or usable_cpus()
        msm_threads = min(threads, y_size)      # the MSM port is parallel over windows (pushforward.rs:401), as the reference
        hinfo = host_info()
        pts_h = harness.to_host(d_pts).reshape(n, 8)
        # MSM: the whole workload on the CPU (windows in parallel = the reference's own granularity, pushforward.rs:401 /
        # msm_nonaffine.rs:123: for this leg "reference-faithful" and "fair" coincide)
        xs = min(x_log, 20)
        t1 = time.perf_counter()
        ref = O.msm(pts_h[: 1 << xs], sc[: 1 << xs], xs, d_log, y_size, threads=msm_threads, want_aux=False)
        O.msm_combine(ref["window_cols"], d_log)
        cpu_dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round((1 << xs) / cpu_dt, 1), "unit": "points/s", "cores": msm_threads,
                               "kind": "port", "variant": "fair = reference-faithful (parallel over the %d windows, as the reference)" % y_size,
                               "sample": "the whole workload: same inputs, 2^%d points x %d windows, %.2f s" % (xs, y_size, cpu_dt),
                               "host": hinfo}
        if xs == x_log:
            ok = np.array_equal(raw, ref["window_cols"])
            out["parity"] = "bit-exact vs oracle (window points, %d x %d Fr)" % raw.shape[:2] if ok else "MISMATCH"
            assert ok, "GPU window points differ from the CPU oracle"
        out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        del ref
        if not args.no_sumcheck:
            y_log = (y_size - 1).bit_length()

            def cpu_vs_gpu_prover(xs2, faithful):
                n2 = 1 << xs2
                plan2 = harness.MsmPlan(xs2, d_log, y_size)
                d_pts2, d_sc2 = harness.to_dev(pts_h[:n2]), harness.to_dev(sc[:n2])
                plan2.run(d_pts2, d_sc2)
                w2 = harness.PipWitness(plan2, d_pts2, y_log)
                r_pt, r_evs, tape = claims_for(w2, y_log, 8)
                w2.prove_image_part(r_pt, r_evs, tape)
                g = w2.prove_image_part(r_pt, r_evs, tape)
                gpu_dt = g["call_s"]
                w2.close()
                plan2.close()
                L.gm_release_cached_memory()
                O.lib().or_set_reference_faithful(1 if faithful else 0)
                try:
                    t1 = time.perf_counter()
                    cw = O.PipWitness(pts_h[:n2], sc[:n2], xs2, d_log, y_size, y_log, threads)
                    cpu_wit = time.perf_counter() - t1
                    t1 = time.perf_counter()
                    c = cw.prove_image_part(codec.to_mont_limbs(r_pt), codec.to_mont_limbs(r_evs), codec.ints_to_limbs(tape))
                    cpu_prove = time.perf_counter() - t1
                    cw.close()
                finally:
                    O.lib().or_set_reference_faithful(0)
                same = codec.from_mont_limbs(c["msgs"]) == g["msgs"] and codec.from_mont_limbs(c["evs"]) == g["evs"]
                assert same, "GPU prover messages differ from the CPU oracle"
                return {"value": round(c["rounds"] / cpu_prove, 1), "unit": "rounds/s", "cores": threads, "kind": "port",
                        "variant": "reference-faithful (round sums and vecvec_map_split serial, as dense_eq.rs:121-139 / "
                                   "vecvec_eq.rs:320-361 / vecvec.rs:579-594)" if faithful else "fair (every loop threaded)",
                        "sample": "same prover at x_logsize=%d (%d rounds): cpu witness %.2f s + prove %.2f s" % (
                            xs2, c["rounds"], cpu_wit, cpu_prove),
                        "cpu_witness_s": round(cpu_wit, 3), "cpu_prove_s": round(cpu_prove, 3),
                        "gpu_same_sample_rounds_per_sec": round(g["rounds"] / gpu_dt, 1),
                        "parity": "bit-exact (%d prover messages + final claims)" % len(g["msgs"]), "host": hinfo}
            out["sumcheck"]["cpu_baseline"] = cpu_vs_gpu_prover(min(x_log, args.cpu_sumcheck_xlog), False)
            out["sumcheck"]["cpu_baseline_reference_faithful"] = cpu_vs_gpu_prover(min(x_log, args.cpu_faithful_xlog), True)
            out["sumcheck"]["speedup_vs_cpu_fair"] = round(out["sumcheck"]["value"] / out["sumcheck"]["cpu_baseline"]["value"], 1)
            # the north star's "host-CPU Pippenger + sumcheck wall clock" at config B, CPU fair variant vs GPU
            cb = out["sumcheck"]["cpu_baseline"]
            if min(x_log, args.cpu_sumcheck_xlog) == x_log:
                cpu_wall = cpu_dt + cb["cpu_witness_s"] + cb["cpu_prove_s"]
                gpu_wall = (out["ms_per_step"] + out["sumcheck"]["witness_build_ms"] + out["sumcheck"]["prove_ms"]) * 1e-3
                # the CPU port had `threads` hardware threads of a host with `nproc` (the box's cgroup quota); the bound below scales
                # its time LINEARLY to every hardware thread of the host -- more than any real run would gain (the MSM leg cannot use
                # more threads than windows, memory bandwidth is shared) -- so the speed-up against the whole host is AT LEAST this
                lin = cpu_wall * threads / max(hinfo["nproc"], threads)
                out["cpu_baseline"]["linear_bound_at_nproc"] = {
                    "nproc": hinfo["nproc"], "threads_used": threads, "cpu_wall_s_if_scaling_were_linear": round(lin, 3),
                    "speedup_lower_bound": round(lin / gpu_wall, 1), "target_10x_met_against_the_whole_host": bool(lin / gpu_wall >= 10.0)}
                out["pippenger_plus_sumcheck_wall"] = {"cpu_s": round(cpu_wall, 2), "gpu_s": round(gpu_wall, 4),
                                                       "speedup": round(cpu_wall / gpu_wall, 1),
                                                       "speedup_lower_bound_if_the_cpu_port_scaled_linearly_to_all_%d_threads" % hinfo["nproc"]:
                                                           round(lin / gpu_wall, 1),
                                                       "what": "MSM + witness build + image-part prover at x_logsize=%d d_logsize=%d nbits=%d, "
                                                               "C port with %d threads (MSM: %d, one per window) of %d usable vs 1 MI355X" % (
                                                                   x_log, d_log, nbits, threads, msm_threads, hinfo["usable_cpus"]),
                                                       "target_10x_met": bool(cpu_wall / gpu_wall >= 10.0)}
            if "gen1" in out and "error" not in out["gen1"]:
                lp2, lb2 = min(args.cpu_gen1_log_points, args.gen1_log_points), 8
                b8 = np.random.default_rng(12).integers(0, 2, size=(1 << (lp2 + lb2)), dtype=np.uint8)
                r13 = np.random.default_rng(13)
                tape2 = [int.from_bytes(r13.bytes(64), "little") % P for _ in range(4000)]
                t1 = time.perf_counter()
                cg = O.gkr_msm_prove(pts_h[: 1 << lp2], b8, lp2, lb2, codec.ints_to_limbs(tape2), threads, msgs_cap=1 << 16)
                cpu_g = time.perf_counter() - t1
                gg = harness.gkr_msm_prove(harness.to_dev(pts_h[: 1 << lp2]), torch.from_numpy(b8).cuda(), lp2, lb2, tape2,
                                           msgs_cap=1 << 16)
                gg = harness.gkr_msm_prove(harness.to_dev(pts_h[: 1 << lp2]), torch.from_numpy(b8).cuda(), lp2, lb2, tape2,
                                           msgs_cap=1 << 16)
                gpu_g = gg["call_s"]
                assert codec.from_mont_limbs(cg["msgs"]) == gg["msgs"], "gen-1 GPU transcript differs from the CPU oracle"
                out["gen1"]["cpu_baseline"] = {"value": round((1 << lp2) / cpu_g, 1), "unit": "points/s", "cores": threads,
                                               "kind": "port",
                                               "sample": "gkr_msm_prove log_num_points=%d (GPU figure above is at %d): cpu %.2f s" % (
                                                   lp2, args.gen1_log_points, cpu_g),
                                               "gpu_same_sample_points_per_sec": round((1 << lp2) / gpu_g, 1),
                                               "parity": "bit-exact (%d transcript messages)" % len(gg["msgs"])}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if rcomm is not None:
        sync_all()
        rcomm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
