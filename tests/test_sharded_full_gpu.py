"""GPU test of the WHOLE gen-2 proof sharded by windows (gm_pippenger_wg_create_sharded + gm_pippenger_prove on it; SURVEY 8e,
BASELINE.json configs[4]): every rank owns a block of windows and only the key ranges it reads; the G1 commitments are partial MSMs
combined over the communicator, the opening witnesses / MultiOpenReduction / Knuckles opening run on slices.  The sharded run must
give, on EVERY rank, the unsharded prover's transcript bit for bit -- every scalar, every G1 point, the pairing pair -- and the pair
must satisfy A = tau B (the SRS is tau^i G: the proof verifies).  The unsharded prover is pinned to the oracle by
tests/test_pippenger_full_gpu.py.

Ranks share the one GPU of the box.  A box allows few PROCESSES on its card, so world 8 runs as 4 processes x 2 rank threads (each
thread = one rank with its own stream, plan, communicator handle: the deployment form "one process drives several GPUs"); ranks that
are threads of one process reach each other's buffers by address, ranks in different processes through HIP IPC."""
import os
import sys
import threading

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# Ranks that are threads of ONE process on ONE device: their streams must not share a hardware queue -- a rank's pre-enqueued gate
# kernel spins at the head of its queue until the rank's host has exchanged round sums with the others, whose kernels would sit
# behind it (the HIP runtime multiplexes streams over GPU_MAX_HW_QUEUES = 4 queues by default).  With one GPU per rank -- the
# deployment -- every rank has queues of its own.
MANY_QUEUES = {"GPU_MAX_HW_QUEUES": "16"}


def _proc(ranks, world, tag, shape, q, key_mode, env):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        for k_, v_ in (env or {}).items():
            os.environ[k_] = v_
        import torch
        from gkr_msm_amd import codec, dist as gd, harness as H
        from pyref import field as F
        from pyref import g1 as G
        from pyref import gkr as GK
        x_log, d_log, nbits, clm = shape
        y_size = (nbits + d_log - 1) // d_log
        y_log = (y_size - 1).bit_length()
        assert y_size == 1 << y_log
        n = 1 << x_log
        nv = x_log + clm
        rng = F.SplitMix64(4242 + x_log + clm)
        pts = F.random_points(n, 21)
        sc = F.random_scalars(n, nbits, 22)
        sc[0] = 0
        if os.environ.get("GM_TEST_SCALARS") == "all_same":   # every point in ONE bucket per window (a zero digit among them)
            sc = [0x9B1B9B1B9B1B9B1B & ((1 << nbits) - 1)] * n
        tau, k = rng.next_fr(), 2
        d_pts = H.to_dev(codec.points_to_mont(pts))
        d_sc = H.to_dev(codec.ints_to_limbs(sc))
        n_key = (2 << nv) - 1
        d_basis = H.g1_mock_srs(tau, n_key, G.GEN)
        # the unsharded proof (once per process)
        plan = H.MsmPlan(x_log, d_log, y_size)
        plan.run(d_pts, d_sc)
        wg = H.PippengerWG(plan, d_pts, y_log, clm, d_basis)
        out = wg.dense_output()
        r = [rng.next_fr() for _ in range(y_log)]
        claims = GK.pippenger_claims(out, r)
        tape = [rng.next_bits(128) for _ in range(6000)]
        d_inv = H.knuckles_setup(k, nv)
        ref = wg.prove(claims[0], claims[1], d_inv, k, tape)
        assert ref["pair"][0] == G.mul(ref["pair"][1], tau), "the unsharded proof does not verify"
        torch.cuda.synchronize()
        # GM_TEST_EXTRA_STREAMS=k: k more streams with work on them before the sharded part (a process that also pipelines MSM steps holds
        # that many hardware queues; with GPU_MAX_HW_QUEUES raised, four such processes oversubscribe the device's queues)
        extra_streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("GM_TEST_EXTRA_STREAMS", "0")))]
        for st_ in extra_streams:
            with torch.cuda.stream(st_):
                torch.zeros(1024, device="cuda").add_(1)
        torch.cuda.synchronize()
        results = {}
        prove_ms = {}

        def run_rank(rank):
            try:
                torch.cuda.set_device(0)
                stream = torch.cuda.Stream()
                with torch.cuda.stream(stream):
                    comm = gd.ShmComm("/gm-test-full-%s" % tag, rank, world)
                    y0, y1 = gd.window_range(rank, world, y_size)
                    plan_s = H.MsmPlan(x_log, d_log, y_size, y0, y1)
                    plan_s.run(d_pts, d_sc)
                    key = (H.KeyView.whole(d_basis, n_key) if key_mode == "whole"
                           else H.KeyView.minimal(d_basis, x_log, d_log, y_log, clm, rank, world))
                    wgs = H.PippengerWGSharded(plan_s, d_pts, y_log, clm, key, comm)
                    first, cnt = H.knuckles_slice_of(nv, rank, world)
                    d_inv_s = H.knuckles_setup_range(k, nv, first, cnt)
                    import time as _t
                    t0 = _t.perf_counter()
                    got = wgs.prove(claims[0], claims[1], d_inv_s, k, tape)
                    stream.synchronize()
                    prove_ms[rank] = 1e3 * (_t.perf_counter() - t0)
                    bad = [kk for kk in ("msgs", "points", "pair", "tape_used", "rounds") if got[kk] != ref[kk]]
                    results[rank] = (not bad, "differs from the unsharded proof in %s" % bad if bad else "",
                                     sum(c for _, _, c in zip(key.keep, key.first, key.count)), comm.ipc_stats())
                    wgs.close()
                    comm.close()
            except Exception as e:
                import traceback
                results[rank] = (False, repr(e) + traceback.format_exc(), 0, (0, 0, 0))

        th = [threading.Thread(target=run_rank, args=(rk,)) for rk in ranks]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for rk in ranks:
            q.put((rk,) + results.get(rk, (False, "no result", 0, (0, 0, 0))) + (n_key,))
        if os.environ.get("GM_TEST_SAY_TIMES"):
            print("[sharded whole proof] ranks %s: prove %s ms (incl. Python marshalling), rounds %d, %d transcript scalars, %d points" % (
                ranks, [round(prove_ms.get(rk, -1), 1) for rk in ranks], ref["rounds"], len(ref["msgs"]), len(ref["points"])), flush=True)
    except Exception as e:
        import traceback
        for rk in ranks:
            q.put((rk, False, repr(e) + traceback.format_exc(), 0, (0, 0, 0), 0))


def _run(world, shape, key_mode="minimal", threads_per_proc=1, env=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = "%d-%d-%s" % (os.getpid(), world, "-".join(map(str, shape)))
    groups = [list(range(p, p + threads_per_proc)) for p in range(0, world, threads_per_proc)]
    assert len(groups) <= 6, "a GPU box allows six processes on its card"
    procs = [ctx.Process(target=_proc, args=(g, world, tag, shape, q, key_mode, env)) for g in groups]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=600))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert len(res) == world
    bad = ["rank %d: %s" % (rank, info[:1500]) for rank, ok, info, _, _, _ in sorted(res) if ok is not True]
    assert not bad, "\n".join(bad)
    return sorted(res)


@pytest.mark.parametrize("world,shape,key_mode", [
    (2, (3, 2, 8, 1), "whole"),       # 4 windows, 2 per rank = one matrix per rank
    (2, (3, 2, 8, 1), "minimal"),
    (4, (4, 2, 16, 2), "minimal"),    # 8 windows, clm 2: a matrix spans two ranks
    (4, (4, 3, 12, 0), "minimal"),    # clm 0: one matrix per window
    (4, (5, 2, 8, 2), "minimal"),     # 4 windows, one per rank, ONE matrix over all ranks
])
def test_sharded_whole_proof_equals_the_unsharded_one(world, shape, key_mode):
    res = _run(world, shape, key_mode)
    if key_mode == "minimal" and world >= 4:
        # no rank held the whole key (the point of config E's memory plan)
        assert all(kp < n_key for _, _, _, kp, _, n_key in res)


@pytest.mark.parametrize("shape,gather_log", [
    ((4, 2, 32, 2), "8"),     # 16 windows, 2 per rank, matrices of 4 windows
    ((5, 2, 64, 4), "8"),     # config E's structure at x_logsize 5: 32 windows, commitment_log_multiplicity 4, 8 ranks x 4 windows
    ((5, 2, 64, 4), "0"),     # ... with the sharded dense objects exchanging round sums down to one element per rank, and with
    ((5, 2, 64, 4), "2"),     # ... the slices gathered at four elements per rank
])
def test_sharded_whole_proof_world_8(shape, gather_log):
    """8 ranks = 4 processes x 2 rank threads sharing the GPU"""
    _run(8, shape, "minimal", threads_per_proc=2, env=dict(MANY_QUEUES, GM_SC_SHARD_GATHER_LOG=gather_log, GM_PF_DIST_MIN="2"))


def test_sharded_whole_proof_world_8_all_ranks_threads_of_one_process():
    """one process driving all 8 ranks (the one-process-per-node deployment): every peer buffer is reached by address"""
    _run(8, (4, 2, 32, 2), "minimal", threads_per_proc=8, env=MANY_QUEUES)


def test_sharded_whole_proof_with_a_one_entry_ipc_cache_and_host_staging():
    """GM_SHM_MAX_OPENED=1: every pull that touches two peers overflows the cache -- mappings are closed only after the pull's last
    barrier (advisor r03: an eviction under the running call read a closed mapping); then the same proof with every bulk move staged
    through the host all-gather"""
    res = _run(4, (4, 2, 16, 2), "minimal", env={"GM_SHM_MAX_OPENED": "1", "GM_SC_SHARD_GATHER_LOG": "0", "GM_PF_DIST_MIN": "2"})
    assert any(ipc[1] > 0 for _, _, _, _, ipc, _ in res), "no mapping was ever evicted: the bound was not exercised"
    assert all(ipc[2] <= 2 for _, _, _, _, ipc, _ in res)      # what one call touched may stay, nothing more
    _run(4, (4, 2, 16, 2), "minimal", env={"GM_SHM_NO_IPC": "1"})


def test_sharded_whole_proof_with_every_point_in_one_bucket_per_window():
    """collisions at their maximum on the sharded path: every scalar equal, so a rank's windows hold one full bucket each and empty
    rows otherwise; the access counts a rank contributes are X in one place per window"""
    _run(4, (4, 2, 16, 2), "minimal", env={"GM_TEST_SCALARS": "all_same"})


def test_config_e_structure_at_x_logsize_14_over_eight_ranks():
    """BASELINE.json configs[4]'s structure -- 32 windows of 8 bits, commitment_log_multiplicity 4, 8 ranks x 4 windows, every rank
    holding only its key ranges -- at x_logsize 14 (2^19 matrix entries, 2^18-coefficient opening, a 2^19-point key): large enough for
    the logup tree to stay distributed over several levels, for the opening's compute_t passes to read halos AND whole slices, and
    for the stage kernel to run sharded; 4 processes x 2 rank threads on the one GPU.  The whole transcript and the pairing pair equal
    the unsharded prover's on every rank."""
    res = _run(8, (14, 8, 256, 4), "minimal", threads_per_proc=2, env=dict(MANY_QUEUES, GM_TEST_SAY_TIMES="1"))
    assert all(kp < n_key for _, _, _, kp, _, n_key in res)


@pytest.mark.parametrize("ranks,shape", [(2, (8, 4, 32, 1)), (4, (6, 2, 16, 2)), (8, (5, 2, 64, 4))])
def test_c_example_of_the_sharded_proof(ranks, shape):
    """examples/pippenger_sharded.c: one process, one host thread per rank, plain C against include/gkrmsm.h (gm_stream_create, the
    shared-memory communicator, key views INTO the SRS by gm_pippenger_sharded_key_ranges, gm_knuckles_setup_range) -- every rank ends
    with the same merlin proof bytes, the host verifier and the pairing accept them; the last shape is config E's structure (32 windows,
    clm 4, 8 ranks x 4 windows) at x_logsize 5"""
    import subprocess
    exe = os.path.join(ROOT, "build", "examples", "pippenger_sharded")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "examples"])
    x_log, d_log, nbits, clm = shape
    r = subprocess.run([exe, "--ranks", str(ranks), "--x-logsize", str(x_log), "--d-logsize", str(d_log), "--nbits", str(nbits),
                        "--commitment-log-multiplicity", str(clm)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **MANY_QUEUES))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all ranks hold the same proof and pairing pair" in r.stdout and "proof verified" in r.stdout

