"""`python bench.py --gpus N` without a launcher starts its N ranks itself (one child process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) before torch or the HIP library are imported; a failing rank fails the run.  The
GM_BENCH_SPAWN_CHECK hook makes every rank report and leave before touching a GPU, so this runs on the CPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, check, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["GM_BENCH_SPAWN_CHECK"] = check
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                          env=env, capture_output=True, text=True, timeout=120)


def test_bench_starts_its_own_ranks():
    out = _run(4, "1")
    assert out.returncode == 0, out.stdout + out.stderr
    lines = sorted(l for l in out.stdout.splitlines() if l.startswith("spawn-check"))
    assert len(lines) == 4
    ports = set()
    for r, l in enumerate(lines):
        f = l.split()
        assert f[2] == str(r) and f[4] == "4" and f[6] == str(r) and f[10] == "1"
        ports.add(f[8])
    assert len(ports) == 1 and ports.pop().isdigit()


def test_a_failing_rank_fails_the_run():
    out = _run(2, "fail:1")
    assert out.returncode != 0


def test_under_a_launcher_nothing_is_spawned():
    # WORLD_SIZE present (torchrun's contract): the process is a rank itself
    out = _run(2, "1", {"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1", "MASTER_PORT": "29999"})
    lines = [l for l in out.stdout.splitlines() if l.startswith("spawn-check")]
    assert out.returncode == 0 and len(lines) == 1 and lines[0].split()[2] == "1" and lines[0].split()[10] == "0"
