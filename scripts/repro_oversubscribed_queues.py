"""Reproducer (development aid): the sharded whole proof over 4 processes sharing one GPU while every process holds many hardware queues
(GPU_MAX_HW_QUEUES=16 + extra streams), the condition under which `bench.py --gpus 4` (gloo rehearsal) once returned
"logup total does not match the suppression term".   python scripts/repro_oversubscribed_queues.py [x_log d_log nbits clm] [streams] [runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

if __name__ == "__main__":
    import test_sharded_full_gpu as T
    shape = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (12, 4, 32, 2)
    streams = sys.argv[5] if len(sys.argv) > 5 else "12"
    for it in range(int(sys.argv[6]) if len(sys.argv) > 6 else 3):
        print("run", it, "...", flush=True)
        res = T._run(4, shape, "minimal", env={"GPU_MAX_HW_QUEUES": "16", "GM_TEST_EXTRA_STREAMS": streams, "GM_TEST_SAY_TIMES": "1"})
        print("run", it, "ok:", [r[1] for r in res], flush=True)
