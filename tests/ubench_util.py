"""Device-side self-check binaries (scripts/ubench/*.hip): `make ubench` prebuilds them next to a digest of their sources; the GPU
tests run the prebuilt binary when the digest is current and rebuild with hipcc otherwise (file times do not survive the copy
to the GPU box, a digest does)."""
import hashlib
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Makefile: UBENCH_HDRS, same order.  The generated multiplier bodies (*.inc) are part of the digest: regenerating one must make the
# prebuilt self-check stale, or the GPU tests would validate the previous multiplier.
HDRS = ("fq14.hip.h", "g1.hip.h", "fq.hip.h", "fr9.hip.h", "fr.hip.h",
        "fq14_mul_gen.inc", "fr9_mul_asm.inc", "fr9_sqr_asm.inc", "fr_mul_asm.inc")


def ubench_exe(name, tmp_path, build_timeout=1200):
    exe = os.path.join(ROOT, "build", "ubench", name)
    src = os.path.join(ROOT, "scripts", "ubench", name + ".hip")
    h = hashlib.sha256()
    for f in [src] + [os.path.join(ROOT, "gkr_msm_amd", "csrc", x) for x in HDRS]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    stamp = exe + ".srchash"
    if os.path.exists(exe) and os.path.exists(stamp) and open(stamp).read().strip() == h.hexdigest():
        return exe
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.fail("no current prebuilt build/ubench/%s and no hipcc on this box" % name)
    exe = str(tmp_path / name)
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-w", "-mllvm", "-enable-misched=0", "-o", exe, src],
                          timeout=build_timeout)
    return exe
