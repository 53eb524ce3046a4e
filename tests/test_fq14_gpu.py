"""The 14 x 28-bit form of the BLS12-381 base field (csrc/fq14.hip.h) and the three G1 additions evaluated in it (csrc/g1.hip.h:
g1_add14 / g1_add_mixed14 / g1_add_aff14) against the 12 x 32 path, on the device: 2^18 random operand sets plus the edge cases
(zeros, q - 1, all limbs full, P + P, P + (-P), infinity on either side, equal x with Z = 1) must store bit-identical (X, Y, Z),
and a chain of 101 dependent additions must end in the same point -- scripts/ubench/fq14_test.hip (prebuilt by `make`, built here
when missing).  tests/test_fq14_model_cpu.py is the integer model with the bound assertions; the G1 kernels that use the form are
pinned by the oracle tests (test_g1_gpu.py, test_pushforward_gpu.py, test_pippenger_full_gpu.py)."""
import hashlib
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fq14_field_and_g1_additions_match_the_12x32_path(tmp_path):
    exe = os.path.join(ROOT, "build", "ubench", "fq14_test")
    src = os.path.join(ROOT, "scripts", "ubench", "fq14_test.hip")
    hdrs = [os.path.join(ROOT, "gkr_msm_amd", "csrc", h) for h in ("fq14.hip.h", "g1.hip.h", "fq.hip.h", "fr9.hip.h", "fr.hip.h")]
    h = hashlib.sha256()
    for f in [src] + hdrs:   # the Makefile records the same digest next to the binary it builds (file times do not survive a copy)
        with open(f, "rb") as fh:
            h.update(fh.read())
    stamp = exe + ".srchash"
    fresh = os.path.exists(exe) and os.path.exists(stamp) and open(stamp).read().strip() == h.hexdigest()
    if not fresh:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("no prebuilt build/ubench/fq14_test and no hipcc on this box")
        exe = str(tmp_path / "fq14_test")
        subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-w", "-mllvm", "-enable-misched=0", "-o", exe, src],
                              timeout=1200)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fq14 field mismatches: 0 / 262144" in out.stdout
    assert "g1 add mismatches: jacobian 0 mixed 0 affine 0 / 262144" in out.stdout
    assert "chain results equal: yes" in out.stdout
