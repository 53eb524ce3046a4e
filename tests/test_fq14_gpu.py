"""The 14 x 28-bit form of the BLS12-381 base field (csrc/fq14.hip.h) and the three G1 additions evaluated in it (csrc/g1.hip.h:
g1_add14 / g1_add_mixed14 / g1_add_aff14) against the 12 x 32 path, on the device: 2^18 random operand sets plus the edge cases
(zeros, q - 1, all limbs full, P + P, P + (-P), infinity on either side, equal x with Z = 1) must store bit-identical (X, Y, Z),
and a chain of 101 dependent additions must end in the same point -- scripts/ubench/fq14_test.hip (prebuilt by `make`, built here
when missing).  tests/test_fq14_model_cpu.py is the integer model with the bound assertions; the G1 kernels that use the form are
pinned by the oracle tests (test_g1_gpu.py, test_pushforward_gpu.py, test_pippenger_full_gpu.py)."""
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ubench_util import ubench_exe  # noqa: E402

pytestmark = pytest.mark.gpu


def test_fq14_field_and_g1_additions_match_the_12x32_path(tmp_path):
    exe = ubench_exe("fq14_test", tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fq14 field mismatches: 0 / 262144" in out.stdout
    assert "g1 add mismatches: jacobian 0 mixed 0 affine 0 / 262144" in out.stdout
    assert "chain results equal: yes" in out.stdout
