"""The 9 x 29-bit field form (csrc/fr9.hip.h) against the 8 x 32-bit one on the device: 2^20 random products + the edge cases
(0, 1, p - 1, all-ones below p), a formula through every lazy helper (sum, difference with the 32 p bias, times 5, square,
normalise, times d), and the chain of 400 dependent products -- scripts/ubench/fr9_mul_test.hip (prebuilt by `make`, rebuilt here only when its sources changed).
A property test of the HIP path against itself, not parity:
The kernels that compute in this form (MSM levels, the 3- and 4-input large-round kernels) are pinned by the oracle tests."""
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ubench_util import ubench_exe  # noqa: E402

pytestmark = pytest.mark.gpu


def test_fr9_products_and_lazy_formulas_match_the_8x32_field(tmp_path):
    exe = ubench_exe("fr9_mul_test", tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fr9 vs fr mismatches: 0 / 1048576" in out.stdout and "lazy formula mismatches: 0" in out.stdout
    assert out.stdout.count("chain results equal: yes") == 1
