"""gm_msm_combine_host (the Horner recombination of pippenger.rs:586-602, host code of the library: 4 x 64-bit Montgomery limbs,
dedicated doubling) against the oracle's recombination on random window points, identity points included -- no GPU needed."""
import numpy as np
import pytest

import oracle_ffi as O
from gkr_msm_amd import codec, harness as H
from pyref import field as F


@pytest.mark.parametrize("d_log,nwin", [(8, 32), (3, 5), (2, 1), (10, 26), (6, 22)])
def test_combine_host_matches_the_oracle(d_log, nwin):
    pts = F.random_points(64, 3 + d_log)
    rng = F.SplitMix64(5 + nwin)
    raw = np.zeros((3 * (d_log + 1), nwin, 4), dtype=np.uint64)
    for i in range(d_log + 1):
        for w in range(nwin):
            x, y = pts[(i * nwin + w) % len(pts)] if (i + w) % 7 else (0, 1)      # every seventh point is the identity
            z = rng.next_fr() or 1
            raw[3 * i + 0, w] = codec.to_mont_limbs([x * z % F.P])[0]
            raw[3 * i + 1, w] = codec.to_mont_limbs([y * z % F.P])[0]
            raw[3 * i + 2, w] = codec.to_mont_limbs([z])[0]
    assert H.combine_host(raw, d_log) == tuple(codec.from_mont_limbs(O.msm_combine(raw, d_log)))
