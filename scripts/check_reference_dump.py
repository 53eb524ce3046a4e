#!/usr/bin/env python3
"""Replays a dump of the Rust reference (scripts/dump_reference_vectors.rs.txt) on the GPU and compares the proof bytes.

    python scripts/check_reference_dump.py gkrmsm_dump [num_bits]

Runs gm_msm_run + gm_pippenger_wg_create + gm_pippenger_prove_tr with the library's ProofTranscript2 (label b"fgstglsp", as
examples/pippenger.rs uses) on the dumped points / scalars / claim point / SRS and checks that the proof equals proof.bin
byte for byte.  This is the step that turns "parity unpinned"  into pinned parity; it needs a machine with
cargo to produce the dump and an MI355X to run this script."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_msm_amd import codec, ffi, harness as H  # noqa: E402


def ints(path, width):
    raw = open(path, "rb").read()
    return [int.from_bytes(raw[i:i + width], "little") for i in range(0, len(raw), width)]


def main():
    d = sys.argv[1]
    x_log, d_log, nbits_meta, clm = [int(v) for v in open(os.path.join(d, "meta.txt")).read().split()]
    nbits = int(sys.argv[2]) if len(sys.argv) > 2 else nbits_meta
    assert nbits > 0, "pass num_bits (the dump patch does not know it)"
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    P = codec.P
    pv = ints(os.path.join(d, "points.bin"), 32)
    pts = [(pv[2 * i], pv[2 * i + 1]) for i in range(n)]
    sc = ints(os.path.join(d, "coefs.bin"), 32)
    r = ints(os.path.join(d, "r.bin"), 32)
    bv = ints(os.path.join(d, "basis.bin"), 48)
    basis = [(bv[2 * i], bv[2 * i + 1]) for i in range(len(bv) // 2)]
    nv = x_log + clm
    assert len(basis) >= (2 << nv) - 1 and len(r) == y_log and len(sc) == n
    want = open(os.path.join(d, "proof.bin"), "rb").read()

    L = ffi.lib()
    d_pts = H.to_dev(codec.points_to_mont(pts))
    plan = H.MsmPlan(x_log, d_log, y_size)
    plan.run(d_pts, H.to_dev(codec.ints_to_limbs(sc)))
    wg = H.PippengerWG(plan, d_pts, y_log, clm, H.g1_aff_dev(basis))
    outs = wg.dense_output()

    def ev(poly):
        cur = list(poly)
        for f in reversed(r):
            cur = [(cur[2 * i] + f * (cur[2 * i + 1] - cur[2 * i])) % P for i in range(len(cur) // 2)]
        return cur[0]
    evs = [ev(o) for o in outs]
    d_inv = H.knuckles_setup(2, nv)
    h = C.c_void_p()
    ffi.check(L.gm_merlin_create(b"fgstglsp", 8, C.byref(h)))
    tr = ffi.GmTranscript()
    ffi.check(L.gm_merlin_transcript(h, C.byref(tr)))
    cp, ce, kk = H.fr_arg(r), H.fr_arg(evs), H.fr_arg([2])
    pair = np.zeros(24, dtype=np.uint64)
    used, rounds = C.c_uint64(), C.c_uint64()
    ffi.check(L.gm_pippenger_prove_tr(wg.h, cp.ctypes.data, ce.ctypes.data, C.c_void_p(d_inv.data_ptr()), kk.ctypes.data, C.byref(tr),
                                      pair.ctypes.data, C.byref(used), C.byref(rounds)))
    pp, pn = C.c_void_p(), C.c_uint64()
    ffi.check(L.gm_merlin_proof(h, C.byref(pp), C.byref(pn)))
    got = C.string_at(pp, pn.value)
    print("proof: %d bytes from the GPU, %d bytes in the dump, %d sumcheck rounds" % (len(got), len(want), rounds.value))
    if got == want:
        print("PARITY PINNED: the GPU proof equals the Rust reference's proof byte for byte")
        return 0
    first = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
    print("MISMATCH at byte %d" % first)
    return 1


if __name__ == "__main__":
    sys.exit(main())
