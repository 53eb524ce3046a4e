"""Writes a dump directory in the format of scripts/dump_reference_vectors.rs.txt from the Python oracle's inputs and THIS library's
proof -- only to exercise scripts/check_reference_dump.py end to end (it proves nothing about the Rust reference)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from pyref import field as F, g1 as G  # noqa: E402

out = sys.argv[1]
x_log, d_log, nbits, clm = 4, 2, 8, 1
os.makedirs(out, exist_ok=True)
n = 1 << x_log
pts = F.random_points(n, 2)
sc = F.random_scalars(n, nbits, 3)
rng = F.SplitMix64(5)
y_log = (((nbits + d_log - 1) // d_log) - 1).bit_length()
r = [rng.next_fr() for _ in range(y_log)]
basis = G.random_points((2 << (x_log + clm)) - 1, 4)
open(os.path.join(out, "meta.txt"), "w").write("%d %d %d %d\n" % (x_log, d_log, nbits, clm))
open(os.path.join(out, "points.bin"), "wb").write(b"".join(p[0].to_bytes(32, "little") + p[1].to_bytes(32, "little") for p in pts))
open(os.path.join(out, "coefs.bin"), "wb").write(b"".join(s.to_bytes(32, "little") for s in sc))
open(os.path.join(out, "r.bin"), "wb").write(b"".join(s.to_bytes(32, "little") for s in r))
open(os.path.join(out, "basis.bin"), "wb").write(b"".join(p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little") for p in basis))
open(os.path.join(out, "proof.bin"), "wb").write(b"")
