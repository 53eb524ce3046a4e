"""Summarise a rocprofv3 --kernel-trace sqlite database: kernel time per name over the LAST repetition of the
traced region (split at the middle k_round launch), and the largest individual launches.  Development aid."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "k_round"
rows = list(db.execute("select name,start,end,grid_x,grid_y from kernels order by start"))
idx = [i for i, r in enumerate(rows) if pat in r[0]]
sub = rows[idx[len(idx) // 2]:] if idx else rows
tot = collections.defaultdict(lambda: [0, 0])
for n, s, e, gx, gy in sub:
    k = n.split("(")[0][:60]
    tot[k][0] += e - s
    tot[k][1] += 1
span = sub[-1][2] - sub[0][1]
print("span %.2f ms, kernel %.2f ms, %d launches" % (span / 1e6, sum(v[0] for v in tot.values()) / 1e6, len(sub)))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%-62s %8.3f ms %5d  avg %.1f us" % (k, v[0] / 1e6, v[1], v[0] / v[1] / 1e3))
for n, s, e, gx, gy in sorted(sub, key=lambda r: -(r[2] - r[1]))[:12]:
    print("%.3f ms at %.2f grid %dx%d %s" % ((e - s) / 1e6, (s - sub[0][1]) / 1e6, gx, gy, n[:60]))
