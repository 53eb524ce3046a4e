#!/usr/bin/env python3
"""Per-kernel busy time and the idle gap in front of each launch, from a rocprofv3 --kernel-trace sqlite database.

usage: trace_gaps.py results.db [first_kernel_index]
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if first < 0:  # start after the largest idle gap in the second half of the trace (the last timed repetition)
    h = len(rows) // 3
    first = max(range(h, len(rows) - 1), key=lambda i: rows[i + 1][1] - rows[i][2]) + 1
seg = rows[first:]
print(len(seg), "launches,", (seg[-1][2] - seg[0][1]) / 1e6, "ms")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev = seg[0][1]
for n, s, e in seg:
    k = n.split('(')[0].replace('void ', '').replace('gm::', '')[:44]
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
    agg[k][2] += max(0, s - prev) / 1e3
    prev = e
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-46s n=%5d busy %9.1f us  idle-before %9.1f us" % (k, *v))
