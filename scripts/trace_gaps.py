#!/usr/bin/env python3
"""Per-kernel busy time and the idle gap in front of each launch, from a rocprofv3 --kernel-trace output (the sqlite database
or the *_kernel_trace.csv), plus the fraction of the wall time some kernel was running (kernel time / wall).

usage: trace_gaps.py results.db|kernel_trace.csv [first_kernel_index]      (-1: the last repetition of the traced program)
"""
import collections
import csv
import sqlite3
import sys

if sys.argv[1].endswith(".csv"):
    with open(sys.argv[1], newline="") as fh:
        rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(fh)),
                      key=lambda r: r[1])
else:
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name,start,end from kernels order by start"))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if first < 0:  # start after the largest idle gap in the second half of the trace (the last timed repetition)
    h = len(rows) // 3
    first = max(range(h, len(rows) - 1), key=lambda i: rows[i + 1][1] - rows[i][2]) + 1
seg = rows[first:]
wall = seg[-1][2] - seg[0][1]
# time covered by at least one kernel (launches of the stage kernel overlap nothing else; streams may overlap in other traces)
cov, cur_s, cur_e = 0, seg[0][1], seg[0][2]
for n, s_, e_ in seg[1:]:
    if s_ > cur_e:
        cov += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
cov += cur_e - cur_s
print(len(seg), "launches,", wall / 1e6, "ms; some kernel running %.1f ms = %.3f of the wall time" % (cov / 1e6, cov / wall))
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
