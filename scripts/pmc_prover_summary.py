#!/usr/bin/env python3
"""HBM traffic of the prover's kernels: totals of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, they do not
fit one pass) divided by the total kernel time of a --kernel-trace run of the same script.  FETCH_SIZE is doubled (the gfx950
correction for wide coalesced streams, MI355X_MICROARCH.md).

usage: pmc_prover_summary.py fetch_counter_collection.csv write_counter_collection.csv kernel_trace_results.db > out.csv
"""
import collections
import csv
import sqlite3
import sys


def key(n):
    return n.split("(")[0].replace("void ", "").replace("gm::", "")


def load(path, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            a = acc[key(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"]) * 1024
    return acc


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
dur = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in sqlite3.connect(sys.argv[3]).execute("select name,start,end from kernels"):
    d = dur[key(n)]
    d[0] += 1
    d[1] += (e - s) / 1e9
rows = []
for k in f:
    fb, wb, d = f[k][1], w.get(k, [0, 0])[1], dur.get(k, [0, 0.0])
    if d[1] > 0 and 2 * fb + wb > 5e8:
        rows.append((2 * fb + wb, k, f[k][0], d[0], fb, wb, d[1]))
print("kernel,pmc_dispatches,trace_dispatches,FETCH_SIZE_bytes_total,WRITE_SIZE_bytes_total,hbm_bytes_fetch_x2,total_kernel_seconds,avg_GB_per_s")
for tot, k, n, nd, fb, wb, ds in sorted(rows, reverse=True):
    print("%s,%d,%d,%d,%d,%d,%.6f,%.0f" % (k, n, nd, fb, wb, tot, ds, tot / ds / 1e9))
