"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected separately: they do not fit one
pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: pmc_summary.py <fetch counter_collection.csv> <write ...csv> > out.csv
Units: the counters are in KB per dispatch.  gfx950 correction (same guide, HBM section): FETCH_SIZE reports half of the bytes
of wide coalesced 16 B/lane streaming reads, so both the raw sum and the sum with FETCH_SIZE doubled are listed."""
import collections
import csv
import sys


def load(path, name):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on: %s" % (sys.argv[3] if len(sys.argv) > 3 else "bench.py"))
print("# units: KB per dispatch (mean over dispatches of that kernel); gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide")
print("# coalesced 16 B/lane streams by 2x; gathers are uncalibrated, so both raw and corrected HBM bytes are listed.")
print("kernel,calls,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_bytes_raw,hbm_bytes_fetch_x2")
rows = []
for k in fetch:
    f = sum(fetch[k]) / len(fetch[k])
    w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
    rows.append((f + w, k, len(fetch[k]), f, w))
for _, k, n, f, w in sorted(rows, reverse=True):
    print("%s,%d,%.1f,%.1f,%d,%d" % (k, n, f, w, int((f + w) * 1024), int((2 * f + w) * 1024)))
