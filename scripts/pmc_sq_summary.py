#!/usr/bin/env python3
"""Summarise an SQ-counter pass of rocprofv3 (--pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_INSTS_VALU) per kernel: the largest launch of every kernel, counters as fractions of its wave cycles, and
VALU wave-instructions per SIMD per shader cycle (1024 SIMDs; duration from the dispatch timestamps).

usage: pmc_sq_summary.py counter_collection.csv > summary.csv
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows:
    d = by[(r["Kernel_Name"].split("(")[0].replace("gm::", "").replace("void ", ""), r["Dispatch_Id"])]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    d["vgpr"] = r["VGPR_Count"]
best = {}
for (k, _), d in by.items():
    if k not in best or d.get("SQ_WAVE_CYCLES", 0) > best[k].get("SQ_WAVE_CYCLES", 0):
        best[k] = d
w = csv.writer(sys.stdout)
w.writerow(["kernel", "vgpr_count_as_reported", "duration_us", "active_any_per_wave_cycle", "active_valu_per_wave_cycle", "wait_any_per_wave_cycle",
            "wait_inst_any_per_wave_cycle", "valu_wave_insts", "valu_insts_per_simd_per_kcycle_at_2.4GHz"])
for k, d in sorted(best.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    cyc = d["ns"] * 2.4
    w.writerow([k, d["vgpr"], "%.1f" % (d["ns"] / 1e3)] + ["%.3f" % (d.get(c, 0) / wc) for c in
               ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")] +
               ["%.4g" % d.get("SQ_INSTS_VALU", 0), "%.1f" % (1000 * d.get("SQ_INSTS_VALU", 0) / 1024 / cyc if cyc else 0)])
