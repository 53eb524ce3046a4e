#!/usr/bin/env python3
"""Turn the raw rocprofv3 outputs of scripts/collect_profiles.sh (gpurun_out/<dir>) into the committed summaries under
profiles/<round>/: kernel statistics, HBM traffic per launch (two PMC passes; FETCH_SIZE doubled per the gfx950 note in
MI355X_MICROARCH.md) and the per-launch traffic table bench.py reads for `sumcheck.roofline.traffic`.

usage: summarise_profiles.py gpurun_out/r03_prof profiles/r03"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)   # the newest run when a directory was collected twice
    return f[-1] if f else None


def key(n):
    return n.split("(")[0].replace("void ", "").replace("gm::", "")


def pmc(path, name):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name:
                acc[key(row["Kernel_Name"])].append(float(row["Counter_Value"]) * 1024)
    return acc


def durations(path):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[key(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e9)
    return acc


for tag in ("bench", "msm", "prover", "g1"):
    ks = one("%s_kt/*/*kernel_stats.csv" % tag)
    if ks:
        shutil.copy(ks, os.path.join(dst, "%s_kernel_stats.csv" % ("msm_only" if tag == "msm" else tag)))
for f in ("bench_profiled_run.json", "msm_only_profiled_run.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))

# FETCH_SIZE correction per kernel: x 2 for wide coalesced streams (the guide's gfx950 note); x 1 for kernels whose reads are 64-byte
# GATHERS -- calibrated with scripts/ubench/fetch_size_gather.hip (profiles/r04/fetch_size_gather_factors.json: 64-byte gathers from a
# 64 MB or a 1 GiB table read 1.00-1.06 of the known bytes, 32-byte gathers 0.50-0.53: a 64-byte request is tallied at 64 bytes)
GATHER_KERNELS = ("k_add_level0", "k_add_level01", "k_g1_level")


def fetch_factor(k):
    return 1.0 if any(k == g or k.startswith(g + "<") for g in GATHER_KERNELS) else 2.0


per_launch = {}
launches_of = {}
for tag in ("msm", "prover", "g1"):
    ff, wf, kt = one("%s_FETCH_SIZE/*/*counter_collection.csv" % tag), one("%s_WRITE_SIZE/*/*counter_collection.csv" % tag), one("%s_kt/*/*kernel_trace.csv" % tag)
    if not (ff and wf):
        continue
    fetch, write = pmc(ff, "FETCH_SIZE"), pmc(wf, "WRITE_SIZE")
    dur = durations(kt) if kt else {}
    rows = []
    for k in fetch:
        fb, wb = sum(fetch[k]), sum(write.get(k, [0.0]))
        n = len(fetch[k])
        secs = sum(dur.get(k, [])) * (n / max(len(dur.get(k, [])), 1)) if dur.get(k) else 0.0
        ffac = fetch_factor(k)
        rows.append((ffac * fb + wb, k, n, fb / n, wb / n, (fb + wb) / n, (ffac * fb + wb) / n, secs))
    with open(os.path.join(dst, "%s_pmc_hbm.csv" % ("msm_bench" if tag == "msm" else tag)), "w") as out:
        out.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes per launch = counter (KB) * 1024, mean over the launches\n")
        out.write("# of that kernel; hbm_bytes_fetch_x2 applies the gfx950 correction (FETCH_SIZE reports half of wide coalesced reads; gathers are\n")
        out.write("# uncalibrated, so the raw sum is listed too).  avg_GB_per_s = corrected bytes / kernel time of a --kernel-trace run of the same command.\n")
        out.write("# hbm_bytes_corrected: FETCH_SIZE x 2 for coalesced streams, x 1 for the 64-byte-gather kernels (k_add_level0 / k_add_level01 / k_g1_level: calibrated, fetch_size_gather_factors.json)\n")
        out.write("kernel,launches,FETCH_SIZE_bytes_per_launch,WRITE_SIZE_bytes_per_launch,hbm_bytes_raw_per_launch,hbm_bytes_corrected_per_launch,avg_GB_per_s\n")
        for tot, k, n, f1, w1, raw, cor, secs in sorted(rows, reverse=True):
            if tot < 1e7:
                continue
            out.write("%s,%d,%d,%d,%d,%d,%s\n" % (k, n, f1, w1, raw, cor, ("%.0f" % (tot / secs / 1e9)) if secs > 0 else ""))
            if tag != "g1" and k not in per_launch:   # (the prover's run also launches the MSM once: the MSM's own pass comes first and stays)
                per_launch[k] = int(cor)
                launches_of[k] = n

# kernel names as bench.py prints them (k_round_deg2_lean<PROJ_L1,vecvec> ...) from the template arguments rocprof shows
PRIM = {1: "AFF_L1", 2: "AFF_L2", 3: "AFF_L3", 4: "PROJ_L1", 5: "PROJ_L2", 6: "PROJ_L3", 10: "PT_BIT_CHOICE", 100: "AFF_L1+BITCHECK",
        11: "ADD_INVERSES", 12: "LOGUP_LAYER"}
table = {}
for k, v in per_launch.items():
    if k.startswith("k_round_deg2_lean<") or k.startswith("k_round_deg2_lean9<") or k.startswith("k_round_deg2_lean9x2<"):   # all forms print under the bench's one name
        a, b = k[k.index("<") + 1:-1].split(",")
        # only the LARGE launches are what bench.py times; the PMC mean is over all launches of the kernel, which are the same set
        table["k_round_deg2_lean<%s,%s>" % (PRIM.get(int(a), a), "vecvec" if b.strip() == "true" else "dense")] = v
    elif k in ("k_add_level0", "k_add_level01"):
        table[k] = v
# all launches of the level kernels above level 0 of ONE step: the MSM PMC run is `bench.py --steps 3 --warmup 1` = 3 timed + 1 warm-up
# + 2 pipeline-priming + 1 stage-breakdown step = 7 steps (bench.py msm_leg)
ge1 = 0.0
for k in ("k_add_level", "k_add_tail"):
    if k in per_launch:
        ge1 += per_launch[k] * launches_of[k]
steps_in_pmc_run = launches_of.get("k_add_level0", 0) or launches_of.get("k_add_level01", 0)
if ge1 and steps_in_pmc_run:
    # with levels 0 + 1 fused (k_add_level01) the flat k_add_level launches start at level 2
    table["k_add_levels_ge2_per_step" if "k_add_level01" in launches_of else "k_add_levels_ge1_per_step"] = int(ge1 / steps_in_pmc_run)
with open(os.path.join(dst, "prover_pmc_per_launch.json"), "w") as f:
    json.dump(table, f, indent=1, sort_keys=True)
# SQ-counter passes: per kernel, the largest launch (scripts/pmc_sq_summary.py)
import subprocess
for w in ("prover", "msm", "g1"):
    for n_ in (1, 2):
        cc = one("%s_sq%d/*/*counter_collection.csv" % (w, n_))
        if cc:
            with open(os.path.join(dst, "%s_pmc_sq%d.csv" % (w, n_)), "w") as out:
                subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_sq_summary.py"), cc], stdout=out, check=False)
print("wrote", sorted(os.listdir(dst)))
