#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics of bench.py and the two PMC passes (FETCH_SIZE, WRITE_SIZE:
# they do not fit one pass) for the MSM leg and for the image-part prover.  Outputs under gpurun_out/$1 (default r03_prof);
# scripts/summarise_profiles.py turns them into profiles/rNN/*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r04_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_kt -- python3 $R/bench.py > $OUT/bench_profiled_run.json 2> $OUT/bench_kt.log || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/msm_kt -- python3 $R/bench.py --no-sumcheck --no-cpu-baseline --g1-log-points 0 > $OUT/msm_only_profiled_run.json 2> $OUT/msm_kt.log || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/msm_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sumcheck --g1-log-points 0 > $OUT/msm_$c.json 2> $OUT/msm_$c.log || exit 1
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/prover_$c -- python3 $R/scripts/quick_prove_time.py 20 8 256 > $OUT/prover_$c.log 2>&1 || exit 1
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prover_kt -- python3 $R/scripts/quick_prove_time.py 20 8 256 > $OUT/prover_kt.log 2>&1 || exit 1
# SQ counters (what the VALU-bound kernels wait on): two passes per workload, the program directly after `--`
for w in prover msm g1; do
  case $w in
    prover) CMD="python3 $R/scripts/quick_prove_time.py 20 8 256";;
    msm) CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sumcheck --g1-log-points 0";;
    g1) CMD="python3 $R/scripts/quick_g1_time.py 21";;
  esac
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/${w}_sq1 -- $CMD > $OUT/${w}_sq1.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/${w}_sq2 -- $CMD > $OUT/${w}_sq2.log 2>&1 || exit 1
done
# G1: kernel statistics + HBM traffic of the sum-by-key engine (verdict r02, weak 10)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/g1_kt -- python3 $R/scripts/quick_g1_time.py 21 > $OUT/g1_kt.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/g1_$c -- python3 $R/scripts/quick_g1_time.py 21 > $OUT/g1_$c.log 2>&1 || exit 1
done
echo collected
