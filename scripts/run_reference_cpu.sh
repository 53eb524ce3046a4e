#!/bin/bash
# Opportunistic TRUE-reference CPU run: only on a host that has a nightly Rust toolchain and the
# reference's crates available (the build image and the GPU boxes of this project have neither: then this script says so and exits 3).
#
#   scripts/run_reference_cpu.sh /path/to/GKR-MSM [x_logsize] [d_logsize] [nbits] [commitment_log_multiplicity]
#
# Runs the reference's own command (README.md:5 of the reference) and prints the wall time of the spans the reference's
# tracing_span_tree subscriber reports ("compute buckets and commit phase 1", "prove image part", "commit phase 2",
# "prove pushforward", "open"; src/cleanup/protocols/pippenger.rs:518-541, 138-291) as one JSON line, so that it can be pasted next to
# bench.py's cpu_baseline (kind "reference").  The reference's CLI rejects --x-logsize >= 20 (examples/pippenger.rs:39, range 8..20):
# config B needs that bound lifted by hand; this script does not edit the reference.
set -euo pipefail
REF=${1:?usage: run_reference_cpu.sh /path/to/GKR-MSM [x_logsize] [d_logsize] [nbits] [clm]}
X=${2:-16}; D=${3:-8}; NBITS=${4:-128}; CLM=${5:-0}
if ! command -v cargo >/dev/null 2>&1; then
    echo '{"error": "cargo not found: the reference cannot be built on this host; use bench.py cpu_baseline (kind port)"}'
    exit 3
fi
cd "$REF"
OUT=$(mktemp)
START=$(date +%s.%N)
RUSTFLAGS="-Awarnings -C target-cpu=native" cargo run --example pippenger --features parallel --profile release -- \
    --x-logsize "$X" --d-logsize "$D" --nbits "$NBITS" --commitment-log-multiplicity "$CLM" --log 2>&1 | tee "$OUT"
END=$(date +%s.%N)
python3 - "$OUT" "$X" "$D" "$NBITS" "$CLM" "$START" "$END" <<'PY'
import json, re, sys, os
txt = open(sys.argv[1]).read()
spans = {}
for name in ("generating inputs", "computing correct answer", "compute buckets and commit phase 1", "claim computation",
             "prove image part", "commit phase 2", "prove pushforward", "open"):
    m = re.search(r"([0-9.]+)\s*(ns|us|µs|ms|s)\s+" + re.escape(name), txt) or re.search(re.escape(name) + r"[^\n0-9]*([0-9.]+)\s*(ns|us|µs|ms|s)", txt)
    if m:
        v, u = float(m.group(1)), m.group(2)
        spans[name] = v * {"ns": 1e-9, "us": 1e-6, "µs": 1e-6, "ms": 1e-3, "s": 1.0}[u]
print(json.dumps({"kind": "reference", "x_logsize": int(sys.argv[2]), "d_logsize": int(sys.argv[3]), "nbits": int(sys.argv[4]),
                  "commitment_log_multiplicity": int(sys.argv[5]), "cores": os.cpu_count(), "wall_s_incl_build": float(sys.argv[7]) - float(sys.argv[6]),
                  "span_seconds": spans}))
PY
