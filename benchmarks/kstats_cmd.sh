#!/bin/bash
# rocprofv3 kernel statistics of any python command; prints the top kernels.
# Usage (on the MI355X box): bash benchmarks/kstats_cmd.sh <tag> <script.py> [args]   (environment is inherited)
TAG=${1:-k}; shift
OUT=gpurun_out/kstats_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$@" > "$OUT.log" 2>&1 || { echo "profiled run failed"; tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" <<'P'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0])))
for r in rows[:24]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e6:8.3f} ms  total {float(r['TotalDurationNs']) / 1e6:9.2f}")
P
tail -3 "$OUT.log"
