#!/bin/bash
# rocprofv3 kernel statistics of a short headline bench run; prints the top kernels.
# Usage (on the MI355X box): bash benchmarks/kstats.sh <tag> [extra bench.py arguments]
TAG=${1:-k}; shift
OUT=gpurun_out/kstats_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > "$OUT.log" 2>&1 || { echo "profiled run failed"; tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" <<'P'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0])))
for r in rows[:12]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>4s} {float(r['AverageNs']) / 1e6:8.3f} ms")
P
grep '^{' "$OUT.log" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'tokens/s', round(d['value']))"
