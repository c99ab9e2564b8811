#!/bin/bash
# rocprofv3 kernel statistics of bench.py with extra arguments (e.g. --force-exchange); top kernels by total time.
TAG=${1:-k}; shift
OUT=gpurun_out/kstats_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-auxk "$@" > "$OUT.log" 2>&1 || { echo "profiled run failed"; tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" <<'P'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0])))
for r in rows[:26]:
    print(f"{r['Name'][:96]:96s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e6:8.3f} ms  total/step {float(r['TotalDurationNs']) / 1e6 / 8:7.2f}")
P
grep '^{' "$OUT.log" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3))"
