#!/bin/bash
# Runs on the MI355X box (gpurun): rocprofv3 kernel stats of the default bench plus the PMC
# passes the roofline numbers cite. Counters go in runs of their own (one --pmc set per run,
# within the per-block slot limits), each under its own time limit; output under gpurun_out/.
# Usage: [BENCH_EXTRA="--lazy-pre-acts"] bash benchmarks/profile.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
ARGS="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-auxk ${BENCH_EXTRA:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; exit 1; }
echo "stats pass done"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout -s KILL 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- python3 $ARGS > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($SET) failed"; exit 1; }
    echo "pmc pass $i done: $SET"
done
python3 benchmarks/summarize_profile.py "$OUT" > "$OUT/summary.json" && echo "summary written"
