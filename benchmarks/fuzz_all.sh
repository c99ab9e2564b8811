#!/bin/bash
# Every seeded fuzzer in turn, SECONDS_EACH seconds each (default 140), one log: gpurun_out/fuzz_all.log.
# bash benchmarks/fuzz_all.sh [seconds each]
set -u
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
T=${1:-140}
mkdir -p gpurun_out
LOG=gpurun_out/fuzz_all.log
: > "$LOG"
for f in topk gemm sparse fused decoder auxk long; do
    echo "== ${f}_fuzz.py $T" | tee -a "$LOG"
    timeout -k 10 $((T + 60)) python -u benchmarks/${f}_fuzz.py "$T" >> "$LOG" 2>&1
    rc=$?
    tail -1 "$LOG"
    if [ $rc -ne 0 ]; then echo "rc=$rc: stopping" | tee -a "$LOG"; exit $rc; fi
done
