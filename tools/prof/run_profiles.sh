#!/bin/bash
# Runs on the GPU box (gpurun): one kernel-trace/stats pass and separate --pmc passes of the SAME bench command,
# everything under gpurun_out/prof_<tag>/.  Usage: bash tools/prof/run_profiles.sh <tag> [pmc]
# The PMC passes are collected on their own with --kernel-trace only (never with sys/runtime/hip traces).
set -u
TAG=${1:-rX}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-fp32-leg"
python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ARGS > "$OUT/stats.log" 2>&1
if [ "${2:-}" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o run -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary --no-fp32-leg > "$OUT/pmc_$c.log" 2>&1
  done
  python3 tools/prof/pmc_summary.py "$OUT" > "$OUT/pmc_summary.csv"
fi
# keep only the small summaries (gpurun_out merge is capped at 64 MiB)
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*_counter_collection.csv" -size +8M -delete
ls -la "$OUT"
