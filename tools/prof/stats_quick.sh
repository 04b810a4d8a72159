#!/bin/bash
# One rocprofv3 kernel-stats pass of the bench step under the CURRENT environment: bash tools/prof/stats_quick.sh <tag>  ->  gpurun_out/prof_<tag>/stats/run_kernel_stats.csv
TAG=${1:-q}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-fp32-leg > "$OUT/stats.log" 2>&1
find "$OUT" -name "*_kernel_trace.csv" -delete
ls "$OUT/stats" | head -3
