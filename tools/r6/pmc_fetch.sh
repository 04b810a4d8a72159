#!/bin/bash
# Round 6: FETCH_SIZE / WRITE_SIZE of the bench step under the current environment -> gpurun_out/pmcq_<tag>/summary.csv  (bash tools/r6/pmc_fetch.sh <tag>)
TAG=${1:-q}
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o run -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary --no-fp32-leg > "$OUT/pmc_$c.log" 2>&1
done
python3 tools/prof/pmc_summary.py "$OUT" > "$OUT/summary.csv"
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*_counter_collection.csv" -delete
