#!/bin/bash
# Counter passes of the BASELINE.json configs[4] step (tools/time_config4.py, bf16 mode, seeded backward) on the GPU box, each in its own
# run with --kernel-trace only; -> gpurun_out/prof_<tag>/summary_config4.csv (tools/prof/pmc_summary.py).  bash tools/prof/prof_config4_pmc.sh <tag>
set -u
TAG=${1:-rX}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export C4_MODE=bf16 C4_SEEDED=1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o run -- python3 tools/time_config4.py > "$OUT/pmc_$c.log" 2>&1
done
python3 tools/prof/pmc_summary.py "$OUT" > "$OUT/summary_config4.csv"
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*_counter_collection.csv" -delete
ls -la "$OUT"
