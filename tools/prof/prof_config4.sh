#!/bin/bash
# rocprofv3 kernel statistics of BASELINE.json configs[4] (PointSegDA, N = 2048, k = 40, bf16 operands + bf16 activation storage, B = 16)
# and configs[3] (SA stack) on the GPU box: bash tools/prof/prof_config4.sh <tag>
set -u
TAG=${1:-rX}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
C4_MODE=bf16 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c4" -o run -- python3 tools/time_config4.py > "$OUT/c4.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3" -o run -- python3 tools/time_sa.py > "$OUT/c3.log" 2>&1
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
ls -la "$OUT"/*
