#!/bin/bash
# Interleaved same-box A/B of two builds of the library: bash tools/ab/ab_libs.sh <libA.so> <libB.so> [rounds]   (MLSP_HIP_LIB selects the library)
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for l in $A $B; do
    printf "%s " "$l"
    MLSP_HIP_LIB=$l python bench.py --no-cpu-baseline --no-secondary --no-fp32-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])"
  done
done
