#!/bin/bash
# Interleaved same-box A/B of one environment switch: bash tools/ab/ab_env.sh VAR=a VAR=b [rounds] [extra bench args]
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for e in "$A" "$B"; do
    printf "%s " "$e"
    env "$e" python bench.py --no-cpu-baseline --no-secondary --no-fp32-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  launches %s' % (d['ms_per_step'], d.get('launches_per_step')))"
  done
done
