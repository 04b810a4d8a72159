#!/bin/bash
# A/B on ONE box: build a variant library with tools/ab/<file>_old.hip substituted, run bench with both, interleaved.
set -e
cd "$(dirname "$0")/../.."
SRC=mlsp_amd/csrc
mkdir -p /tmp/ab && cp $SRC/*.hip $SRC/common.h /tmp/ab/ && mkdir -p /tmp/include && cp include/mlsp_hip.h /tmp/include/
for f in tools/ab/*_old.hip; do b=$(basename $f _old.hip); cp $f /tmp/ab/$b.hip; done
sed -i 's#../../include/mlsp_hip.h#/tmp/include/mlsp_hip.h#' /tmp/ab/api.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared -o /tmp/ab/libold.so /tmp/ab/*.hip
for i in 1 2 3; do
  echo -n "new: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo -n "old: "; MLSP_HIP_LIB=/tmp/ab/libold.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
