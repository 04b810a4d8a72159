#!/bin/bash
# Phase timing of knn6_kernel on the GPU box: builds probe variants of knn6.hip (early exits, -DK6_PROBE=n) against the shipped objects
# and times mlsp_knn_f32 at the configs[1] shapes.   1: after pass A | 2: after tau | 3: after pass B + key conversion | 4: no exact
# recompute (F2 skipped) | 0: the whole kernel.    bash tools/knn6_probe.sh > gpurun_out/knn6_probe.txt
set -e
cd "$(dirname "$0")/.."
SRC=mlsp_amd/csrc
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DK6_PROBE=$n -c $SRC/knn6.hip -o /tmp/k6_$n.o &
done
wait
OBJS=$(ls $SRC/build/*.o | grep -v knn6.o)
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libk6_$n.so $OBJS /tmp/k6_$n.o
done
echo "== full kernel"; python tools/time_knn_shapes.py 2>/dev/null | head -3
for n in 1 2 3 4; do echo "== probe $n"; MLSP_HIP_LIB=/tmp/libk6_$n.so python tools/time_knn_shapes.py 2>/dev/null | head -3; done
