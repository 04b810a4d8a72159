#!/bin/bash
# What would an operand transform (BatchNorm scale / shift + LeakyReLU on the A operand) cost inside gemm_split_kernel's split stage?
# Builds gemm.hip with -DSX_XF_ABLATION (three more vector instructions per A element, identity parameters: same results) against the
# shipped objects and times the head-layer shapes with tools/x6/lib_bench in mode 2, interleaved with the shipped library.  GPU box:
#   bash tools/x6/xf_ablation.sh > gpurun_out/xf_ablation.txt
set -e
cd "$(dirname "$0")/../.."
SRC=mlsp_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSX_XF_ABLATION -c $SRC/gemm.hip -o /tmp/gemm_xf.o
OBJS=$(ls $SRC/build/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmlsp_xf.so $OBJS /tmp/gemm_xf.o
/opt/rocm/bin/hipcc -O2 -o /tmp/lib_bench tools/x6/lib_bench.cpp -ldl
for shape in "dens1 fwd" "head2 fwd" "conv5 fwd" "heads K128"; do
  for rep in 1 2; do
    echo "== shipped   : $(/tmp/lib_bench "$shape" 2>/dev/null | grep -v '^$' | tail -1)"
    echo "== +transform: $(MLSP_HIP_LIB=/tmp/libmlsp_xf.so /tmp/lib_bench "$shape" 2>/dev/null | grep -v '^$' | tail -1)"
  done
done
