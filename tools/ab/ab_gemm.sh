set -e
cd $GRAFT_REPO_ROOT
SRC=mlsp_amd/csrc
mkdir -p /tmp/ab && cp $SRC/*.hip $SRC/common.h /tmp/ab/ && mkdir -p /tmp/include && cp include/mlsp_hip.h /tmp/include/
cp tools/ab/gemm_old.hip /tmp/ab/gemm.hip
sed -i 's#../../include/mlsp_hip.h#/tmp/include/mlsp_hip.h#' /tmp/ab/api.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared -o /tmp/ab/libold.so /tmp/ab/*.hip
for i in 1 2; do
echo "== new"; python tools/bench_gemm.py 2>/dev/null | grep -E "conv5|head1|head2|dens1|edge4"
echo "== old"; MLSP_HIP_LIB=/tmp/ab/libold.so python tools/bench_gemm.py 2>/dev/null | grep -E "conv5|head1|head2|dens1|edge4"
done
