# Phase probes of knn_reverse_kernel: builds that return after the histogram pass (1), the scan (2), the fill pass (3).
set -e
cd $GRAFT_REPO_ROOT
SRC=mlsp_amd/csrc
mkdir -p /tmp/ab && cp $SRC/*.hip $SRC/common.h /tmp/ab/
for v in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DRV_PROBE=$v -c /tmp/ab/knn.hip -o /tmp/ab/knn_$v.o &
done; wait
for v in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab/lib_rv$v.so $(ls $SRC/build/*.o | grep -v knn.o) /tmp/ab/knn_$v.o
done
echo -n "base: "; python tools/time_reverse.py 2>/dev/null | tail -n 1
for v in 1 2 3; do echo -n "probe $v: "; MLSP_HIP_LIB=/tmp/ab/lib_rv$v.so python tools/time_reverse.py 2>/dev/null | tail -n 1; done
