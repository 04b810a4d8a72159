# Phase probes of the register-resident T-Net forward (tnet_edge_fwd2_kernel) on ONE box: -DTF_PROBE_NOMFMA / -DTF_PROBE_NOGATHER builds
# against the stock library, timed with tools/time_tnet.py.  Round-2 result (B=32 N=1024 k=20, whole forward op incl. the uv GEMM):
# base 230 us, no MFMA loop 135 us, no neighbour gather 224 us; delaying the second workgroup of every CU by 4-20 us: +2..+14 us.
set -e
cd $GRAFT_REPO_ROOT
SRC=mlsp_amd/csrc
mkdir -p /tmp/ab /tmp/include && cp $SRC/*.hip $SRC/common.h /tmp/ab/ && cp include/mlsp_hip.h /tmp/include/
sed -i 's#../../include/mlsp_hip.h#/tmp/include/mlsp_hip.h#' /tmp/ab/api.hip
cd /tmp/ab && for f in *.hip; do [ $f = tnet.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c $f -o ${f%.hip}.o & done; wait
for v in NOMFMA NOGATHER; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DTF_PROBE_$v -c /tmp/ab/tnet.hip -o /tmp/ab/tnet_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab/lib_$v.so $(ls /tmp/ab/*.o | grep -v tnet_) /tmp/ab/tnet_$v.o
done
cd $GRAFT_REPO_ROOT
echo -n "base:     "; python tools/time_tnet.py 2>/dev/null
for v in NOMFMA NOGATHER; do echo -n "$v: "; MLSP_HIP_LIB=/tmp/ab/lib_$v.so python tools/time_tnet.py 2>/dev/null; done
