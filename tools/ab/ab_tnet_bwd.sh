# Phase probes of the Gram-form T-Net backward (tnet_edge_bwdg_kernel) on ONE box: builds with one phase compiled out each
# (-DTG_PROBE_NOMFMA / NOG1 / NOSCATTER / NOEPI; results are wrong, only the time is of interest), kernel time from rocprofv3.
set -e
cd $GRAFT_REPO_ROOT
SRC=mlsp_amd/csrc
mkdir -p /tmp/ab /tmp/include && cp $SRC/*.hip $SRC/common.h /tmp/ab/ && cp include/mlsp_hip.h /tmp/include/
sed -i 's#../../include/mlsp_hip.h#/tmp/include/mlsp_hip.h#' /tmp/ab/api.hip
VARS="${VARS:-NOMFMA NOG1 NOSCATTER NOEPI}"
for v in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DTG_PROBE_$v -c /tmp/ab/tnet.hip -o /tmp/ab/tnet_$v.o &
done; wait
for v in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab/lib_$v.so $(ls $SRC/build/*.o | grep -v tnet.o) /tmp/ab/tnet_$v.o
done
export TMPDIR=/tmp
run() { # name lib
  rm -rf /tmp/prof_$1
  MLSP_HIP_LIB=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o p -- python3 tools/time_tnet.py > /dev/null 2>&1 || true
  python - $1 <<'PY'
import sys, glob, csv
f = glob.glob('/tmp/prof_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)
for row in csv.DictReader(open(f[0])):
    if 'tnet_edge_bwd' in row['Name'] and 'bwd2' not in row['Name']:
        print('%-10s %-40s calls %s avg %.1f us' % (sys.argv[1], row['Name'][:40], row['Calls'], float(row['AverageNs']) / 1e3))
PY
}
run base $GRAFT_REPO_ROOT/mlsp_amd/libmlsp_hip.so
for v in $VARS; do run $v /tmp/ab/lib_$v.so; done
