cd $GRAFT_REPO_ROOT
for p in 2 3 4; do for sh in "32 1024 20" "16 2048 40"; do set -- $sh; printf "probe %s: " $p; MLSP_HIP_LIB=$PWD/ab_libs/rvp$p.so TN_B=$1 TN_N=$2 TN_K=$3 python tools/time_reverse.py 2>/dev/null; done; done
for sh in "32 1024 20" "16 2048 40"; do set -- $sh; printf "full: "; TN_B=$1 TN_N=$2 TN_K=$3 python tools/time_reverse.py 2>/dev/null; done
