cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/prof/run_profiles.sh r5s2 pmc > gpurun_out/f2_profiles.log 2>&1
echo "profiles done" 
bash tools/prof/prof_config4.sh r5s2c > gpurun_out/f2_c4.log 2>&1
echo "c4 done"
ls gpurun_out/prof_r5s2 gpurun_out/prof_r5s2c
