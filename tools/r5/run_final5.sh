cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/prof/run_profiles.sh r5s4 pmc > gpurun_out/f5_profiles.log 2>&1
echo "profiles done"
bash tools/prof/prof_config4.sh r5s4c > gpurun_out/f5_c4.log 2>&1
echo "c4 done"
find gpurun_out/prof_r5s4 gpurun_out/prof_r5s4c -name "*_kernel_trace.csv" -delete
find gpurun_out/prof_r5s4 -name "*_counter_collection.csv" -delete
python tools/prof/stats_table.py gpurun_out/prof_r5s4/stats/run_kernel_stats.csv | head -2
python tools/prof/stats_table.py gpurun_out/prof_r5s4c/c4/run_kernel_stats.csv | head -2
cut -c1-180 gpurun_out/prof_r5s4/bench.json
