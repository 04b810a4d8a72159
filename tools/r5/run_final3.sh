cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/prof/run_profiles.sh r5s3 > gpurun_out/f3_profiles.log 2>&1
python3 tools/r5/step_sequence.py gpurun_out/prof_r5s3/stats > gpurun_out/f3_sequence.txt 2>/dev/null
find gpurun_out/prof_r5s3 -name "*_kernel_trace.csv" -delete
python tools/prof/stats_table.py gpurun_out/prof_r5s3/stats/run_kernel_stats.csv | head -3
cut -c1-200 gpurun_out/prof_r5s3/bench.json
