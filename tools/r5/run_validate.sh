set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/s2_tests.log 2>&1 || (tail -30 gpurun_out/s2_tests.log; exit 1)
tail -3 gpurun_out/s2_tests.log
python bench.py --no-cpu-baseline --no-secondary --no-fp32-leg > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err
OUT=$PWD/gpurun_out/prof_s2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-fp32-leg > "$OUT/stats.log" 2>&1
python3 tools/r5/step_sequence.py $OUT/stats > gpurun_out/s2_sequence.txt
find "$OUT" -name "*_kernel_trace.csv" -delete
cut -c1-400 gpurun_out/s2_bench.json
