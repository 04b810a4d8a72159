cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_c4s
mkdir -p $OUT
C4_MODE=bf16 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c4" -o run -- python3 tools/time_config4.py > "$OUT/c4.log" 2>&1
python3 tools/r5/step_sequence.py $OUT/c4 > gpurun_out/c4_sequence.txt
find "$OUT" -name "*_kernel_trace.csv" -delete
tail -2 $OUT/c4.log
