cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r5c4pmc
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  C4_MODE=bf16 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o run -- python3 tools/time_config4.py > "$OUT/pmc_$c.log" 2>&1
  echo "$c done"
done
python3 tools/prof/pmc_summary.py "$OUT" > "$OUT/pmc_summary.csv"
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*_counter_collection.csv" -delete
head -12 "$OUT/pmc_summary.csv" | cut -c1-200
