set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "knn or segda or config4" > gpurun_out/kw2_tests.log 2>&1 || (tail -40 gpurun_out/kw2_tests.log; exit 1)
tail -3 gpurun_out/kw2_tests.log
for i in 1 2; do
echo "== v6w"; C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null
echo "== v5";  MLSP_KNN_V5=1 C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null
done
