set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_trainer_loop.py -x -q -k "not knn" > gpurun_out/c4c_tests.log 2>&1 || (tail -40 gpurun_out/c4c_tests.log; exit 1)
tail -2 gpurun_out/c4c_tests.log
for i in 1 2 3; do
echo "== new"; C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
echo "== old lib"; MLSP_HIP_LIB=$PWD/ab_libs/new.so C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
done
echo "== new seeded"; C4_SEEDED=1 C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
