set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -k "not knn" > gpurun_out/mg_tests.log 2>&1 || (tail -40 gpurun_out/mg_tests.log; exit 1)
tail -3 gpurun_out/mg_tests.log
bash tools/ab/ab_env.sh MLSP_SKINNY_NO_PAIR=1 MLSP_X=0 3 > gpurun_out/mg_ab.txt 2>&1
cat gpurun_out/mg_ab.txt
