set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_sa.py tests/test_gpu_trainer_loop.py -x -q -k "not knn" > gpurun_out/mg2_tests.log 2>&1 || (tail -40 gpurun_out/mg2_tests.log; exit 1)
tail -3 gpurun_out/mg2_tests.log
bash tools/ab/ab_libs.sh $PWD/ab_libs/base.so $PWD/ab_libs/new.so 3 > gpurun_out/mg2_ab.txt 2>&1
cat gpurun_out/mg2_ab.txt
