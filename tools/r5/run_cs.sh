set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -k "dgcnn or colmax or segda or pointnet or tnet" > gpurun_out/cs_tests.log 2>&1 || (tail -30 gpurun_out/cs_tests.log; exit 1)
tail -2 gpurun_out/cs_tests.log
bash tools/ab/ab_libs.sh $PWD/ab_libs/new.so $PWD/ab_libs/cs.so 3 > gpurun_out/cs_ab.txt 2>&1
cat gpurun_out/cs_ab.txt
