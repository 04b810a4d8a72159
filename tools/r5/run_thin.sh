set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "thin or dgcnn or deferred or merged" > gpurun_out/thin_tests.log 2>&1 || (tail -40 gpurun_out/thin_tests.log; exit 1)
tail -3 gpurun_out/thin_tests.log
bash tools/ab/ab_libs.sh $PWD/ab_libs/new.so $PWD/mlsp_amd/libmlsp_hip.so 3
