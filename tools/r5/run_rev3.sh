set -e
cd $GRAFT_REPO_ROOT
python tools/time_reverse.py 2>/dev/null; TN_B=16 TN_N=2048 TN_K=40 python tools/time_reverse.py 2>/dev/null; TN_B=32 TN_N=2048 TN_K=20 python tools/time_reverse.py 2>/dev/null
timeout -k 10 800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sa.py tests/test_gpu_model.py -x -q -k "knn or reverse or sa or group or ball or segda or dgcnn" > gpurun_out/rev3_tests.log 2>&1 || (tail -40 gpurun_out/rev3_tests.log; exit 1)
tail -3 gpurun_out/rev3_tests.log
bash tools/ab/ab_libs.sh $PWD/ab_libs/new.so $PWD/mlsp_amd/libmlsp_hip.so 3
