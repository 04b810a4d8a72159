set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -k "knn" > gpurun_out/kw_tests.log 2>&1 || (tail -40 gpurun_out/kw_tests.log; exit 1)
tail -3 gpurun_out/kw_tests.log
(echo "== default"; timeout -k 10 120 python tools/time_knn_shapes.py; echo "== MLSP_KNN_V6W_ALL"; MLSP_KNN_V6W_ALL=1 timeout -k 10 120 python tools/time_knn_shapes.py; echo "== MLSP_KNN_V5"; MLSP_KNN_V5=1 timeout -k 10 120 python tools/time_knn_shapes.py; echo "== OFFSET 1.9 default"; OFFSET=1.9 timeout -k 10 120 python tools/time_knn_shapes.py; echo "== OFFSET 1.9 v5";  OFFSET=1.9 MLSP_KNN_V5=1 timeout -k 10 120 python tools/time_knn_shapes.py) > gpurun_out/kw_times.txt 2>&1
cat gpurun_out/kw_times.txt
