set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "wide_mixed or skinny or fc_" > gpurun_out/sk_tests.log 2>&1 || (tail -30 gpurun_out/sk_tests.log; exit 1)
tail -2 gpurun_out/sk_tests.log
for i in 1 2 3; do
for l in new sk16 skw16; do printf "%s " $l; MLSP_HIP_LIB=$PWD/ab_libs/$l.so python bench.py --no-cpu-baseline --no-secondary --no-fp32-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])"; done
done > gpurun_out/sk_ab.txt 2>&1
cat gpurun_out/sk_ab.txt
