set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "compose or segda or config4" > gpurun_out/c4b_tests.log 2>&1 || (tail -40 gpurun_out/c4b_tests.log; exit 1)
tail -3 gpurun_out/c4b_tests.log
for i in 1 2; do
echo "== B16"; C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
echo "== B16 seeded"; C4_SEEDED=1 C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
done
echo "== B32"; C4_B=32 C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
echo "== B32 seeded"; C4_SEEDED=1 C4_B=32 C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
