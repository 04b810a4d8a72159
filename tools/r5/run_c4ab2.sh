cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo "v6w"; C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
echo "v5";  MLSP_KNN_V5=1 C4_MODE=bf16 timeout -k 10 200 python tools/time_config4.py 2>/dev/null | cut -c120-
done
