# same-box A/B of the T-Net per-edge backward: dense split form (tnet_edge_bwds_kernel) vs the fp32 kernel with the register-indexed sparse
# half (MLSP_TNET_BWD_F32=1), interleaved.  GPU box: bash tools/ab/ab_tnet_bwd_split.sh
for i in 1 2 3; do
  echo "split: $(timeout -k 10 100 python tools/time_tnet.py)"
  echo "f32:   $(MLSP_TNET_BWD_F32=1 timeout -k 10 100 python tools/time_tnet.py)"
done
echo "k=40 split: $(TN_N=2048 TN_K=40 TN_B=8 timeout -k 10 100 python tools/time_tnet.py)"
echo "k=40 f32:   $(TN_N=2048 TN_K=40 TN_B=8 MLSP_TNET_BWD_F32=1 timeout -k 10 100 python tools/time_tnet.py)"
