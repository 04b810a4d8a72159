"""Round 6: what the measuring launches read in the trainer-shaped step (MLSP_AMAX_DUMP=1 python tools/r6/amax_dump_trainer.py 2>&1 | sort | uniq -c)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
out = bench.trainer_shaped_workload(dev, steps=2, repeats=1)
sys.stderr.write("steps_marker ms_per_step %.3f\n" % out["ms_per_step"])
