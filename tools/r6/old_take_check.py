"""One-off (round 6): shows that the subset-loss tests catch the advisor's defect -- BwdStats.take() with the ROUND-5 behaviour
(partial delivery -> None -> the producer masks a second time) must make them fail.  Run on the GPU box:
    python tools/r6/old_take_check.py"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from mlsp_amd import functional as Fh  # noqa: E402


def old_take(self):
    if self.part is None or self.task != torch._C._current_graph_task_id() or self.delivered != len(self.promised) or not self.agreed():
        self.part = self.task = None
        return None
    part, self.part, self.task = self.part, None, None
    return part, part.shape[0]


Fh.BwdStats.take = old_take
rc = pytest.main(["-q", "-m", "gpu", os.path.join(ROOT, "tests"), "-k", "subset_of_the_consumers or subset_of_the_heads", "--tb=line"])
print("old behaviour: pytest rc = %d (expected: failures)" % rc)
sys.exit(0 if rc == 1 else 1)
