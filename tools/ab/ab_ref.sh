#!/bin/bash
# A/B of the working tree against a git ref ON ONE BOX, whole package (Python + library), bench runs interleaved.
#   here  (build container):  bash tools/ab/ab_ref.sh prepare <ref>     -> builds <ref> into ab_base/ (git-ignored, shipped by gpurun)
#   there (gpurun):           bash tools/ab/ab_ref.sh run [n]           -> n interleaved pairs, prints ms per step of each
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = "prepare" ]; then
  rm -rf /tmp/ab_ref ab_base
  git worktree remove --force /tmp/ab_ref 2>/dev/null || true
  git worktree add --detach /tmp/ab_ref "$2" > /dev/null
  make -s -j8 -C /tmp/ab_ref/mlsp_amd/csrc > /dev/null
  mkdir -p ab_base
  cp -r /tmp/ab_ref/mlsp_amd /tmp/ab_ref/bench.py /tmp/ab_ref/tests /tmp/ab_ref/oracle ab_base/ 2>/dev/null
  rm -rf ab_base/mlsp_amd/csrc/build ab_base/tests/golden
  git worktree remove --force /tmp/ab_ref
  echo "ab_base = $(git rev-parse --short "$2")"
  exit 0
fi
n=${2:-3}
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], d['blocks_ms_per_step']['min'])"; }
for i in $(seq 1 $n); do
  echo -n "new: "; python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | ms
  echo -n "old: "; (cd ab_base && python bench.py --no-cpu-baseline --no-secondary 2>/dev/null || python bench.py --no-cpu-baseline 2>/dev/null) | ms
done
