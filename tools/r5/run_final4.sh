set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/f4_tests.log 2>&1 || (tail -40 gpurun_out/f4_tests.log; exit 1)
tail -3 gpurun_out/f4_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/f4_bench_full.json 2> gpurun_out/f4_bench_full.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/f4_bench_full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('launches_per_step'), d['fp32_mfma']['ms_per_step'])
for s in d.get('secondary',[]): print(s['workload'][:60], s['ms_per_step'], s.get('seeded_backward',{}).get('ms_per_step'))
P
