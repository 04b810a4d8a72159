#!/bin/bash
# SQ counters of the bench step (one pass, 8 SQ slots): where do the non-GEMM kernels spend their wave cycles?
set -u
OUT=$PWD/gpurun_out/prof_${1:-sq}
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d "$OUT/sq" -o run -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary --no-fp32-leg > "$OUT/sq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60] + " g" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
cols = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"]
print("kernel,launches," + ",".join(c + "_per_launch" for c in cols))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    print(k.replace(",", ";") + "," + str(n[k]) + "," + ",".join("%.0f" % (v[c] / max(n[k], 1)) for c in cols))
PY
