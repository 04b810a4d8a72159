#!/bin/bash
# SQ counters of the GEMM kernels of tools/x6/lib_bench (fp32 MFMA vs split), separate --pmc passes with --kernel-trace only.
set -u
OUT=$PWD/gpurun_out/x6_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
SHAPE=${1:-conv5}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o run -- tools/x6/lib_bench "$SHAPE" 3 > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" not in r["Kernel_Name"] or "reduce" in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][:48] + " g" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, v in sorted(acc.items()):
    print(k)
    for c, x in sorted(v.items()):
        print("    %-32s %14.0f per launch" % (c, x / n[k][c]))
PY
find "$OUT" -name "*.csv" -size +4M -delete
