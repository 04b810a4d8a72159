cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in on off; do
  rm -rf /tmp/pf_$v
  if [ $v = on ]; then export MLSP_DEFERRED_ACT=1; else unset MLSP_DEFERRED_ACT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$v -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob
for v in ("on","off"):
    f=glob.glob("/tmp/pf_%s/**/*kernel_stats.csv"%v,recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)/30e3
    print(v,"total us/step %.0f"%tot)
    for r in rows:
        n=r["Name"]
        if "gemm_f32_kernel" in n or "bn_act_fwd" in n:
            print("   %-70s %4s  %8.1f us/step  avg %6.1f"%(n[:70],r["Calls"],float(r["TotalDurationNs"])/30e3,float(r["AverageNs"])/1e3))
PY
