"""Per-kernel summary of the separate rocprofv3 --pmc passes written by run_profiles.sh.

FETCH_SIZE / WRITE_SIZE are in KB per dispatch.  On gfx950 FETCH_SIZE undercounts wide (128 B) requests by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM counters section), so the HBM-side estimate is 2*FETCH + WRITE.
Output: CSV on stdout, one row per kernel name (template arguments kept, parameter lists dropped).
"""
import csv, glob, re, sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace(",", ";")[:64]


acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(float)
ndur = defaultdict(int)
for d in glob.glob(root + "/pmc_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                acc[k][c] += float(r["Counter_Value"])
                cnt[k][c] += 1
                if c == "FETCH_SIZE":
                    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    ndur[k] += 1
cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"]
print("kernel,launches,fetch_KB_per_launch_raw,write_KB_per_launch,avg_us_profiled,hbm_side_GBps_2xfetch_plus_write,"
      "mfma_busy_cycles_per_launch,cu_busy_cycles_per_launch,mfma_busy_over_4x_cu_busy")
rows = []
for k in acc:
    n = max(cnt[k].values())
    v = {c: (acc[k][c] / cnt[k][c] if cnt[k][c] else 0.0) for c in cols}
    us = dur[k] / ndur[k] if ndur[k] else 0.0
    gbps = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / (us * 1e-6) / 1e9 if us else 0.0
    # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD (4 per CU); SQ_BUSY_CU_CYCLES per CU
    util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * v["SQ_BUSY_CU_CYCLES"]) if v["SQ_BUSY_CU_CYCLES"] else 0.0
    rows.append((us * n, k, n, v, us, gbps, util))
for tot, k, n, v, us, gbps, util in sorted(rows, reverse=True):
    print(f"{k},{n},{v['FETCH_SIZE']:.0f},{v['WRITE_SIZE']:.0f},{us:.1f},{gbps:.0f},"
          f"{v['SQ_VALU_MFMA_BUSY_CYCLES']:.0f},{v['SQ_BUSY_CU_CYCLES']:.0f},{util:.3f}")
