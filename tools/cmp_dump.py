"""Compares the per-launch GEMM listings of `DUMP=1 python tools/time_precision.py` (stderr -> file) between the two modes."""
import re
import sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/dump.txt").read()
parts = txt.split("==== per-launch listing,")
rows = {}
for part in parts[1:]:
    mode = part.split()[0]
    for m in re.finditer(r"gemm (\w\w) M=(\d+) N=(\d+) K=(\d+) split=(\d+) bm=(\d+) epi=(\d+)\s+([\d.]+) us", part):
        key = m.groups()[:4] + (m.group(7),)
        rows.setdefault(key, {}).setdefault(mode, []).append(float(m.group(8)))
tot = {"fp32": 0, "bf16x6": 0}
out = []
for k, v in rows.items():
    a, b = sum(v.get("fp32", [0])), sum(v.get("bf16x6", [0]))
    tot["fp32"] += a; tot["bf16x6"] += b
    out.append((a, k, len(v.get("fp32", [])), b))
out.sort(reverse=True)
for a, k, n, b in out[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%s M=%-7s N=%-5s K=%-6s epi=%s  x%d  fp32 %7.1f us  x6 %7.1f us  %.2fx" % (*k, n, a, b, a / b if b else 0))
print(tot)
