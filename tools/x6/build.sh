#!/bin/bash
# builds the harness and leaves the device ISA of every variant in /tmp/x6/x6.s
set -e
cd "$(dirname "$0")"
mkdir -p /tmp/x6
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage -o x6_bench x6_bench.hip 2> /tmp/x6/remarks.txt || { cat /tmp/x6/remarks.txt; exit 1; }
grep -E "error|warning" /tmp/x6/remarks.txt || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o /tmp/x6/x6.s x6_bench.hip 2>/dev/null
python3 - <<'PY'
import re
t = open('/tmp/x6/remarks.txt').read()
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    print("%-50s vgpr %3s agpr %3s occ %s spill %s lds %s" % m.groups())
PY
