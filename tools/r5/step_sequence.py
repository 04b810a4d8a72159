"""Ordered kernel sequence of ONE bench step from a rocprofv3 kernel trace:
python3 tools/r5/step_sequence.py <dir with *_kernel_trace.csv> > sequence.txt
The step is cut between two consecutive launches of adam_flat_kernel (the last kernel of a step)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_flat_kernel")]
a, b = ends[-3], ends[-2]
t0 = int(rows[a]["End_Timestamp"])
prev = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %7.1f gap %5.1f  %s  grid %s wg %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:110],
                                                      r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
    prev = e
