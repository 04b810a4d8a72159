"""Per-step kernel table from a rocprofv3 kernel-stats CSV: python tools/prof/stats_table.py run_kernel_stats.csv [steps] [other.csv]
(steps: how many bench steps the trace holds -- default: calls of tnet_edge_bwds_kernel; with a second CSV: side-by-side deltas)."""
import csv, re, sys


def load(path, steps=None):
    rows = list(csv.DictReader(open(path)))
    if steps is None:
        steps = next((int(r["Calls"]) for r in rows if r["Name"].startswith("tnet_edge_bwds_kernel")), 1)
    out = {}
    for r in rows:
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
        name = re.sub(r"at::native::\(anonymous namespace\)::|at::native::", "", name)[:70]
        c, t = out.get(name, (0.0, 0.0))
        out[name] = (c + int(r["Calls"]) / steps, t + int(r["TotalDurationNs"]) / steps / 1e3)
    return out, steps


a, steps = load(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None)
b = load(sys.argv[-1])[0] if len(sys.argv) > 2 and not sys.argv[-1].isdigit() else None
tot = sum(t for _, t in a.values()); n = sum(c for c, _ in a.values())
small = [(c, t) for c, t in a.values() if t / max(c, 1e-9) < 8.0]
print("steps %d  kernel time %.1f us/step  launches %.1f/step  (<8us: %.1f launches, %.1f us)" % (steps, tot, n, sum(c for c, _ in small), sum(t for _, t in small)))
for name, (c, t) in sorted(a.items(), key=lambda kv: -kv[1][1]):
    extra = ""
    if b is not None:
        c2, t2 = b.get(name, (0.0, 0.0))
        extra = "   | other %6.1f us (%+.1f)" % (t2, t - t2)
    print("%-70s %5.1f x %7.1f us = %7.1f us%s" % (name, c, t / max(c, 1e-9), t, extra))
if b is not None:
    for name, (c2, t2) in sorted(b.items(), key=lambda kv: -kv[1][1]):
        if name not in a:
            print("%-70s (only in other) %5.1f x = %7.1f us" % (name, c2, t2))
