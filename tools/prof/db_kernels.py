"""Per-kernel (and per-grid-size) average durations from rocprofv3's sqlite output: python tools/prof/db_kernels.py a.db [b.db ...] [--match edge]"""
import sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else "edge_bwd_gather"
args = [a for a in args if a != match]
for path in args:
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, d.grid_size_x, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from {kd} d "
         f"join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x order by 5 desc")
    rows = list(db.execute(q))
    print(path, "total us", round(sum(r[4] for r in rows), 1))
    for r in rows:
        if match in r[0]:
            print("  %-60s grid=%8d n=%4d avg=%8.1f us" % (r[0][:60], r[1], r[2], r[3]))
