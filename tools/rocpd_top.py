"""Per-kernel totals of a rocprofv3 rocpd database (development aid): python tools/rocpd_top.py results.db [calls]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
calls = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"total {tot / 1e6:.3f} ms; per call {tot / 1e3 / calls:.1f} us")
for r in rows[:16]:
    print(f"  {r[0][:80]:80s} n={r[1]:5d} avg={r[2] / 1e3:9.1f} us  per call={r[3] / 1e3 / calls:8.1f} us")
