"""Summarise a rocprofv3 rocpd database (kernel-trace --stats run) into a small text table for profiles/.
usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
    "max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = [f"{'kernel':<100} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6} {'vgpr':>5} {'sgpr':>5} {'lds':>6}"]
for name, calls, tot, avg, mn, mx, vg, sg, lds in rows:
    short = name if len(name) <= 100 else name[:97] + "..."
    lines.append(f"{short:<100} {calls:>6} {tot/1e6:>10.3f} {avg/1e3:>10.1f} {mn/1e3:>10.1f} {mx/1e3:>10.1f} {100*tot/total:>6.2f} {vg:>5} {sg:>5} {lds:>6}")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
