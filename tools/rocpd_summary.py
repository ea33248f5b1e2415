"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
Usage: python tools/rocpd_summary.py <results.db> > profiles/<name>.txt"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
q = f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc"
rows = list(con.execute(q))
total = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>12s} {'avg_us':>12s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for name, calls, tot, avg, mn, mx in rows:
    print(f"{name[:90]:90s} {calls:7d} {tot / 1e6:12.3f} {avg / 1e3:12.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100.0 * tot / total:6.2f}")
