"""Where the GPU waits: idle gaps between consecutive kernels of a rocprofv3 (rocpd sqlite) kernel trace, grouped by the
kernel that ENDS before the gap and the one that starts after it.
Usage: python tools/rocpd_gaps.py <results.db> [min_gap_us=15] [skip_first_fraction=0.3]"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 15e3
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
rows = list(con.execute("select name, start, end from kernels order by start"))
rows = rows[int(skip * len(rows)):]  # (set-up and warm-up first)
span = rows[-1][2] - rows[0][1]
busy_until, busy = rows[0][1], 0
gaps = collections.defaultdict(lambda: [0, 0])
idle = 0
prev = rows[0][0]
for name, start, end in rows:
    if start > busy_until:
        g = start - busy_until
        idle += g
        if g >= min_gap:
            key = (prev.split("(")[0][-44:], name.split("(")[0][-44:])
            gaps[key][0] += 1
            gaps[key][1] += g
    else:
        start = busy_until
    if end > busy_until:
        busy += end - max(start, busy_until) if start >= busy_until else 0
        busy_until, prev = end, name
print(f"span {span / 1e6:.1f} ms, kernels {len(rows)}, idle {idle / 1e6:.1f} ms ({100 * idle / span:.1f} %)")
print(f"{'after kernel':46s} {'before kernel':46s} {'count':>6s} {'total_ms':>9s} {'avg_us':>8s}")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{a:46s} {b:46s} {c:6d} {t / 1e6:9.2f} {t / c / 1e3:8.1f}")
