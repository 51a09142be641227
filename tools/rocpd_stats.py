"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table (CSV on stdout)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {namecol}, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r"\s+", " ", name)
    d = agg.setdefault(name, [0, 0, 1 << 62, 0])
    dur = e - s
    d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
total = sum(v[1] for v in agg.values())
print("kernel,calls,total_ms,avg_us,min_us,max_us,pct")
for name, (n, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"\"{name[:150]}\",{n},{t/1e6:.3f},{t/n/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*t/total:.2f}")
