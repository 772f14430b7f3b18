#!/usr/bin/env python3
"""Per-kernel summary (calls, average, total, share) from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats`
writes `*_results.db` on this ROCm): python tools/rocpd_stats.py <results.db> [out.csv] [min_start_fraction]

min_start_fraction (0..1) drops the dispatches of the first part of the run (set-up: rollout generation), so the table is
the steady state of the timed steps."""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select * from kernels").fetchall()
ix = {c: i for i, c in enumerate(cols)}
name_c = "name" if "name" in ix else [c for c in cols if "name" in c][0]
t0 = min(r[ix["start"]] for r in rows)
t1 = max(r[ix["end"]] for r in rows)
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cut = t0 + frac * (t1 - t0)
agg = {}
for r in rows:
    if r[ix["start"]] < cut:
        continue
    n = r[ix[name_c]].replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    n = re.sub(r"^void ", "", n)
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += r[ix["end"]] - r[ix["start"]]
tot = sum(a[1] for a in agg.values())
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
lines = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")]
for n, (c, d) in out:
    lines.append((n, c, d, d / c, 100.0 * d / tot))
if len(sys.argv) > 2 and sys.argv[2] != "-":
    csv.writer(open(sys.argv[2], "w")).writerows(lines)
for n, c, d, a, p in lines[1:40]:
    print(f"{n[:100]:100s} {c:6d} {a / 1e3:9.2f} us {d / 1e6:9.3f} ms {p:5.1f}%")
print(f"total kernel time {tot / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches")
