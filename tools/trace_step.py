#!/usr/bin/env python3
"""Break one PPO update step out of a rocprofv3 kernel trace CSV:

    python tools/trace_step.py <kernel_trace.csv> [v] [step index]

`v` also lists the step's kernels on a time line.  Step index counts the loss kernels of the trace (default -2, a
step of bench.py's last pass -- the serial roofline pass; use e.g. 5 for a step of the first, overlapped, pass)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ts = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
loss_idx = [i for i, t in enumerate(ts) if 'ppo_loss_kernel' in t[2]]
build_idx = [i for i, t in enumerate(ts) if 'build_rows_kernel' in t[2]]
li = loss_idx[int(sys.argv[3]) if len(sys.argv) > 3 else -2]
start = max(i for i in build_idx if i < li)
end = min([i for i in build_idx if i > li] or [len(ts)])  # (the trace's last step has no row builder behind it)
seg = ts[start:end]
tot = collections.OrderedDict()
for s, e, n in seg:
    n = n.replace('dppo::', '').replace('void ', '').split('(')[0][:70]
    d = tot.setdefault(n, [0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
wall = (seg[-1][1] - seg[0][0]) / 1e3
print(f"one update step: {len(seg)} kernels, wall {wall:.1f} us, sum of kernel time {sum(v[1] for v in tot.values()):.1f} us")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:70s} x{c:3d} {t:8.1f} us")
if len(sys.argv) > 2:
    prev = seg[0][1]
    for s, e, n in seg:
        print(f"{(s - seg[0][0]) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} gap {(s - prev) / 1e3:6.1f}  {n.replace('dppo::', '')[:80]}")
        prev = e
