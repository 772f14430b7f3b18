#!/usr/bin/env python3
"""Host launch time vs device start time of every kernel of one update step, from
    rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d out -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    python tools/launch_lag.py out/*/*_hip_api_trace.csv out/*/*_kernel_trace.csv [loss index]
Tells a GPU idle gap caused by a late host launch from one caused by a device-side dependency."""
import csv
import sys

api = [r for r in csv.DictReader(open(sys.argv[1])) if "Launch" in r["Function"]]
ker = list(csv.DictReader(open(sys.argv[2])))
by_corr = {r["Correlation_Id"]: r for r in api}
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
loss = [i for i, r in enumerate(ker) if "ppo_loss_kernel" in r["Kernel_Name"]]
li = loss[int(sys.argv[3]) if len(sys.argv) > 3 else 5]
build = [i for i, r in enumerate(ker) if "build_rows_kernel" in r["Kernel_Name"]]
a = max(i for i in build if i < li)
b = min(i for i in build if i > li)
t0 = int(ker[a]["Start_Timestamp"])
prev_end = t0
print(f"{'start':>8s} {'dur':>7s} {'launched':>9s} {'lag':>7s}  kernel      (us; launched = host API call relative to the step's first kernel start)")
for r in ker[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    h = by_corr.get(r["Correlation_Id"])
    ht = (int(h["Start_Timestamp"]) - t0) / 1e3 if h else float("nan")
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {ht:9.1f} {(s - t0) / 1e3 - ht:7.1f}  {r['Kernel_Name'].replace('dppo::', '').replace('void ', '')[:60]}")
