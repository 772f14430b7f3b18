#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out -- python3 bench.py ...
    python tools/mfma_util.py out/*/*counter_collection.csv [out.json]

util = MFMA-busy cycles summed over the chip's 1024 SIMDs / (1024 x GPU-active cycles of the dispatch).  rocprofv3
reports GRBM_GUI_ACTIVE summed over the 8 XCDs (checked against event-timed kernels: gemm_tn 42 us x 2.4 GHz = 1.0e5
cycles, counter 8.1e5), hence the / 8.  In a --pmc run dispatches are serialised: each kernel has the chip to itself.
"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4
XCDS = 8
rows = collections.defaultdict(lambda: collections.defaultdict(dict))
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["Kernel_Name"]][r["Dispatch_Id"]].setdefault(r["Counter_Name"], 0.0)
    rows[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
out = []
for k, disp in rows.items():
    mf = [d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in disp.values()]
    ga = [d.get("GRBM_GUI_ACTIVE", 0.0) for d in disp.values()]
    if sum(mf) <= 0 or sum(ga) <= 0:
        continue
    out.append({"kernel": k.replace("dppo::", "")[:100], "dispatches": len(mf),
                "mfma_busy_cycles_avg": sum(mf) / len(mf), "gpu_active_cycles_avg": sum(ga) / len(ga) / XCDS,
                "mfma_util": sum(mf) / (SIMDS * sum(ga) / XCDS)})
out.sort(key=lambda d: -d["mfma_busy_cycles_avg"] * d["dispatches"])
for d in out:
    print(f"{d['mfma_util'] * 100:6.1f} %  x{d['dispatches']:4d}  {d['kernel']}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
