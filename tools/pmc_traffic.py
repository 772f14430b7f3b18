#!/usr/bin/env python3
"""Per-launch HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> [out.json]

Units/corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced read stream, so it is doubled; WRITE_SIZE is exact for 16 B/lane
streaming stores.  The two passes are separate runs of the same command (the TCC block cannot hold both counters).
"""
import csv
import json
import sys


def avg(path, counter, needle):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            vals.setdefault(r["Dispatch_Id"], 0.0)
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = list(vals.values())
    return (sum(v) / len(v) if v else float("nan")), len(v)


fetch, nf = avg(sys.argv[1], "FETCH_SIZE", sys.argv[3])
write, nw = avg(sys.argv[2], "WRITE_SIZE", sys.argv[3])
out = {"kernel": sys.argv[3], "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
       "hbm_read_bytes_per_launch": 2.0 * fetch * 1024, "hbm_write_bytes_per_launch": write * 1024,
       "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE as is, KiB -> bytes"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
