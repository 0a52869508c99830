#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes — they do
not fit one pass on gfx950, MI355X_MICROARCH.md §rocprofv3 PMC slots).

    pmc_traffic.py <fetch.db> <write.db> <out.json>

Both counters are in KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly 1/2 of the
bytes of wide (16 B/lane) coalesced reads, so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes; the gathers
of this path are 16 B/lane loads of whole 128-B rows, the pattern the correction was calibrated on.  Averages
are per launch over all launches of a kernel template (like roofline.achieved in bench.py)."""
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, val, n in c.execute("select kernel_name, avg(value), count(*) from counters_collection "
                                  "where counter_name=? group by kernel_name", (counter,)):
        m = re.search(r"egonn::(\w+)(<[^>]*>)?", name)
        if m:
            key = m.group(1) + (m.group(2) or "").replace(" ", "")
            out[key] = (val, n)
    return out


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
    res[k] = {"launches": f.get(k, (0, 0))[1], "FETCH_SIZE_KiB_avg": round(fk, 1), "WRITE_SIZE_KiB_avg": round(wk, 1),
              "traffic_bytes_per_launch": round((2.0 * fk + wk) * 1024.0)}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(f"{len(res)} kernels -> {sys.argv[3]}")
