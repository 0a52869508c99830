#!/usr/bin/env python
"""Dump the per-kernel statistics (`rocprofv3 --kernel-trace --stats`) of a rocpd .db into a CSV that is
small enough to commit under profiles/.   usage: rocpd_summary.py <results.db> <out.csv> [steps]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                      "order by total_duration desc"))
# a step launches points_to_keys_kernel exactly once: when it is in the trace, its call count IS the number of steps
for name, calls, *_ in rows:
    if "points_to_keys_kernel" in name:
        steps = calls
        break
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    hdr = ["kernel", "calls", "total_us", "avg_us", "percent"]
    if steps:
        hdr += ["calls_per_step", "us_per_step"]
    w.writerow(hdr)
    for name, calls, tot, avg, pct in rows:
        # durations are stored in ns by rocpd's top_kernels view when large; normalise to microseconds
        r = [name, calls, round(tot, 3), round(avg, 3), round(pct, 3)]
        if steps:
            r += [round(calls / steps, 2), round(tot / steps, 2)]
        w.writerow(r)
print(f"{len(rows)} kernels -> {out}")
