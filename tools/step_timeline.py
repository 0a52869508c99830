#!/usr/bin/env python
"""Ordered kernel timeline of ONE serial step out of a rocprofv3 --kernel-trace rocpd .db:
name, grid, workgroup, start offset, duration, gap to the previous kernel (all in microseconds).
usage: step_timeline.py <results.db> [anchor-kernel-substring] [which]   (a step = anchor to the next anchor launch;
`which` counts anchors from the end, default 3)"""
import sqlite3
import sys

db = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "points_to_keys_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = sqlite3.connect(db)
views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
src = "kernels" if "kernels" in views else [v for v in views if "kernel_dispatch" in v][0]
cols = [r[1] for r in c.execute(f"pragma table_info({src})")]
pick = lambda *names: next((n for n in names if n in cols), None)
name, start, end = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
gx, wx = pick("grid_x", "grid_size_x", "grid_size"), pick("workgroup_x", "workgroup_size_x", "workgroup_size")
if name is None or start is None:
    print("columns of", src, cols)
    sys.exit(1)
q = f"select {name}, {start}, {end}, {gx or 0}, {wx or 0} from {src} order by {start}"
rows = list(c.execute(q))
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
a, b = idx[-which], idx[-which + 1]
t0, prev = rows[a][1], rows[a][1]
tot = 0.0
for n, s, e, g, w in rows[a:b]:
    short = n.split("(")[0].replace("void egonn::", "").replace("egonn::", "")[:64]
    print(f"{(s - t0) / 1e3:9.1f}  +{(s - prev) / 1e3:6.1f}  {(e - s) / 1e3:8.2f}  grid {g:>9} wg {w:>5}  {short}")
    prev = e
    tot += (e - s) / 1e3
print(f"kernels {b - a}  busy {tot:.1f} us  span {(rows[b][1] - t0) / 1e3:.1f} us")
