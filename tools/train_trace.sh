#!/bin/bash
# Per-launch durations of the training-step kernels (rocprofv3 --kernel-trace of tools/bench_train.py): for the kernels named in
# $1 (regex) the grid size and duration of every launch of the LAST step.   tools/train_trace.sh 'col_stats|affine' (through gpurun)
set -u
PAT=${1:-col_stats}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/tt -o tt -- python $REPO/tools/bench_train.py --steps 3 > $OUT/tt.log 2>&1
cd $REPO
python - "$PAT" <<'PY'
import sqlite3, glob, re, sys
pat = re.compile(sys.argv[1])
db = glob.glob("gpurun_out/tt/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
sym = [t for t in tabs if "kernel_symbol" in t]
print("table", kt, cols[:20])
q = f"select * from {kt} order by start"
rows = list(c.execute(q))
ci = {n: i for i, n in enumerate(cols)}
names = {}
if sym:
    scol = [r[1] for r in c.execute(f"pragma table_info({sym[0]})")]
    for r in c.execute(f"select * from {sym[0]}"):
        d = dict(zip(scol, r)); names[d.get("id")] = d.get("display_name") or d.get("kernel_name")
out = []
for r in rows:
    nm = names.get(r[ci["kernel_id"]], str(r[ci["kernel_id"]])) if "kernel_id" in ci else ""
    out.append((nm, r[ci["start"]], r[ci["end"]], r[ci.get("grid_size_x", ci.get("grid_x", 0))], r[ci.get("workgroup_size_x", ci.get("workgroup_x", 0))]))
# the last step: everything after the last points/coords key kernel
last = max(i for i, o in enumerate(out) if "to_keys" in o[0])
tot = {}
for nm, s, e, gx, wx in out[last:]:
    if pat.search(nm):
        print(f"{(e - s) / 1e3:9.1f} us  grid {gx:>9}  wg {wx:>4}  {nm[:70]}")
    k = nm.split("(")[0][:60]
    tot[k] = tot.get(k, 0) + (e - s) / 1e3
print("--- totals of the last step")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{v:9.1f} us  {k}")
print("sum", round(sum(tot.values()), 1), "us; span", round((out[-1][2] - out[last][1]) / 1e3, 1), "us")
PY
rm -rf $OUT/tt
