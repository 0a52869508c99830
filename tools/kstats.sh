#!/bin/bash
# Per-kernel time of one bench step:  tools/kstats.sh <tag> [bench args]   (run through gpurun from the repo root)
# rocprofv3 --kernel-trace --stats of `bench.py --mode eager --streams 1` (one batch in flight: the sum of the column
# us_per_step is the serial GPU time of a step); writes gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=${1:-ks}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
STEPS=20
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- python $REPO/bench.py --mode eager --streams 1 --steps $STEPS --warmup 3 --repeats 1 --no-cpu-baseline "$@" > $OUT/${TAG}_kt.log 2>&1
cd $REPO
# bench.py runs warmup (3) + 5 profiled + 1 + 2 + STEPS timed + 5 exclusive eager steps = 36 steps
python tools/rocpd_summary.py $(find $OUT/${TAG}_kt -name '*.db' | head -1) $OUT/${TAG}_kernel_stats.csv $((STEPS + 16)) > /dev/null
rm -rf $OUT/${TAG}_kt
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv")))
tot = 0.0
for r in rows:
    n = r["kernel"].split("(")[0].replace("void egonn::", "").replace("egonn::", "")[:58]
    tot += float(r["us_per_step"])
    print(f"{n:60s} {float(r['calls_per_step']):6.2f} x {float(r['avg_us']):8.2f} = {float(r['us_per_step']):8.1f}")
print("sum us/step", round(tot, 1))
PY
