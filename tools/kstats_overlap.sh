#!/bin/bash
# Per-kernel time of the benchmark step with four batches in flight (graph mode):  tools/kstats_overlap.sh <tag> [env...]
# rocprofv3 --kernel-trace --stats of the default bench command; prints calls/step x avg us under overlap.
set -u
TAG=${1:-ko}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- python $REPO/bench.py --steps 40 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_kt.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find $OUT/${TAG}_kt -name '*.db' | head -1) $OUT/${TAG}_kernel_stats.csv > /dev/null
rm -rf $OUT/${TAG}_kt
tail -1 $OUT/${TAG}_kt.log | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scans/s', d['value'])"
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv")))
tot = 0.0
for r in rows[:40]:
    n = r["kernel"].split("(")[0].replace("void egonn::", "").replace("egonn::", "")[:62]
    tot += float(r["us_per_step"])
    print(f"{n:64s} {float(r['calls_per_step']):6.2f} x {float(r['avg_us']):8.2f} = {float(r['us_per_step']):8.1f}")
print("sum us/step (top 40)", round(tot, 1))
PY
