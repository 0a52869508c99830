#!/bin/bash
# Per-kernel totals of ONE training step (rocprofv3 --kernel-trace --stats of tools/bench_train.py): tools/train_kstats.sh <tag>
set -u
TAG=${1:-train}; REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
STEPS=6
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- python $REPO/tools/bench_train.py --batch 32 --steps $STEPS > $OUT/${TAG}_kt.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find $OUT/${TAG}_kt -name '*.db' | head -1) $OUT/${TAG}_kernel_stats.csv $STEPS > /dev/null
rm -rf $OUT/${TAG}_kt
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv")))
tot = 0.0
for r in rows[:45]:
    n = r["kernel"].split("(")[0].replace("void egonn::", "").replace("egonn::", "")[:70]
    print(f"{n:72s} {float(r['calls_per_step']):7.2f} x {float(r['avg_us']):8.2f} = {float(r['us_per_step']):8.1f}")
for r in rows: tot += float(r["us_per_step"])
print("sum us/step (all launches / $STEPS steps, warm-up included)", round(tot, 1))
PY
