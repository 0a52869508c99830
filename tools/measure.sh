#!/bin/bash
# One measurement round on the GPU box:  tools/measure.sh <tag>   (run through gpurun from the repo root)
# bench line, per-layer table, rocprofv3 kernel stats, and the two PMC passes (FETCH_SIZE / WRITE_SIZE) of the
# same bench command.  Everything lands in gpurun_out/<tag>_*; copy what should be kept into profiles/.
set -u
TAG=${1:-meas}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py ${BENCH_ARGS:-} --layer-table $OUT/${TAG}_layers.json > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -1 $OUT/${TAG}_bench.json | cut -c1-600
BENCH="python $REPO/bench.py ${BENCH_ARGS:-} --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- $BENCH > $OUT/${TAG}_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_pf -o pf -- $BENCH > $OUT/${TAG}_pf.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_pw -o pw -- $BENCH > $OUT/${TAG}_pw.log 2>&1
cd $REPO
KT=$(find $OUT/${TAG}_kt -name '*.db' | head -1); PF=$(find $OUT/${TAG}_pf -name '*.db' | head -1); PW=$(find $OUT/${TAG}_pw -name '*.db' | head -1)
python tools/rocpd_summary.py $KT $OUT/${TAG}_kernel_stats.csv 34
python tools/pmc_traffic.py $PF $PW $OUT/${TAG}_traffic.json
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pf $OUT/${TAG}_pw
