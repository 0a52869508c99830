#!/bin/bash
# The measurement set of a round (through gpurun from the repo root):  tools/measure_round.sh <tag>
#   smoke, full default bench line (+ extras, cpu_baseline), per-layer table, rocprofv3 kernel stats, PMC traffic (tools/measure.sh),
#   serial kernel stats of one eager step (tools/kstats.sh), the bitwise check of the fusion switches, and the bf16 batch-64 set.
# Everything lands in gpurun_out/<tag>*; copy what should be kept into profiles/.
TAG=${1:-rXX}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
python tools/check_bitwise_switches.py 2>&1 | grep "bitwise equal"
bash tools/measure.sh ${TAG} > gpurun_out/${TAG}_measure.log 2>&1; tail -3 gpurun_out/${TAG}_measure.log | cut -c1-300
bash tools/kstats.sh ${TAG}_serial > gpurun_out/${TAG}_serial.txt 2>&1; tail -46 gpurun_out/${TAG}_serial.txt
BENCH_ARGS="--dtype bf16 --batch 64 --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-extras" bash tools/measure.sh ${TAG}b_bf16_b64 > gpurun_out/${TAG}b_measure.log 2>&1; tail -2 gpurun_out/${TAG}b_measure.log | cut -c1-300
