#!/bin/bash
# (measurement of round 3, profiles/r03j_small_levels.txt; the switches it sets exist only with tools/exp/r03j_small_map_experiments.patch applied)
# which levels run which split kernel: bench.py throughput (4 batches in flight) and one-batch latency
cd /root/repo
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'graph_latency_ms', d['latency'].get('graph_latency_ms'))"; }
run "old: lock-step split <= L4, exact fp32 above" "EGONN_SPLIT_MAX_LEVEL=4"
run "new default: lock-step <= L4, per-tile split L5-7" "X=1"
run "lock-step <= L3, per-tile split L4-7" "EGONN_SPLIT_TILE_MIN_LEVEL=4"
run "lock-step <= L2, per-tile split L3-7" "EGONN_SPLIT_TILE_MIN_LEVEL=3"
run "lock-step <= L4, per-tile split L5 only, exact L6-7" "EGONN_SPLIT_MAX_LEVEL=5"
run "old again" "EGONN_SPLIT_MAX_LEVEL=4"
