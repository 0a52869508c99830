#!/bin/bash
# (measurement of round 3, profiles/r03j_small_levels.txt; the switches it sets exist only with tools/exp/r03j_small_map_experiments.patch applied)
# marginal cost of each kernel class in the overlapped step: bench.py with EGONN_SKIP_MASK (model.hip), results of a masked run are garbage
cd /root/repo
mkdir -p gpurun_out
for m in 0 1 2 4 8 16 32 64 128 256 30 62 511 0; do
  for s in 4 1; do
    echo -n "mask $m streams $s: "
    EGONN_SKIP_MASK=$m timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --repeats 3 --streams $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
