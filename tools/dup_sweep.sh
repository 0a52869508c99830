#!/bin/bash
# (measurement, profiles/r03j_small_levels.txt; EGONN_DUP_MASK exists only in a measurement build) marginal cost of a SECOND launch of
# an idempotent plan kernel: bit 1 the five sort_scatter passes, 2 pyramid_apply, 4 the four nbr27 launches, 8 rowgroup_build
cd /root/repo
for m in 0 1 2 4 8 15 0; do for s in 4 1; do
  echo -n "dup mask $m streams $s: "
  EGONN_DUP_MASK=$m timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --repeats 3 --streams $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
