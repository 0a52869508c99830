#!/bin/bash
# (measurement of round 3, profiles/r03j_small_levels.txt; the switches it sets exist only with tools/exp/r03j_small_map_experiments.patch applied)
# marginal cost of one more (empty) kernel launch per step, one batch in flight and four
cd /root/repo
for n in 0 50 100 0; do for s in 4 1; do
  echo -n "dummy launches $n streams $s: "
  EGONN_DUMMY_LAUNCHES=$n timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --repeats 3 --streams $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
