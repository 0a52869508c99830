#!/bin/bash
# (measurement of round 3, profiles/r03j_small_levels.txt; the switches it sets exist only with tools/exp/r03j_small_map_experiments.patch applied)
# ring depth of the lock-step split kernel on the small maps: stand-alone launches (tools/bench_sconv.py), fp32 maps
cd /root/repo
for d in 2 3 4; do
  echo "=== EGONN_SPLIT_DEPTH=$d (forced on every launch), AB=0 (product rule) / 1142 (split kernel on every level)"
  EGONN_SPLIT_DEPTH=$d F32ONLY=1 AB=0,1142 ONLY=${ONLY:-1,4,5,6,7,8,9,10,11,12,13,14,15,16} timeout 600 python tools/bench_sconv.py 2>&1 | grep "^kind"
done
