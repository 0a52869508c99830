#!/bin/bash
# (measurement, profiles/r03j_small_levels.txt section 11; needs tools/exp/r03k_lockstep_ksp.patch applied)
# wave sets (KSP) x column parts of the lock-step split kernel, stand-alone launches (tools/bench_sconv.py; ~5-9 us of weight packing
# per call included): variant 1142 = one wave set, 5142 = EGONN_SPLIT_KSP sets (clipped to the channel plan / LDS)
cd /root/repo
for parts in 0 1 2 4; do for k in 2 4; do
  echo "== EGONN_SPLIT_PARTS=$parts (0 = product rule) EGONN_SPLIT_KSP=$k"
  EGONN_SPLIT_PARTS=$parts EGONN_SPLIT_KSP=$k F32ONLY=1 AB=1142,5142 ONLY=${ONLY:-4,5,6,7,8,9,10,11,12,13,16} timeout 400 python tools/bench_sconv.py 2>&1 | grep "^kind" | awk '{printf "%s %s %s %s %s %-16s %7s us  err %s\n", $1,$2,$3,$4,$5,$6,$8,$(NF-5)}'
done; done
