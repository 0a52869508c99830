#!/bin/bash
# (measurement of round 3, profiles/r03j_small_levels.txt; the switches it sets exist only with tools/exp/r03j_small_map_experiments.patch applied)
# instruction-cache behaviour of the small-map kernels (stand-alone L7 128->128 launches, exact fp32 vs per-tile split)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE" | head -40 > $OUT/icache_counters.txt
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  F32ONLY=1 QUICK=1 AB=0,2000 ONLY=13,12,11 EGONN_SPLIT_MAX_LEVEL=4 timeout 300 rocprofv3 --kernel-trace --pmc $line -d $OUT/ic$i -o s -- python $REPO/tools/bench_sconv.py > $OUT/ic$i.log 2>&1
done <<'PASSES'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY
TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
PASSES
cd $REPO
python - <<'PY'
import sqlite3, glob, re, collections
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/ic*/")):
    dbs = glob.glob(d + "**/*.db", recursive=True)
    if not dbs: continue
    c = sqlite3.connect(dbs[0])
    try:
        rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(d, e); continue
    for name, cn, v, n in rows:
        m = re.search(r"sconv_(rg|dma|dmasplit|wg|split|wide)_kernel<[^>]*>", name)
        if m: res[m.group(0).replace(" ", "")][cn] = (v, n)
for k, d in res.items():
    print(k)
    for cn in sorted(d): print(f"   {cn:34s} {d[cn][0]:16.0f}   (x{d[cn][1]})")
PY
cat $OUT/icache_counters.txt | cut -c1-160
rm -rf $OUT/ic*/
