#!/bin/bash
# PMC passes over the window-resident kernel alone (tools/bench_sconv.py ONLY=1, fp32, product choice); per-kernel averages
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export ONLY=${ONLY:-1} F32ONLY=1 QUICK=1
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $line -d $OUT/pw$i -o s -- python $REPO/tools/bench_sconv.py > $OUT/pw$i.log 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS SQ_WAVES SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT
SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_WAIT_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT
TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES
TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY
TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ
PASSES
cd $REPO
python - <<'PY'
import sqlite3, glob, re, collections
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pw*/")):
    dbs = glob.glob(d + "**/*.db", recursive=True)
    if not dbs: continue
    c = sqlite3.connect(dbs[0])
    try:
        for name, cn, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            m = re.search(r"sconv_(rg|dma|wg|split|wide|win)_kernel<[^>]*>", name)
            if m: res[m.group(0).replace(" ", "")][cn] = v
    except Exception as e:
        print("pass", d, "failed:", e)
for k, d in res.items():
    print(k)
    for cn in sorted(d): print(f"   {cn:34s} {d[cn]:16.0f}")
PY
rm -rf $OUT/pw*/
