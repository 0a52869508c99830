#!/bin/bash
# LDS / issue counters of stand-alone sparse-conv layers:  ONLY=4 tools/pmc_lds.sh   (through gpurun from the repo root)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export F32ONLY=1 QUICK=1
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $line -d $OUT/pl$i -o s -- python $REPO/tools/bench_sconv.py > $OUT/pl$i.log 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS
GRBM_GUI_ACTIVE TA_TA_BUSY
PASSES
cd $REPO
python - <<'PY'
import sqlite3, glob, re, collections
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pl*/")):
    dbs = glob.glob(d + "**/*.db", recursive=True)
    if not dbs: continue
    c = sqlite3.connect(dbs[0])
    for name, cn, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        m = re.search(r"sconv_(rg|dma|wg|split|wide)_kernel<[^>]*>", name)
        if m: res[m.group(0).replace(" ", "")][cn] = v
for k, d in res.items():
    print(k)
    for cn in sorted(d): print(f"   {cn:34s} {d[cn]:16.0f}")
PY
rm -rf $OUT/pl*/
