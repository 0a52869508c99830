#!/bin/bash
# L2 / L1 request counters of the sparse-conv layers (tools/bench_sconv.py, one variant) -> gpurun_out/pmc_l2_*.csv
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
VARS=0 timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/l2a -o a -- python $REPO/tools/bench_sconv.py > $OUT/l2a.log 2>&1
VARS=0 timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/l2b -o b -- python $REPO/tools/bench_sconv.py > $OUT/l2b.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob, re
for tag in ("l2a", "l2b"):
    dbs = glob.glob(f"gpurun_out/{tag}/**/*.db", recursive=True)
    if not dbs: print(tag, "no db"); continue
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    for name, cn, v, n in rows:
        if "sconv" in name:
            m = re.search(r"sconv_\w+<[^>]*>", name)
            print(tag, m.group(0) if m else name[:40], cn, round(v), n)
PY
rm -rf $OUT/l2a $OUT/l2b
