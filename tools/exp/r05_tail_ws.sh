#!/bin/bash
# (measurement of round 3, profiles/r03j_small_levels.txt; the switches it sets exist only with tools/exp/r03j_small_map_experiments.patch applied)
cd /root/repo
echo "== correctness (stand-alone, vs the plain kernel): AB=3000"
EGONN_SPLIT_MAX_LEVEL=4 F32ONLY=1 AB=0,3000 ONLY=10,11,12,13,14,15 timeout 300 python tools/bench_sconv.py 2>&1 | grep "^kind" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$17,$18,$19,$20}'
run() { echo "== $1"; env $2 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 100 --repeats 3 --layer-table /tmp/lt.json 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scans/s', d['value'], 'graph_latency_ms', d['latency'].get('graph_latency_ms'))
t=json.load(open('/tmp/lt.json'))
tot=0
for r in sorted(t['rows'], key=lambda r: r['layer'].split('/')[1:]):
    l=r['layer']
    if any(x in l for x in ('/L5','/L6','/L7')): print('   %-52s %6.1f' % (l, r['us'])); tot+=r['us']
print('   sum', round(tot,1))
"; }
run "old: exact fp32 on L5-7" "EGONN_SPLIT_MAX_LEVEL=4"
run "weight-stationary split L5-7 (default)" "X=1"
for c in 4 8 30; do run "weight-stationary, chunks $c" "EGONN_WS_CHUNKS=$c"; done
