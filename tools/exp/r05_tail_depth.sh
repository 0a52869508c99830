#!/bin/bash
# (measurement of round 3, profiles/r03j_small_levels.txt; the switches it sets exist only with tools/exp/r03j_small_map_experiments.patch applied)
cd /root/repo
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
for d in 4 6 8; do run "per-tile split L5-7, ring depth $d" "EGONN_PF_BLOCKS=0 EGONN_COL_XCD=0 EGONN_TILE_DEPTH=$d"; done
