#!/bin/bash
# Sweep of the offset-split rule through bench.py (from the repo root, through gpurun):
#   tools/ksplit_sweep.sh "ENV..." "ENV..."      each argument = one environment (space separated VAR=val), "" = defaults
# prints scans/s [repeats], one-batch graph latency, the aggregate conv roofline and the conv layers of levels >= 3
for v in "$@"; do
env $v python bench.py --no-extras --no-cpu-baseline --repeats 3 --steps 60 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'), 'agg', d['roofline']['aggregate']['frac'], d['roofline']['aggregate']['serial_us_per_step'])
rows=[r for r in d['roofline']['layers'] if r['layer'].startswith('sconv')]
def key(r):
    p=r['layer'].split('/'); return (int(p[1][1:]), p[2])
print('   ', '  '.join('%s/%s %.1f' % (r['layer'].split('/')[1], r['layer'].split('/')[2].replace('k3.',''), r['us']) for r in sorted(rows,key=key) if key(r)[0]>=3))
"
done
