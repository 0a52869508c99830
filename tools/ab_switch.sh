#!/bin/bash
# A/B of one measurement switch through bench.py (from the repo root, through gpurun):  tools/ab_switch.sh "ENV=1" ["ENV2=1" ...]
# prints scans/s [repeats], one-batch graph latency and the aggregate conv roofline with and without every switch
python tools/check_bitwise_switches.py 2>&1 | grep "bitwise equal"
for v in "" "$@"; do
env $v python bench.py --no-extras --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'), 'agg', d['roofline']['aggregate']['frac'], d['roofline']['aggregate']['serial_us_per_step'])"
done
