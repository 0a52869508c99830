mkdir -p gpurun_out
B="python bench.py --no-extras --no-cpu-baseline --repeats 3"
pick() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'), 'agg', d['roofline'].get('aggregate',{}).get('frac'), d['roofline'].get('aggregate',{}).get('serial_us_per_step'))"; }
$B 2>/dev/null | pick base
EGONN_TAIL=1 $B 2>/dev/null | pick tail
EGONN_SPLIT_MAX_LEVEL=5 $B 2>/dev/null | pick split5
EGONN_SPLIT_MAX_LEVEL=7 $B 2>/dev/null | pick split7
