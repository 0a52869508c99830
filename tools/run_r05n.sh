for v in "" "EGONN_NO_FUSED_ECA=1" "EGONN_POOL_TWO_LAUNCHES=1" "EGONN_NO_FUSED_ECA=1 EGONN_POOL_TWO_LAUNCHES=1"; do
env $v python bench.py --no-extras --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'))"
done
