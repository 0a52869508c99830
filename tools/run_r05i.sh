mkdir -p gpurun_out
python tools/check_bitwise_switches.py 2>&1 | tail -8
python -m pytest tests -m gpu -x -q > gpurun_out/r05i_tests.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r05i_tests.log
for v in "" "EGONN_NO_FUSED_LATERAL=1"; do
env $v python bench.py --no-extras --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'))"
done
