mkdir -p gpurun_out
python tools/check_bitwise_switches.py 2>&1 | tail -3
python -m pytest tests -m gpu -x -q > gpurun_out/r05h_tests.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r05h_tests.log
for v in "" "EGONN_NO_PRESPLIT=1"; do
env $v python bench.py --no-extras --no-cpu-baseline --repeats 3 --layer-table gpurun_out/r05h_layers_${v:-presplit}.json 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'), 'agg', d['roofline']['aggregate']['frac'], d['roofline']['aggregate']['serial_us_per_step'])"
done
python - <<'PY'
import json,glob
a=json.load(open('gpurun_out/r05h_layers_presplit.json')); b=json.load(open('gpurun_out/r05h_layers_EGONN_NO_PRESPLIT=1.json'))
bm={r['layer']:r for r in b['rows']}
for r in a['rows']:
    if 'conv2' in r['layer']: print(r['layer'], r['us'], 'vs', bm[r['layer']]['us'], 'hbm', r['hbm_frac'])
PY
