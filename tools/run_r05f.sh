mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05f_tests.log 2>&1; echo "tests rc=$?" ; tail -4 gpurun_out/r05f_tests.log
for v in "" "EGONN_SORT_PAIRS=1"; do
env $v python bench.py --no-extras --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'))"
done
bash tools/kstats.sh r05f_serial > gpurun_out/r05f_serial.txt 2>&1; grep -E "sort|points_to|pyramid|sum us" gpurun_out/r05f_serial.txt
