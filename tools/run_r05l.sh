mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05l_tests.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r05l_tests.log
python bench.py --no-extras --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 b16', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'))"
python bench.py --dtype bf16 --batch 64 --steps 30 --warmup 3 --no-extras --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 b64', d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'].get('graph_latency_ms'))"
bash tools/kstats.sh r05l_serial > gpurun_out/r05l_serial.txt 2>&1; grep -E "nbr27|plan_tables|rowgroup|eca_apply|dense_lds|local_heads|sum us" gpurun_out/r05l_serial.txt
