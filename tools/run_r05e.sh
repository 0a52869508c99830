mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05e_tests.log 2>&1; echo "tests rc=$?" ; tail -4 gpurun_out/r05e_tests.log
python bench.py --no-extras --no-cpu-baseline --repeats 3 > gpurun_out/r05e_bench.json 2> gpurun_out/r05e_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05e_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['repeats']['scans_per_s'], 'lat', d['latency'], 'agg', d['roofline'].get('aggregate'))
print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
PY
bash tools/kstats.sh r05e_serial > gpurun_out/r05e_serial.txt 2>&1; tail -48 gpurun_out/r05e_serial.txt
