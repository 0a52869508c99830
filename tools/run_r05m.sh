for s in 3 4 5 6; do
python bench.py --no-extras --no-cpu-baseline --repeats 3 --streams $s 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s', d['value'], d['repeats']['scans_per_s'])"
done
GPU_MAX_HW_QUEUES=4 python bench.py --no-extras --no-cpu-baseline --repeats 3 --streams 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams 4 queues 4', d['value'], d['repeats']['scans_per_s'])"
