mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05a_tests.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r05a_tests.log
python tools/bench_tail.py 16 > gpurun_out/r05a_tail_ab.txt 2>&1; tail -8 gpurun_out/r05a_tail_ab.txt
python tools/tail_trace.py 16 > gpurun_out/r05a_tail_trace.txt 2>&1; tail -24 gpurun_out/r05a_tail_trace.txt
python bench.py --no-extras --no-cpu-baseline --repeats 3 --layer-table gpurun_out/r05a_layers.json > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; tail -1 gpurun_out/r05a_bench.json | cut -c1-400
bash tools/kstats.sh r05a_serial > gpurun_out/r05a_serial.txt 2>&1; tail -70 gpurun_out/r05a_serial.txt
