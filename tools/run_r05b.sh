mkdir -p gpurun_out
python -m pytest tests/test_gpu_tail.py -x -q > gpurun_out/r05b_tests.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r05b_tests.log
python tools/bench_tail.py 16 > gpurun_out/r05b_tail_ab.txt 2>&1; tail -7 gpurun_out/r05b_tail_ab.txt
python tools/tail_trace.py 16 > gpurun_out/r05b_tail_trace.txt 2>&1; tail -22 gpurun_out/r05b_tail_trace.txt
