mkdir -p gpurun_out
bash tools/measure.sh r05z > gpurun_out/r05z_measure.log 2>&1; tail -3 gpurun_out/r05z_measure.log
bash tools/kstats.sh r05z_serial > gpurun_out/r05z_serial.txt 2>&1; tail -50 gpurun_out/r05z_serial.txt
BENCH_ARGS="--dtype bf16 --batch 64 --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-extras" bash tools/measure.sh r05zb_bf16_b64 > gpurun_out/r05zb_measure.log 2>&1; tail -2 gpurun_out/r05zb_measure.log
