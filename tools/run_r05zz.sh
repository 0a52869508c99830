mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/measure.sh r05zz > gpurun_out/r05zz_measure.log 2>&1; tail -3 gpurun_out/r05zz_measure.log | cut -c1-300
bash tools/kstats.sh r05zz_serial > gpurun_out/r05zz_serial.txt 2>&1; tail -46 gpurun_out/r05zz_serial.txt
