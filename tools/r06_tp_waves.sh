#!/bin/bash
# GPU: the scan's combination with eight wavefronts (17 .. 64 rows) against four: kernel durations, then the scan tests
set -e
mkdir -p gpurun_out
: > gpurun_out/r06_tp_waves.txt
for cfg in "12 0 SHO" "16 0 SHO" "20 0 SHO" "20 0 DRWCelerite"; do
  for wv in 8 4; do
    echo "## $cfg, $wv wavefronts per combination" >> gpurun_out/r06_tp_waves.txt
    PIORAN_TP_SCAN_WAVES=$wv bash tools/kstats.sh tp_w tools/prof_tp.py $cfg 2>&1 | head -4 >> gpurun_out/r06_tp_waves.txt
  done
done
cat gpurun_out/r06_tp_waves.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "time_parallel or quad" 2>&1 | tail -3
