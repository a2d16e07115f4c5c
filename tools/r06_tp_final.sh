#!/bin/bash
# GPU: the three time-parallel sweeps on the final kernels (eight wavefronts per combination from 17 rows on), then the whole GPU suite
set -e
mkdir -p gpurun_out
timeout -k 10 300 python tools/tp_scan_sweep.py > gpurun_out/r06_tp_scan_sweep.txt 2>&1
timeout -k 10 300 python tools/tp_scan_sweep.py wide > gpurun_out/r06_tp_scan_sweep_wide.txt 2>&1
timeout -k 10 400 python tools/tp_scan_batch_sweep.py 2048 4096 10000 > gpurun_out/r06_tp_scan_batch_sweep3.txt 2>&1
tail -3 gpurun_out/r06_tp_scan_sweep.txt | cut -c1-400
python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1 || { tail -40 gpurun_out/r06_gpu_suite.txt; exit 1; }
tail -3 gpurun_out/r06_gpu_suite.txt
