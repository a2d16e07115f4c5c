#!/bin/bash
# GPU: the time-parallel tests after the boundary scan went in, then the whole GPU suite
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "time_parallel" > gpurun_out/r06_tp_tests.txt 2>&1 || { tail -40 gpurun_out/r06_tp_tests.txt; exit 1; }
tail -3 gpurun_out/r06_tp_tests.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1 || { tail -40 gpurun_out/r06_gpu_suite.txt; exit 1; }
tail -3 gpurun_out/r06_gpu_suite.txt
