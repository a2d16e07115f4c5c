#!/bin/bash
# GPU box: factor probe, the dense tests, one timed dense call and its per-launch trace
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out
timeout -k 10 120 ./tools/factor_probe > $OUT/r06_factor_probe.txt 2>&1; echo "probe rc=$?" >> $OUT/r06_factor_probe.txt
cat $OUT/r06_factor_probe.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "dense" > $OUT/r06_dense_tests.txt 2>&1; echo "tests rc=$?" >> $OUT/r06_dense_tests.txt
tail -5 $OUT/r06_dense_tests.txt
CPU=0 YARDSTICK=0 timeout -k 10 300 python tools/bench_dense.py > $OUT/r06_dense_bench.txt 2>&1
tail -2 $OUT/r06_dense_bench.txt
cd /tmp && export TMPDIR=/tmp
CPU=0 YARDSTICK=0 REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/dense_trace -- python3 $ROOT/tools/bench_dense.py > $OUT/dense_trace.log 2>&1
python3 $ROOT/tools/dense_trace_steps.py $OUT/dense_trace table > $OUT/r06_dense_per_kernel_us.txt 2>&1
tail -3 $OUT/r06_dense_per_kernel_us.txt
rm -rf $OUT/dense_trace
