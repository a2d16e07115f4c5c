#!/bin/bash
# GPU box: the round's remaining evidence on the final kernels (outputs under gpurun_out/r06/extra2/): kernel stats and counters of the time-parallel family and of the
# many-chain reverse mode, the quad-truth table, the stored reference values through every family, the dense per-launch table.  PART=a|b|c (a call is limited to 20 minutes).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
X=$ROOT/gpurun_out/r06/extra2
mkdir -p $X
PART=${PART:-a}
if [[ $PART == a ]]; then
  : > $X/time_parallel_kernel_stats.txt
  for cfg in "8 0 SHO 1" "12 0 SHO 1" "20 0 SHO 1" "20 0 DRWCelerite 1" "20 0 SHO 4" "20 0 SHO 8"; do
    echo "## prof_tp.py $cfg (components, segments 0 = automatic, basis, draws): 20 forced evaluations at N = 1e4" >> $X/time_parallel_kernel_stats.txt
    bash tools/kstats.sh tp_ks tools/prof_tp.py $cfg >> $X/time_parallel_kernel_stats.txt 2>&1
  done
  tail -12 $X/time_parallel_kernel_stats.txt
  bash tools/pmc_kernels.sh tp20 tools/prof_tp.py 20 0 > $X/pmc_time_parallel_kernels.txt 2>&1
  tail -3 $X/pmc_time_parallel_kernels.txt | cut -c1-300
  timeout -k 10 300 python tools/quad_truth_gpu.py > $X/quad_truth.txt 2>&1; tail -5 $X/quad_truth.txt
fi
if [[ $PART == b ]]; then
  bash tools/kstats.sh tg tools/prof_tile_grad.py > $X/gradient_tile_kernel_stats.txt 2>&1
  bash tools/kstats.sh tgcd tools/prof_tile_grad.py cd >> $X/gradient_tile_kernel_stats.txt 2>&1
  cat $X/gradient_tile_kernel_stats.txt | cut -c1-200
  bash tools/pmc_kernels.sh tgcd tools/prof_tile_grad.py cd > $X/pmc_gradient_kernels.txt 2>&1
  tail -3 $X/pmc_gradient_kernels.txt | cut -c1-300
fi
if [[ $PART == c ]]; then
  timeout -k 10 900 python tools/validate_reference_values.py > $X/reference_values_by_family.txt 2>&1; tail -8 $X/reference_values_by_family.txt
  bash tools/r06_dense_check.sh > $X/dense_check.log 2>&1; tail -5 $X/dense_check.log
  cp gpurun_out/r06_dense_per_kernel_us.txt $X/dense_per_kernel_us.txt
fi
