#!/bin/bash
# GPU box: tools/pmc_kernels.sh <tag> <python script> [args] — one rocprofv3 --pmc pass per counter group over the script (no trace domains beside
# the counters; the program sits directly after `--`), then a per-kernel table of the medians over the dispatches (tools/pmc_table.py).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/g$i -- python3 "$@" > $out/g$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/g$i.log; }
done
python3 tools/pmc_table.py $out
