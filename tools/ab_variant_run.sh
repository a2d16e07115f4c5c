#!/bin/bash
# GPU box: tools/ab_variant_run.sh <tag> [bench.py args] — the headline bench line with the shipped library and with tools/_ab/libpioran_hip_<tag>.so,
# alternating three times on the same box (kernel_ms of each run)
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  a=$(python3 bench.py --no-secondary --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])")
  b=$(PIORAN_HIP_LIB=tools/_ab/libpioran_hip_$tag.so python3 bench.py --no-secondary --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])")
  echo "shipped $a ms | $tag $b ms"
done
