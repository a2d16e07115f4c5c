#!/bin/bash
# GPU box: tools/ab_variant_run_grad.sh <tag> [drw] [cd] — value + gradient of 4096 chains with the shipped library and with tools/_ab/libpioran_hip_<tag>.so,
# alternating three times on the same box
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  a=$(python3 tools/time_tile_grad.py "$@" 2>/dev/null | tail -1)
  b=$(PIORAN_HIP_LIB=tools/_ab/libpioran_hip_$tag.so python3 tools/time_tile_grad.py "$@" 2>/dev/null | tail -1)
  echo "shipped $a ms | $tag $b ms"
done
