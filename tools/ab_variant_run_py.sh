#!/bin/bash
# GPU box: tools/ab_variant_run_py.sh <tag> <script.py> [args] — last line of the script with the shipped library and with tools/_ab/libpioran_hip_<tag>.so,
# alternating three times on the same box
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  a=$(python3 "$@" 2>/dev/null | tail -1)
  b=$(PIORAN_HIP_LIB=tools/_ab/libpioran_hip_$tag.so python3 "$@" 2>/dev/null | tail -1)
  echo "shipped $a | $tag $b"
done
