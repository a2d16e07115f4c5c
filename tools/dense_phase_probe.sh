#!/bin/bash
# timing-only probe (results are wrong by construction): which phase of dense_panel_kernel costs what
set -e
cd "$(dirname "$0")/.."
for flag in "" "-DPANEL_SKIP1" "-DPANEL_SKIP2" "-DPANEL_SKIP1 -DPANEL_SKIP2"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flag -c pioran.jl_amd/csrc/dense.hip -o /tmp/dense_probe.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o pioran.jl_amd/libpioran_hip.so pioran.jl_amd/_obj/celerite_scan.o pioran.jl_amd/_obj/celerite_fallback.o pioran.jl_amd/_obj/table.o /tmp/dense_probe.o pioran.jl_amd/_obj/capi.o
  echo "== flags: [$flag]"
  CPU=0 REPS=5 python tools/bench_dense.py 2>&1 | tail -1 | cut -c1-130
done
