#!/bin/bash
# Build container: an EXPERIMENT build of the library with s_memtime stamps in the combination kernels (-DPIORAN_TP_STAMP) as tools/experiments/libpioran_stamp.so
# (git-ignored; travels to the GPU box).  GPU box: PIORAN_HIP_LIB=tools/experiments/libpioran_stamp.so python tools/tp_combine_stamps.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
python pioran.jl_amd/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DPIORAN_TP_STAMP -c pioran.jl_amd/csrc/celerite_tp.hip -o /tmp/celerite_tp_stamp.o
objs=$(ls pioran.jl_amd/_obj/*.o | grep -v celerite_tp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libpioran_stamp.so $objs /tmp/celerite_tp_stamp.o
ls -la tools/experiments/libpioran_stamp.so
