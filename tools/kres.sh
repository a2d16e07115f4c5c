#!/bin/bash
# tools/kres.sh <file.hip> [name filter]: compile one source of the library for gfx950 with -save-temps and print every kernel's
# registers / spills / LDS / scratch from the code object's metadata.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
f=$1; stem=$(basename "$f" .hip); filt=${2:-.}
mkdir -p /tmp/kres && cd /tmp/kres
# everything under /tmp/kres: the product's objects (pioran.jl_amd/_obj, what libpioran_hip.so is linked from) are never touched by this tool
# (up to round 4 it compiled INTO _obj with -save-temps, and the shipped library was linked from that object)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -S --cuda-device-only "$ROOT/pioran.jl_amd/csrc/$stem.hip" -o /tmp/kres/$stem.s 2>&1 | grep -v "^$" || true
python3 - "$stem" "$filt" <<'PY'
import re, sys, subprocess
stem, filt = sys.argv[1], sys.argv[2]
s = open(f'/tmp/kres/{stem}.s').read()
md = s[s.index('amdhsa.kernels'):]
for b in md.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', b).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if not re.search(filt, dem): continue
    print(f"{dem[:110]:110s} agpr {b.split(chr(10))[0].strip():>3s} vgpr {g('vgpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")
PY
