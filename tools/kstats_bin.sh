#!/bin/bash
# GPU box: tools/kstats_bin.sh <tag> <binary> [args] — rocprofv3 kernel-trace summary of one native program
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf gpurun_out/$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- "$@" > gpurun_out/$tag.log 2>&1 || { tail -20 gpurun_out/$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:6]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
