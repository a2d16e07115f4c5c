#!/bin/bash
# GPU box: per-launch durations of dense_step_kernel with all roles and with each role alone (rocprofv3 kernel trace of tools/dense_roles.py)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/roles_trace -- python3 $ROOT/tools/dense_roles.py > $ROOT/gpurun_out/roles_trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$ROOT/gpurun_out/roles_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'diag0' in r['Kernel_Name']]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
# calls: 5 modes x (1 + 7) calls; take the last call of each mode
per_mode = 8
names = ["all", "crit", "strips", "bulk", "all2"]
for m, name in enumerate(names):
    i0 = starts[m * per_mode + per_mode - 1]
    i1 = starts[m * per_mode + per_mode] if m * per_mode + per_mode < len(starts) else len(rows)
    seq = [r for r in rows[i0:i1] if 'dense_step' in r['Kernel_Name']]
    print(name, len(seq), [round(dur(r), 1) for r in seq])
PY
rm -rf $ROOT/gpurun_out/roles_trace
