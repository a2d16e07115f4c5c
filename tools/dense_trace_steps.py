#!/usr/bin/env python3
"""Per-launch durations of the dense kernels of the LAST pioran_dense_nll call in a rocprofv3 kernel trace (csv).
usage: dense_trace_steps.py <dir> [table]     (table: one line per launch — kernel, start, duration, workgroups — like profiles/r02_dense_*_per_kernel_us.txt)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'diag0' in r['Kernel_Name']]
i0 = starts[-1]
# the covariance build precedes diag0: include the (up to three) dense_build launches right before it
j0 = i0
while j0 > 0 and 'dense_build' in rows[j0 - 1]['Kernel_Name']: j0 -= 1
seq = [r for r in rows[j0:] if 'dense_' in r['Kernel_Name']]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
def short(n):
    for k in ('build_diag_batch', 'build_fast_batch', 'build_fast', 'build', 'diag0', 'panel', 'syrk', 'step', 'finish'):
        if 'dense_' + k in n: return k + (n[n.index('<'):n.index('>') + 1] if '<' in n and k in ('syrk', 'step', 'panel') else '')
    return n[:30]
if len(sys.argv) > 2 and sys.argv[2] == 'table':
    t0 = int(seq[0]['Start_Timestamp'])
    print('# kernel              start_us  duration_us  workgroups')
    for r in seq:
        wg = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 256)) or 256))
        print(f"{short(r['Kernel_Name']):20s} {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {dur(r):10.1f} {wg:8d}")
    fac = [r for r in seq if any(k in r['Kernel_Name'] for k in ('diag0', 'panel', 'syrk', 'step'))]
    span = (int(fac[-1]['End_Timestamp']) - int(fac[0]['Start_Timestamp'])) / 1e3
    print(f'# factorisation: {len(fac)} launches, span {span:.1f} us (diag0 .. last step), whole sequence {len(seq)} launches')
else:
    print('syrk ', [round(dur(r), 1) for r in seq if 'syrk' in r['Kernel_Name']])
    print('panel', [round(dur(r), 1) for r in seq if 'panel' in r['Kernel_Name']])
    print('step ', [round(dur(r), 1) for r in seq if 'step' in r['Kernel_Name']])
    print('gaps ', [round((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3, 1) for a, b in zip(seq, seq[1:])])
    span = (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3
    busy = sum(dur(r) for r in seq)
    print(f'factor+finish span {span:.1f} us, kernel busy {busy:.1f} us, launches {len(seq)}')
