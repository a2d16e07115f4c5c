#!/usr/bin/env python3
"""Per-step durations of the dense kernels from a rocprofv3 kernel trace (csv).  usage: dense_trace_steps.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'diag0' in r['Kernel_Name']]
i0 = starts[-1]
seq = [r for r in rows[i0:] if 'dense_' in r['Kernel_Name']]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('syrk ', [round(dur(r), 1) for r in seq if 'syrk' in r['Kernel_Name']])
print('panel', [round(dur(r), 1) for r in seq if 'panel' in r['Kernel_Name']])
print('step ', [round(dur(r), 1) for r in seq if 'step' in r['Kernel_Name']])
print('gaps ', [round((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3, 1) for a, b in zip(seq, seq[1:])])
span = (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3
busy = sum(dur(r) for r in seq)
print(f'factor+finish span {span:.1f} us, kernel busy {busy:.1f} us, launches {len(seq)}')
