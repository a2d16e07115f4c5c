#!/usr/bin/env python3
"""Instruction mix of the hot loop of each celerite_scan_kernel instantiation (reads a hipcc -S dump).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o scan.s celerite_scan.hip; isa_loop_stats.py scan.s [filter]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "Lb1E"   # substring of the mangled name, e.g. Li3ELi2ELi7ELb1ELi1ELb1E (headline kernel)
for m in re.finditer(r'^(_ZN12_GLOBAL__N_120celerite_scan_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d)E\w*EEv10ScanParams):', s, re.M):
    name = m.group(1)
    if flt not in name: continue
    i = m.start(); j = s.index('.Lfunc_end', i)
    body = s[i:j].split('\n')
    labels = {}
    for idx, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm: labels[mm.group(1)] = idx
    loops = []
    for idx, l in enumerate(body):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < idx:
            loops.append((idx - labels[mm.group(1)], labels[mm.group(1)], idx))
    loops.sort(reverse=True)
    n, a, b = loops[0]
    c = Counter()
    for l in body[a:b + 1]:
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'): continue
        c[l.split()[0]] += 1
    k = s.find('.name:           ' + name)
    meta = s[k - 1500:k + 800] if k > 0 else ''
    vg = re.search(r'\.vgpr_count:\s+(\d+)', meta); ag = re.search(r'\.agpr_count:\s+(\d+)', meta)
    print(f"rpl{m.group(2)} cbr{m.group(3)} nsrc{m.group(4)} tab{m.group(5)} minw{m.group(6)}: loop={sum(c.values())} vgpr={vg.group(1) if vg else '?'} agpr={ag.group(1) if ag else '?'} | " +
          " ".join(f"{k}:{v}" for k, v in c.most_common(12)))
