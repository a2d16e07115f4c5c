#!/usr/bin/env python3
"""Instruction counts of the hot loop of one instantiation of celerite_scan_kernel, from the compiler's ISA listing
(hipcc -S --cuda-device-only).  The loop is recognised by its number of v_rcp_f64 (1 = step by step, 2 = two-step, 3 = three-step).
usage: python tools/scan_loop_counts.py <listing.s> <substring of the mangled kernel name> [rcp count]
e.g.   ... ILi3ELi2ELi7ELb1ELi1ELb1ELb0ELi0ELb1ELi0ELb0ELb0E 2     (rpl3_cbr2_nsrc7_p, shared table, two-step: the headline kernel)"""
import re, sys
from collections import Counter

s = open(sys.argv[1]).read()
want = sys.argv[2]
nrcp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for name in re.findall(r'^(_ZN12_GLOBAL__N_120celerite_scan_kernel\w+):', s, re.M):
    if want not in name: continue
    i = s.index('\n' + name + ':'); j = s.index('.Lfunc_end', i)
    body = s[i:j].split('\n')
    labels = {}
    for idx, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = idx
    for idx, l in enumerate(body):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < idx:
            a = labels[m.group(1)]
            L = [x.strip().split(';')[0].strip() for x in body[a:idx + 1] if x.strip() and not x.strip().startswith((';', '.'))]
            c = Counter(x.split()[0] for x in L)
            if c['v_rcp_f64_e32'] == nrcp:
                valu = sum(v for k, v in c.items() if k.startswith('v_'))
                fma = sum(v for k, v in c.items() if k.startswith(('v_fma_f64', 'v_fmac_f64')))
                print(f"{name[40:]}: loop {len(L)} instructions, VALU {valu} ({valu / nrcp:.1f} per step), fma/fmac {fma}, v_mul_f64 {c['v_mul_f64']}, "
                      f"v_mov_b64 {c['v_mov_b64_e32']}, ds_bpermute {c['ds_bpermute_b32']}, buffer loads {sum(v for k, v in c.items() if k.startswith('buffer_load'))}, "
                      f"scalar {sum(v for k, v in c.items() if k.startswith('s_'))}, scratch {sum(v for k, v in c.items() if 'scratch' in k)}")
