#!/usr/bin/env python3
"""tools/isa_vmcnt0.py <file.s> [min block size]: per kernel, the basic blocks inside loops that contain `s_waitcnt vmcnt(0)` together with what
they load — the places where a wavefront drains its memory queue inside a hot loop (a default value on a conditionally loaded register, an LDS read
the compiler cannot tell apart from an LDS-DMA target, a global -> LDS copy through a register ...)."""
import re, sys, subprocess
s = open(sys.argv[1]).read()
minb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for m in re.finditer(r'^(_Z\S+):\s*;\s*@', s, re.M):
    name = m.group(1)
    end = s.find('.Lfunc_end', m.end())
    body = s[m.end():end].split('\n')
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l.strip())
        if mm: labels[mm.group(1)] = i
    # backward branches -> loop ranges
    loops = []
    for i, l in enumerate(body):
        mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if mm:
            tgt = mm.group(1) or mm.group(2)
            if tgt in labels and labels[tgt] < i: loops.append((labels[tgt], i))
    if not loops: continue
    hits = []
    for i, l in enumerate(body):
        t = l.strip()
        if t.startswith('s_waitcnt') and re.search(r'vmcnt\(0\)', t) and any(a <= i <= b for a, b in loops):
            # context: loads in the 40 instructions before
            ctx = [x.strip().split()[0] for x in body[max(0, i - 40):i] if x.strip() and not x.strip().startswith((';', '.'))]
            hits.append((i, sum(c.startswith(('global_load', 'buffer_load')) for c in ctx), sum(c.startswith('ds_') for c in ctx)))
    if hits:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:100]
        print(f"{dem}: {len(hits)} x vmcnt(0) inside loops; (line, loads in the 40 instr before, LDS ops before): {hits[:8]}")
