#!/usr/bin/env python3
"""tools/isa_blocks.py <file.s> <mangled-name substring>: per basic block of one kernel, instruction counts by kind
(scratch traffic, AGPR copies, LDS, global loads, fp64 ops, waits) — where a kernel's spills sit."""
import re, sys, collections
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
start = m.end(); end = s.index('.Lfunc_end', start)
cur = 'entry'; stats = collections.OrderedDict()
for ln in s[start:end].split('\n'):
    t = ln.strip()
    mm = re.match(r'^(\.LBB\d+_\d+):', t)
    if mm: cur = mm.group(1); continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    d = stats.setdefault(cur, collections.Counter()); op = t.split()[0]
    d['n'] += 1
    for k, pre in (('sst', 'scratch_store'), ('sld', 'scratch_load'), ('acc', 'v_accvgpr'), ('ds', 'ds_'), ('gld', 'global_load'), ('wait', 's_waitcnt'), ('dpp', 'v_mov_b32_dpp')):
        if op.startswith(pre): d[k] += 1
    if 'f64' in op: d['f64'] += 1
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for k, v in stats.items():
    if v['n'] >= minn: print(k, dict(v))
