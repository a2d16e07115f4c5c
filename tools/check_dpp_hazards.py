#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing (hipcc -S --cuda-device-only): every *_dpp instruction must see at least
two wait states between the last VALU write of its DPP source (src0) and itself.  The compiler inserts those for code it
generates; inline-asm DPP sequences are our responsibility.  usage: check_dpp_hazards.py file.s [kernel-substring]"""
import re, sys

def regs(tok):
    tok = tok.strip().lstrip('-|').rstrip('|')
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()

def main():
    path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ''
    cur = None; body = []; bad = 0; checked = 0
    def scan(name, ins):
        nonlocal bad, checked
        for i, (op, args) in enumerate(ins):
            if not op.endswith('_dpp'): continue
            src = regs(args[1]) if len(args) > 1 else set()
            ws = 0
            for j in range(i - 1, -1, -1):
                pop, pargs = ins[j]
                if pop == 's_nop':
                    ws += int(pargs[0], 0) + 1; continue
                if pop.startswith('v_') and pargs and regs(pargs[0]) & src:
                    checked += 1
                    if ws < 2:
                        bad += 1; print(f'{name}: {op} {",".join(args)} only {ws} wait state(s) after {pop} {",".join(pargs)}')
                    break
                if not pop.startswith(('.', ';')) and not pop.endswith(':'): ws += 1
                if ws >= 2 and not (pop.startswith('v_')): pass
                if ws > 8: break
    for line in open(path):
        line = line.split(';')[0].strip()
        if not line: continue
        m = re.match(r'^(\S+):$', line)
        if m and not line.startswith('.L'):
            if cur and want in cur: scan(cur, body)
            cur = m.group(1); body = []; continue
        if line.startswith('.') or line.endswith(':'): continue
        parts = line.split(None, 1)
        op = parts[0]; args = [a.strip() for a in re.split(r',\s*(?![^\[]*\])', parts[1].split(' row_')[0].split(' quad_perm')[0])] if len(parts) > 1 else []
        body.append((op, args))
    if cur and want in cur: scan(cur, body)
    print(f'checked {checked} DPP source dependencies, {bad} hazard(s)')
    return 1 if bad else 0

if __name__ == '__main__':
    sys.exit(main())
