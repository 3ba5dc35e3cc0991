#!/usr/bin/env python3
"""Instruction mix per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only): total, fp64 VALU, other VALU, scalar,
LDS, memory, scratch, accvgpr moves, plus the largest backward-branch loop of each kernel.  usage: tools/isa_mix.py file.s [filter]"""
import re, sys
from collections import Counter

def kind(i):
    if i.startswith('scratch_'): return 'scratch'
    if i.startswith('v_accvgpr'): return 'accvgpr'
    if i.startswith('v_mfma'): return 'mfma'
    if i.startswith('v_') and 'f64' in i: return 'f64'
    if i.startswith('v_'): return 'v_other'
    if i.startswith('s_'): return 's_'
    if i.startswith('ds_'): return 'ds'
    if i.startswith(('global_', 'buffer_', 'flat_')): return 'mem'
    return 'other'

src = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
name, start = None, 0
funcs = []
for n, l in enumerate(src):
    m = re.match(r'^(_Z\w+):', l)
    if m: name, start = m.group(1), n
    elif l.startswith('.Lfunc_end') and name:
        funcs.append((name, start, n)); name = None
for name, a, b in funcs:
    if flt not in name: continue
    body = src[a + 1:b]
    ins = []      # (line index, mnemonic)
    labels = {}
    for n, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith((';', '//')): continue
        if t.endswith(':') and not t.startswith('.L') is False or re.match(r'^\.LBB\d+_\d+:', t):
            labels[t.rstrip(':')] = len(ins); continue
        if t.startswith('.'): continue
        ins.append((n, t))
    c = Counter(kind(t.split()[0]) for _, t in ins)
    print(f"{name[:70]}: {len(ins)} instructions {dict(c)}")
    loops = []
    for idx, (_, t) in enumerate(ins):
        m = re.match(r'^s_cbranch\w*\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)', t)
        if m:
            lab = m.group(1) or m.group(2)
            if lab in labels and labels[lab] <= idx: loops.append((idx - labels[lab], lab, labels[lab], idx))
    for ln, lab, s, e in sorted(loops, reverse=True)[:4]:
        cc = Counter(kind(t.split()[0]) for _, t in ins[s:e + 1])
        print(f"    loop {lab}: {ln} instructions {dict(cc)}")
