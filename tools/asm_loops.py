"""Lists the loops of one kernel in a hipcc -S listing with their instruction mix (a quick look at where a latency-bound
kernel spends its issue slots).   python tools/asm_loops.py file.s <first_line> <last_line>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
lo, hi = int(sys.argv[2]), int(sys.argv[3])
labels = {}
for i in range(lo, hi):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m:
        labels[m.group(1)] = i
loops = []
for i in range(lo, hi):
    m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s+s_branch\s+(\.LBB\d+_\d+)", lines[i])
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))
for a, b in sorted(loops):
    mix = Counter()
    for l in lines[a:b + 1]:
        l = l.strip()
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        op = l.split()[0]
        key = ("valu_f64" if re.match(r"v_(fma|mul|add|fmac|max|min|rcp|div|cmp\w*)_f64|v_(fma|mul|add)_f64", op) else
               "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "scratch" if op.startswith("scratch_") else
               "global" if op.startswith(("global_", "buffer_", "flat_")) else "waitcnt" if op.startswith("s_waitcnt") else
               "salu" if op.startswith("s_") else "other")
        mix[key] += 1
    print(f"loop {a + 1}-{b + 1}: {sum(mix.values())} instr", dict(mix))
