#!/usr/bin/env python3
"""Instruction mix of every loop (backward branch) in a gfx950 .s file -- used to see what the
PDIPM iteration of the DPP-row kernels spends its issue slots on."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
def stats(a, b):
    c = {}; n = 0
    for l in lines[a:b]:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
        op = t.split()[0]; n += 1
        if op in ('v_fma_f64', 'v_fmac_f64_e32', 'v_mul_f64', 'v_add_f64', 'v_fmac_f64_dpp', 'v_fma_f64_dpp'): key = 'fp64'
        elif 'dpp' in t and op.startswith('v_mov'): key = 'dppmov'
        elif op.startswith('ds_'): key = 'ds'
        elif op.startswith('scratch'): key = 'scratch'
        elif 'accvgpr' in op: key = 'acc'
        elif op.startswith('v_mov'): key = 'vmov'
        elif op == 's_nop': key = 's_nop'
        elif op.startswith('s_waitcnt'): key = 's_waitcnt'
        elif op.startswith('s_'): key = 'salu'
        elif op.startswith('v_cndmask'): key = 'cndmask'
        elif op.startswith(('v_div', 'v_rcp', 'v_rsq', 'v_sqrt')): key = 'div'
        elif op.startswith('v_cmp'): key = 'cmp'
        else: key = 'other:' + op
        c[key] = c.get(key, 0) + 1
    return n, c
for i, l in enumerate(lines):
    m = re.search(r's_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            n, c = stats(labels[t], i)
            print('loop lines %d..%d: %d instructions' % (labels[t], i, n))
            print('   ', ', '.join('%s %d' % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])[:18]))
n, c = stats(0, len(lines))
print('whole file: %d instructions' % n)
