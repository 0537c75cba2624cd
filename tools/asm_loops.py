#!/usr/bin/env python3
"""Developer tool: loop structure + instruction mix of one kernel in a hipcc -S dump.
usage: asm_loops.py kernels.s <substring of mangled kernel name>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i


def is_instr(l):
    l = l.strip()
    return bool(l) and not l.startswith(('.', ';', '_Z')) and not l.endswith(':')


print(f'{key}: {sum(map(is_instr, body))} instructions total')
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = labels.get(m.group(1) or m.group(2), 10 ** 9)
        if t < i:
            loops.append((t, i))
for t, i in loops:
    seg = [l.strip() for l in body[t:i + 1] if is_instr(l)]
    mix = collections.Counter()
    for ins in seg:
        op = ins.split()[0]
        if op.startswith('v_'):
            if any(x in op for x in ('rcp', 'rsq', 'sqrt', 'sin', 'cos', 'exp', 'log')):
                mix['v_transc'] += 1
            elif 'f64' in op:
                mix['v_f64'] += 1
            elif op.startswith('v_cmp'):
                mix['v_cmp'] += 1
            elif op.startswith('v_cndmask'):
                mix['v_cndmask'] += 1
            elif op.startswith('v_pk_'):
                mix['v_pk'] += 1
            else:
                mix['v_other'] += 1
        elif op.startswith('s_'):
            mix['s_waitcnt' if 'waitcnt' in op else ('s_nop' if 'nop' in op else 'salu')] += 1
        else:
            mix['mem'] += 1
    print(f'loop lines {t}-{i}: {len(seg)} instr  {dict(mix)}')
