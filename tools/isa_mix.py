#!/usr/bin/env python3
"""Instruction mix of kernels in an AMDGPU .s file: tools/isa_mix.py file.s [name-substring]"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    ops = collections.Counter()
    for l in body.split('\n'):
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'):
            continue
        ops[l.split()[0]] += 1
    tot = sum(ops.values())
    cls = collections.Counter()
    for k, v in ops.items():
        if k.startswith(('v_mad_u64', 'v_mul_lo', 'v_mul_hi')): cls['vmul'] += v
        elif k.startswith('v_lshl_add_u64'): cls['v64add'] += v
        elif k.startswith('v_'): cls['valu'] += v
        elif k.startswith('s_'): cls['salu'] += v
        elif k.startswith('ds_'): cls['lds'] += v
        elif k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cls['vmem'] += v
        else: cls['other'] += v
    print(name, 'total', tot, dict(cls))
    print('   ', dict(ops.most_common(18)))
