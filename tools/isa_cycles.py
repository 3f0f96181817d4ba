#!/usr/bin/env python3
"""Issue-cycle estimate per kernel from an AMDGPU .s: half-rate ops (32-bit multiplies, 64-bit add/compare) = 4 cycles,
other VALU = 2, s_nop N = N+1, everything else 1.  tools/isa_cycles.py file.s [substr]"""
import collections, re, sys
s = open(sys.argv[1]).read(); pat = sys.argv[2] if len(sys.argv) > 2 else ""
HALF = ('v_mad_u64', 'v_mul_lo', 'v_mul_hi', 'v_lshl_add_u64', 'v_cmp_le_u64', 'v_cmp_gt_u64', 'v_cmp_lt_u64', 'v_cmp_ge_u64', 'v_lshlrev_b64', 'v_lshrrev_b64')
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name: continue
    cyc = 0; n = 0; cls = collections.Counter()
    for l in body.split('\n'):
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'): continue
        op = l.split()[0]; n += 1
        if op.startswith(HALF): cyc += 4; cls['half'] += 1
        elif op.startswith('v_'): cyc += 2; cls['valu'] += 1
        elif op == 's_nop': cyc += int(l.split()[1]) + 1; cls['nop'] += 1
        else: cyc += 1; cls['other'] += 1
    print("%-60s instrs %5d est-cycles %6d %s" % (name[:60], n, cyc, dict(cls)))
