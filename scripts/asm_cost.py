#!/usr/bin/env python3
"""Static issue-cost estimate of one kernel in a -gline-tables-only .s file, by source line.
Cycle prices per wave64 instruction from scripts/ubench/valu_issue*.hip (gfx950): plain VOP2 add / sub / mul f32,
add / sub u32, and / or / xor, mov, lshr / ashr = 2; every other vector instruction 4 (rcp / sqrt / rsq 8);
scalar 4 (s_nop / s_waitcnt / s_barrier / branches not counted); LDS and memory instructions are listed apart.
usage: asm_cost.py file.s <mangled-function-prefix> [top] [--files a.hpp,b.hpp]"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split('\n')
prefix = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 40
CHEAP = {'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_mul_f32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32',
         'v_or_b32', 'v_xor_b32', 'v_mov_b32', 'v_lshrrev_b32', 'v_ashrrev_i32', 'v_add_co_u32', 'v_addc_co_u32',
         'v_not_b32'}
SLOW = {'v_rcp_f32', 'v_sqrt_f32', 'v_rsq_f32', 'v_rcp_f64', 'v_sqrt_f64', 'v_rsq_f64', 'v_exp_f32', 'v_log_f32'}
files, cur, on = {}, ('?', 0), False
valu, salu, cyc, lds, mem = (collections.Counter() for _ in range(5))
ops = collections.Counter()
for line in src:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        continue
    if re.match(r'^' + re.escape(prefix) + r'.*:', line):
        on = True
    elif line.startswith('.Lfunc_end'):
        on = False
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
        continue
    if not on:
        continue
    m = re.match(r'^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)', line)
    if not m:
        continue
    op = m.group(1)
    base = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if op.startswith('v_'):
        c = 2 if base in CHEAP else (8 if base in SLOW else 4)
        valu[cur] += 1; cyc[cur] += c; ops[base] += 1
    elif op.startswith('s_'):
        if re.match(r's_(waitcnt|nop|barrier|endpgm|sleep)', op):
            continue
        salu[cur] += 1
    elif op.startswith('ds_'):
        lds[cur] += 1
    else:
        mem[cur] += 1
print('static: VALU %d (%d cycles) SALU %d LDS %d VMEM %d' % (sum(valu.values()), sum(cyc.values()), sum(salu.values()),
                                                             sum(lds.values()), sum(mem.values())))
byfile = collections.Counter()
for k, v in cyc.items():
    byfile[k[0]] += v
print('VALU cycles by file:', dict(byfile))
for k, v in sorted(cyc.items(), key=lambda kv: -kv[1])[:top]:
    print('   %-18s %5d  VALU %4d (%4d cyc)  SALU %4d  LDS %3d  VMEM %3d' % (k[0], k[1], valu[k], v, salu.get(k, 0), lds.get(k, 0), mem.get(k, 0)))
print('top ops:', ', '.join('%s %d' % kv for kv in ops.most_common(25)))
