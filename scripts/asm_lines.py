#!/usr/bin/env python3
"""Static VALU / SALU instruction counts per source line of a -gline-tables-only -save-temps .s file.
usage: asm_lines.py file.s <mangled-function-prefix> [top]"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split('\n')
prefix = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files, cur, on = {}, ('?', 0), False
cnt, cnt_s = collections.Counter(), collections.Counter()
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
    if on and re.match(r'^\s+v_', line):
        cnt[cur] += 1
    if on and re.match(r'^\s+s_', line) and not re.match(r'^\s+s_(waitcnt|nop|barrier)', line):
        cnt_s[cur] += 1
print('static VALU', sum(cnt.values()), 'SALU', sum(cnt_s.values()))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:top]:
    print('   %-18s %5d  VALU %4d  SALU %4d' % (k[0], k[1], v, cnt_s.get(k, 0)))
