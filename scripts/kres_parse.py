#!/usr/bin/env python3
"""Per-kernel resource table from a hipcc -Rpass-analysis=kernel-resource-usage log.
usage: kres_parse.py <log> [substr ...]"""
import re, subprocess, sys
log = open(sys.argv[1]).read()
pats = sys.argv[2:]
cur = None
rows = {}
for line in log.splitlines():
    m = re.search(r"remark: \S+:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" ")[0]] = v
for name, r in rows.items():
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    short = dem.split("(")[0].replace("void ", "")
    if pats and not any(p in short for p in pats):
        continue
    print(f"{short[:90]:90s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} sgpr {r.get('TotalSGPRs','?'):>4} "
          f"scratch {r.get('ScratchSize','?'):>5} occ {r.get('Occupancy','?'):>2} lds {r.get('LDS','?'):>6}")
