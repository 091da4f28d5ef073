#!/usr/bin/env python3
"""Condense the raw rocprofv3 --pmc CSVs of scripts/gpu_round.sh into profiles/rNN/pmc_summary.txt
(mean per launch of every counter for k_bin* and k_update).  python scripts/pmc_summary.py <round_dir> <out>"""
import collections, csv, glob, json, os, sys
rd, out = sys.argv[1], sys.argv[2]
traffic = json.load(open(os.path.join(rd, "pmc_traffic.json"))) if os.path.exists(os.path.join(rd, "pmc_traffic.json")) else {}
lines = ["# rocprofv3 --pmc summary (mean per launch; separate --pmc passes per counter set;",
         "# command: python bench.py --no-large --no-cpu-baseline [--workload c4]; scripts/gpu_round.sh)",
         "# hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB  (gfx950 FETCH_SIZE reads half of a wide coalesced stream)", ""]
for w in ("c2", "c4"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(rd, f"pmc_{w}", "p*", "p_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = "k_update_bin" if "k_update_bin" in name else ("k_bin" if "k_bin" in name else ("k_update" if "k_update" in name else None))
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in ("k_update_bin", "k_bin", "k_update"):
        if not acc[k]:
            continue
        t = traffic.get(w, {}).get(k, {})
        lines.append(f"[{w}] {k}: launches={t.get('launches', '?')} hbm_bytes/launch={t.get('hbm_bytes_per_launch', 0):.0f} "
                     f"(read {t.get('read_bytes_per_launch', 0):.0f}, write {t.get('write_bytes_per_launch', 0):.0f})")
        for c in sorted(acc[k]):
            v = acc[k][c]
            lines.append(f"    {c:<28} {sum(v) / len(v):.6g}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
