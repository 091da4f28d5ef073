#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE (KiB) of scripts/ubench/pmc_calib.bin against its known byte counts.
usage: pmc_calib.py <dir with p*/..counter_collection.csv> <known.json> <out.json>"""
import collections, csv, glob, json, sys
root, known_f, out = sys.argv[1:4]
known = json.loads(open(known_f).read().strip().splitlines()[-1])
agg = collections.defaultdict(dict)
for f in glob.glob(root + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        for k in known:
            if k + "(" in n or n.startswith(k):
                agg[k][row["Counter_Name"]] = float(row["Counter_Value"])
res = {}
for k, (used, lines) in known.items():
    c = agg.get(k, {})
    e = {"used_bytes": used, "line_bytes": lines}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        if cn in c:
            e[cn + "_bytes"] = c[cn] * 1024.0
            e[cn + "_per_used_byte"] = c[cn] * 1024.0 / used
            e[cn + "_per_line_byte"] = c[cn] * 1024.0 / lines
    res[k] = e
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, e in res.items():
    print(k, {a: round(b, 3) for a, b in e.items() if a.endswith("_byte")})
