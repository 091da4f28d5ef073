#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection CSVs into per-kernel HBM traffic per launch.
   traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE/WRITE_SIZE are in KiB and, on
   gfx950 with this rocprofv3, FETCH_SIZE reports exactly half of a wide coalesced read stream
   (MI355X_MICROARCH.md §HBM) — hence the factor 2 on the read side.
   usage: pmc_traffic.py <workload-tag> <dir-with-p*/...counter_collection.csv> <out.json>"""
import collections, csv, glob, json, sys
tag, root, out = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "fdm::k_update_bin" in name:
            k = "k_update_bin"  # one launch: update of scan t + bin of scan t+1
        elif "fdm::k_bin" in name:
            k = "k_bin"
        elif "fdm::k_update" in name:
            k = "k_update"
        else:
            continue
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
try:
    res = json.load(open(out))
except Exception:
    pass
entry = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {"launches": max(len(v) for v in d.values()), "counters_mean": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        e["read_bytes_per_launch"] = 2.0 * m["FETCH_SIZE"] * 1024.0
        e["write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024.0
    entry[k] = e
res[tag] = entry
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({tag: {k: {kk: vv for kk, vv in e.items() if kk != "counters_mean"} for k, e in entry.items()}}))
