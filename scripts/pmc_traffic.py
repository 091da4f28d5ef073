#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection CSVs -> per-kernel HBM-side traffic per launch.

Calibrated on known-byte kernels (scripts/ubench/pmc_calib.hip, profiles/r02/pmc_calibration.json):
  * FETCH_SIZE tallies 64 B per memory-side read request, but a streaming read issues 128 B requests, so it
    reports 1/2 of a 16 / 8 / 4 B-per-lane stream and exactly the 64 B lines of scattered record / gather
    reads — a blanket 2x is wrong for scattered patterns.  The request-size counters are exact for every
    pattern:  read bytes = 128 * TCC_EA0_RDREQ_128B + 64 * TCC_EA0_RDREQ_64B + 32 * TCC_EA0_RDREQ_32B.
  * WRITE_SIZE (KiB) is exact for streaming stores, counts whole 64 B lines for partial-line record stores,
    and adds 32 B per memory-side atomic.
usage: pmc_traffic.py <tag> <dir-with-p*/...counter_collection.csv> <out.json>"""
import collections, csv, glob, json, sys
tag, root, out = sys.argv[1:4]
NAMES = [("fdm::k_mbatch", "k_mbatch"), ("fdm::k_tupdate_tbin", "k_tupdate_tbin"), ("fdm::k_tupdate", "k_tupdate"), ("fdm::k_tbin", "k_tbin"),
         ("fdm::k_update_bin", "k_update_bin"), ("fdm::k_bin", "k_bin"), ("fdm::k_update", "k_update")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = next((v for key, v in NAMES if key in row["Kernel_Name"]), None)
        if k:
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
    if "TCC_EA0_RDREQ_128B_sum" in m:
        r128, r64, r32 = m["TCC_EA0_RDREQ_128B_sum"], m.get("TCC_EA0_RDREQ_64B_sum", 0.0), m.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        if "TCC_EA0_RDREQ_sum" in m and r64 == 0.0:  # (no 64 B counter in the pass: the rest of the requests)
            r64 = max(0.0, m["TCC_EA0_RDREQ_sum"] - r128 - r32)
        e["read_bytes_per_launch"] = 128.0 * r128 + 64.0 * r64 + 32.0 * r32
    elif "FETCH_SIZE" in m:
        e["read_bytes_per_launch_lower_bound"] = m["FETCH_SIZE"] * 1024.0  # (x2 if it were all streaming)
    if "WRITE_SIZE" in m:
        e["write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024.0
    if "read_bytes_per_launch" in e and "write_bytes_per_launch" in e:
        e["hbm_bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
    entry[k] = e
res[tag] = entry
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({tag: {k: {kk: (round(vv) if isinstance(vv, float) else vv) for kk, vv in e.items() if kk != "counters_mean"}
                        for k, e in entry.items()}}))
