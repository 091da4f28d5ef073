#!/bin/bash
# Copy the judged summaries of scripts/gpu_round.sh (gpurun_out/round/, scratch) into profiles/rNN/ (tracked).
# usage: scripts/collect_round.sh r01
set -e
O=gpurun_out/round; P=profiles/${1:-r01}
mkdir -p $P
cp $O/bench_c2.json $O/bench_c3.json $O/bench_c4.json $O/bench_c5.json $O/pytest_gpu.txt $O/ray_bench.jsonl $O/bench_host.jsonl $P/
cp $O/rocprof_c2/c2_kernel_stats.csv $P/rocprof_bench_c2_kernel_stats.csv
cp $O/rocprof_c4/c4_kernel_stats.csv $P/rocprof_bench_c4_kernel_stats.csv
cp $O/rocprof_ray_c2_kernel_stats.csv $O/rocprof_ray_c4_kernel_stats.csv $O/rocprof_stages_c2_kernel_stats.csv $O/rocprof_stages_c4_kernel_stats.csv $P/
cp $O/stage_bench.jsonl $P/stage_bench.jsonl
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python3 scripts/pmc_summary.py $O $P/pmc_summary.txt > /dev/null
python3 - "$P" <<'PY'
import csv, json, sys
P = sys.argv[1]
for w in ("c2", "c4"):
    rows = list(csv.DictReader(open(f"{P}/rocprof_bench_{w}_kernel_stats.csv")))
    d = json.loads(open(f"{P}/bench_{w}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(w, "bench %.0f Mpts/s %.2f us/scan | events %.2f us, rocprof %s %.2f us | frac %.4f traffic %s" % (
        d["value"], d["ms_per_step"] * 1e3, r["avg_kernel_us"], rows[0]["Name"].split("<")[0][-20:], float(rows[0]["AverageNs"]) / 1e3,
        r["frac"], r["traffic"]))
for w in ("c3", "c5"):
    d = json.loads(open(f"{P}/bench_{w}.json").read().strip().splitlines()[-1])
    print(w, "bench %.0f Mpts/s %.2f us/scan" % (d["value"], d["ms_per_step"] * 1e3))
t = json.load(open("profiles/pmc_traffic.json"))
for w in t:
    print(w, "PMC MB/launch", {k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in t[w].items()})
PY
