#!/bin/bash
# gpurun_out/r06ev (scratch) -> profiles/r06 (tracked): the summaries the round's figures come from
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r06ev
P=$R/profiles/r06
mkdir -p $P
for W in c2 c3 c4 c5; do
  cp $O/rocprof_$W/${W}_kernel_stats.csv $P/rocprof_bench_${W}_kernel_stats.csv
done
cp $O/rocprof_c5_routed/c5r_kernel_stats.csv $P/rocprof_bench_c5_routed_1rank_kernel_stats.csv
cp $O/pmc_traffic.json $P/pmc_traffic.json
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json   # what bench.py's roofline.traffic reads
cp $O/pmc_sq.txt $P/pmc_sq_c2_c4.txt
cp $O/timeline_c2_batch.json $O/timeline_c4_fused.json $P/
for f in bench_default bench_c3 bench_c4 bench_c5_plain bench_c5_routed_1rank; do grep '^{' $O/$f.json | tail -1 > $P/$f.json; done
cp $O/ray_bench.jsonl $P/ray_bench.jsonl
cp $(find $O/rocprof_ray_batch -name "rb_kernel_stats.csv" | head -1) $P/rocprof_ray_batch_c2_kernel_stats.csv
cp $O/ray_batch_c2.json $P/ray_batch_c2.json
python3 - $O/soak.jsonl > $P/soak.json <<'PY'
import json, sys
print(json.dumps({"script": "scripts/soak_r04.py (engine against engine, every layer compared bit for bit every 5 calls)",
                  "profiles": [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]}, indent=1))
PY
ls -la $P
cp $O/stage_bench.jsonl $P/stage_bench.jsonl 2>/dev/null
cp $O/dep_latency.jsonl $P/dep_latency.jsonl 2>/dev/null
cp $O/perf_guard.json $P/perf_guard.json 2>/dev/null
