#!/bin/bash
# gpurun_out/r03 (scratch) -> profiles/r03 (tracked): the summaries the round's figures come from
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03
P=$R/profiles/r03
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
ls -la $P
