#!/bin/bash
# evidence for raycasting inside the batches: stage bench per workload, kernel stats of the batch call, soak
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ray5
mkdir -p $O
cd $R
for W in c2 c3 c4; do timeout 600 python scripts/ray_bench.py $W --cpu-iters 2 2>/dev/null | tee -a $O/ray_bench.jsonl; done
bash scripts/r04_rayprof.sh > $O/rayprof.txt 2>&1; cp gpurun_out/r04_rayprof/rb_kernel_stats.csv $O/rocprof_ray_batch_c2_kernel_stats.csv; cp gpurun_out/r04_rayprof/run.json $O/ray_batch_c2.json
for P in ray rayp2; do timeout 400 python scripts/soak_r04.py ${SOAK_S:-90} 5 no $P 2>&1 | tail -2 | tee -a $O/soak.jsonl; done
