#!/bin/bash
# f1 / f2 rows: stage times and kernel tables of the final kernels (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02; mkdir -p $O
rm -f $O/ray_bench.jsonl $O/stage_bench.jsonl
for W in c2 c3 c4; do timeout 300 python scripts/ray_bench.py $W --steps 20 --cpu-iters 2 >> $O/ray_bench.jsonl 2>> $O/ray_bench.err; done
for W in c2 c4; do timeout 300 python scripts/stage_bench.py $W --iters 20 --cpu-iters 1 2>> $O/stage_bench.err | grep '^{' >> $O/stage_bench.jsonl; done
timeout 400 bash scripts/prof_ray.sh c2 c4 > $O/prof_ray.txt
cp $R/gpurun_out/prof_ray_c2/c2_kernel_stats.csv $O/rocprof_ray_c2_kernel_stats.csv
cp $R/gpurun_out/prof_ray_c4/c4_kernel_stats.csv $O/rocprof_ray_c4_kernel_stats.csv
cat $O/ray_bench.jsonl; grep feature $O/stage_bench.jsonl
