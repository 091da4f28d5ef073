#!/bin/bash
# where the raycasting stage's time goes at configs[3] (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02_ray; mkdir -p $O
for D in 0 1 2 3; do python scripts/ray_bench.py c4 --steps 10 --cpu-iters 1 --dbg-ray $D 2>>$O/err.log | tee -a $O/ray_dbg.jsonl; done
bash scripts/prof_ray.sh c4 | tee $O/prof_ray.txt
bash scripts/pmc_cmd.sh ray k_ray_compact,k_ray_resolve,k_ray scripts/ray_bench.py c4 --steps 4 --cpu-iters 1 | tee $O/pmc.txt
