#!/bin/bash
# raycasting stage after a change: parity tests, stage time per config, kernel table at configs[3]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02_ray; mkdir -p $O
timeout 600 python -m pytest tests/test_raycast_gpu.py -m gpu -x -q 2>&1 | tail -3
for W in c4 c3 c2; do timeout 200 python scripts/ray_bench.py $W --steps 10 --cpu-iters 1 2>>$O/err.log | tee -a $O/ray_after.jsonl; done
timeout 300 bash scripts/prof_ray.sh c4 | tee $O/prof_ray_after.txt
