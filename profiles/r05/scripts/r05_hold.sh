#!/bin/bash
# round 5: the raycasting stage held back with the update (option ray_hold): parity, soak, times per scan
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1500 python -m pytest tests/test_raycast_gpu.py tests/test_batch_ray_gpu.py tests/test_parity_gpu.py tests/test_pipeline_gpu.py -q -x 2>&1 | tail -3
for P in ray rayw; do timeout 300 python3 scripts/soak_r04.py 60 3 no $P 2>&1 | tail -1; done
for H in 1 0; do for W in c3 c4; do timeout 600 python3 scripts/ray_bench.py $W --cpu-iters 1 --set ray_hold=$H 2>/dev/null | tail -1 | cut -c1-260; done; done
