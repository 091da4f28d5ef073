#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ray
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_batch_ray_gpu.py tests/test_raycast_gpu.py -m gpu -q -x 2>&1 | tail -5
bash scripts/r04_rayprof.sh 2>&1 | grep -v '^"' | head -9
timeout 300 python scripts/ray_bench.py c2 --cpu-iters 2 2>/dev/null
