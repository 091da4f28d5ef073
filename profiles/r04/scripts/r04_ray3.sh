#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1200 python -m pytest tests/test_batch_ray_gpu.py tests/test_batch_gpu.py -m gpu -q -x 2>&1 | tail -3
bash scripts/r04_rayprof.sh 2>&1 | grep -v '^"' | head -8
for P in ray rayp2; do timeout 400 python3 scripts/soak_r04.py 60 5 no $P 2>/dev/null | tail -1; done
