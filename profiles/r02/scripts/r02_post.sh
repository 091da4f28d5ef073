#!/bin/bash
# stencil stages after a change: parity tests, stage times at the configs[3] map
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02_post; mkdir -p $O
timeout 600 python -m pytest tests/test_post_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python scripts/stage_bench.py c4 --iters 20 --cpu-iters 1 2>>$O/err.log | grep '^{' | tee -a $O/stage_bench.jsonl
