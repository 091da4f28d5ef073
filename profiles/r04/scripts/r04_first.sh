#!/bin/bash
# round 4, first contact of the tile-batch pipeline (fdm_tbatch.hpp) with the GPU: its parity tests, the small-scan
# batch tests in both fixture variants (tiled_all: through the tile batches), then an A/B at configs[3]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_first
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_tbatch_gpu.py -m gpu -x -q 2>&1 | tail -40 > $O/pytest_tbatch.txt
cat $O/pytest_tbatch.txt
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest_batch.txt
cat $O/pytest_batch.txt
timeout 900 python scripts/c4_ab.py "tbatch=0" "" "tbatch_max=2" "tbatch_max=8" "tb_groups=256" "tb_groups=768" "tbatch_max=8,tb_groups=768" > $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json; tail -5 $O/c4_ab.err
