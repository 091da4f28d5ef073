#!/bin/bash
# round 3, first contact of the batch pipeline with the GPU: its parity tests, then the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03_first
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest_batch.txt
cat $O/pytest_batch.txt
timeout 600 python bench.py --no-large --no-cpu-baseline --no-host-legs > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 1500 $O/bench_c2.json; tail -5 $O/bench_c2.err
