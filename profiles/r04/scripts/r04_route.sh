#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_route
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_tiling_gpu.py tests/test_halo_rccl_gpu.py tests/test_bench_gpu.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
cat $O/pytest.txt
timeout 600 python bench.py --workload c5 --steps 100 --warmup 10 > $O/bench_c5_routed_1rank.json 2> $O/bench_c5.err || tail -5 $O/bench_c5.err
tail -c 1800 $O/bench_c5_routed_1rank.json
