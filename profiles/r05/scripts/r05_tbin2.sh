#!/bin/bash
# round 5: the second edition of the large-scan bin half — parity (both fixture variants), then A/B at configs[3]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_tbin2
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests/test_parity_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
timeout 600 python scripts/c4_ab.py "tbin_ver=1" "tbin_ver=2" > $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json; tail -3 $O/c4_ab.err
