#!/bin/bash
# round 5: 16x16 wave tiles + persistent update wavefronts + second-edition bin — parity first, then configs[3]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_t16
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests/test_parity_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
cat $O/pytest.txt
timeout 900 python scripts/c4_ab.py "" "upd_blocks=256" "upd_blocks=384" "upd_blocks=768" "upd_blocks=1024" "upd_blocks=1444" > $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json; tail -3 $O/c4_ab.err
