#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_e.txt
cat $O/pytest_e.txt
timeout 500 python scripts/r06_probe.py "" "upd_blocks=512" "upd_blocks=1024" "upd_blocks=1407" "tiled_lds_pad=0" "overlap=0" 2>/dev/null | tail -1 > $O/probe_e.json
cat $O/probe_e.json
timeout 300 python3 scripts/phases_tiled.py c4 > $O/phases_c4_e.json 2>$O/phases_c4.err || tail -3 $O/phases_c4.err
cat $O/phases_c4_e.json
