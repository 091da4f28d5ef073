#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_b.txt
cat $O/pytest_b.txt
timeout 500 python scripts/r06_probe.py "" "upd_blocks=384" "upd_blocks=512" "upd_blocks=1024" "upd_blocks=1407" "tiled_lds_pad=0" "tiled_lds_pad=8192" "overlap=0" 2>/dev/null | tail -1 > $O/probe_b.json
cat $O/probe_b.json
timeout 300 python3 scripts/timeline.py c4 > $O/timeline_c4_slab.json 2>/dev/null
