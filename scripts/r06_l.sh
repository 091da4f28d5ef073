#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 400 python scripts/r06_probe.py --lib=$R/fastdem_amd/lib/libfdm_engine_r05.so "" 2>/dev/null | tail -1
timeout 500 python scripts/r06_probe.py "" "upd_blocks=512" "upd_blocks=1024" "tiled_lds_pad=0" "overlap=0" 2>/dev/null | tail -1 > $O/probe_l.json
cat $O/probe_l.json
timeout 300 python3 scripts/timeline.py c4 > $O/timeline_c4_l.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/timeline_c4_l.json')); print(d['span_us'], 'upd_end', d['update_end_us_pct'], 'bin_end', d['bin_end_us_pct'])"
