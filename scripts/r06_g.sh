#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_g.txt
cat $O/pytest_g.txt
timeout 400 python scripts/r06_probe.py --lib=$R/fastdem_amd/lib/libfdm_engine_r05.so "" 2>/dev/null | tail -1 > $O/probe_g_r05.json
cat $O/probe_g_r05.json
timeout 500 python scripts/r06_probe.py "" "upd_blocks=512" "upd_blocks=1024" "tiled_lds_pad=0" "overlap=0" 2>/dev/null | tail -1 > $O/probe_g.json
cat $O/probe_g.json
timeout 300 python3 scripts/phases_tiled.py c4 > $O/phases_c4_g.json 2>$O/phases_c4.err || tail -3 $O/phases_c4.err
python3 -c "
import json; d=json.load(open('$O/phases_c4_g.json')); print(d['span_us']); print(d['update_all']); print(d['update_heavy']); print(d['bin_first_round']); print(d['bin_late'])"
