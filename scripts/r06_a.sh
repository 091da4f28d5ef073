#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 500 python scripts/r06_probe.py "" "dbg_upd=4" "dbg_upd=8" "upd_blocks=384" "upd_blocks=512" "upd_blocks=1024" "upd_blocks=384,dbg_upd=8" "overlap=0" 2>/dev/null | tail -1 > $O/probe_a.json
cat $O/probe_a.json
timeout 400 python scripts/r06_probe.py --naz=8192,10240,12288 "" "overlap=0" 2>/dev/null | tail -1 > $O/probe_naz.json
cat $O/probe_naz.json
timeout 300 python3 scripts/timeline.py c4 --set dbg_upd=8 > $O/timeline_c4_skipheavy.json 2>/dev/null
