#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python scripts/r06_probe.py "" "bin_delay=2" "bin_delay=5" "bin_delay=10" "bin_delay=5,bin_delay_blocks=512" "bin_delay=5,bin_delay_blocks=2048" "bin_delay=20" 2>/dev/null | tail -1 > $O/probe_delay.json
cat $O/probe_delay.json
timeout 300 python3 scripts/timeline.py c4 --set bin_delay=5 > $O/timeline_c4_delay5.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/timeline_c4_delay5.json')); print(d['span_us'], 'upd_end', d['update_end_us_pct'], 'bin_start', d['bin_start_us_pct'], 'bin_end', d['bin_end_us_pct'])"
