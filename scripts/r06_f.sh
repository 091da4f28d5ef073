#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 300 python3 scripts/phases_tiled.py c4 > $O/phases_c4_f.json 2>$O/phases_c4.err || tail -3 $O/phases_c4.err
python3 -c "
import json; d=json.load(open('$O/phases_c4_f.json')); print(d['span_us']); print(d['update_all']); print(d['update_light'])"
