#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 400 python scripts/r06_probe.py --lib=$R/fastdem_amd/lib/libfdm_engine_r05.so "" > $O/probe_m_r05.json 2>/dev/null; tail -1 $O/probe_m_r05.json
timeout 500 python scripts/r06_probe.py "" "tiled_lds_pad=0" "tiled_lds_pad=4096" 2>/dev/null | tail -1 > $O/probe_m.json
cat $O/probe_m.json
for W in c5; do timeout 300 python bench.py --workload c5 --routed 0 --steps 300 --warmup 50 --no-large --no-cpu-baseline > $O/b.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/b.json')); print('c5 plain', d['ms_per_step'], d['roofline']['frac'])"; done
timeout 300 python bench.py --workload c5 --routed 0 --steps 300 --warmup 50 --no-large --no-cpu-baseline --set tiled_lds_pad=4096 > $O/b.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/b.json')); print('c5 plain pad4096', d['ms_per_step'], d['roofline']['frac'])"
