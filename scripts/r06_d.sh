#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 400 python scripts/r06_probe.py --lib=$R/fastdem_amd/lib/libfdm_engine_r05.so "" "overlap=0" 2>/dev/null | tail -1 > $O/probe_d_r05.json
cat $O/probe_d_r05.json
timeout 600 python scripts/r06_probe.py "" "tile_cap=1200" "tile_cap=2048" "tile_cap=1200,overlap=0" "tile_cap=1200,upd_blocks=1024" "tile_cap=1200,tiled_lds_pad=0" 2>/dev/null | tail -1 > $O/probe_d.json
cat $O/probe_d.json
