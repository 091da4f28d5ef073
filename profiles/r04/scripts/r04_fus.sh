#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
FDM_ENGINE_LIB=$R/fastdem_amd/lib/libfdm_engine_nofb.so timeout 900 python3 scripts/stage_bench.py c4 2>/dev/null | grep -E "fusion\(r=0.15" | cut -c1-200
timeout 900 python3 scripts/stage_bench.py c4 2>/dev/null | grep -E "fusion\(r=0.15" | cut -c1-200
