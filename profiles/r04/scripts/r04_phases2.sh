#!/bin/bash
# phase stamps of k_mbatch blocks for profiles/r04: default / with raycasting / configs[2] with and without the walker block
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_phases2
mkdir -p $O
cd $R
python3 scripts/phases_batch.py 2>/dev/null | tail -1 > $O/phases_c2_batch.json
python3 scripts/phases_batch.py raycast=1 2>/dev/null | tail -1 > $O/phases_c2_batch_raycast.json
python3 scripts/phases_batch.py workload=c3 2>/dev/null | tail -1 > $O/phases_c3_batch_walker.json
python3 scripts/phases_batch.py workload=c3 batch_walk=0 2>/dev/null | tail -1 > $O/phases_c3_batch_no_walker.json
python3 scripts/phases_batch.py batch_walk=1 2>/dev/null | tail -1 > $O/phases_c2_batch_walker_on.json
for f in $O/*.json; do echo $f; cut -c1-420 $f; done
