#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab
mkdir -p $O
cd $R
timeout 900 python scripts/c4_ab.py "tbatch=0" "tbatch_max=4,batch_fuse=0,tb_groups=2000" "tbatch_max=8,batch_fuse=0,tb_groups=2000" "tbatch_max=8,batch_fuse=0,tb_groups=1024" "tbatch_max=8,batch_fuse=0" > $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json
