#!/bin/bash
# a longer soak of the end-of-round code: every profile 300 s, engine against engine, bit for bit
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_soak_long
mkdir -p $O
cd $R
rm -f $O/soak.jsonl
for P in small p2 tiled tbatch ray rayp2 walk; do timeout 700 python3 scripts/soak_r04.py 300 5 no $P 2>/dev/null | tail -1 >> $O/soak.jsonl; done
cat $O/soak.jsonl
