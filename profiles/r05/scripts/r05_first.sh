#!/bin/bash
# round 5, first contact: price list of the instruction classes the large-scan kernels are made of (ubench), the
# baseline of this box at configs[3] and the default line
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_first
mkdir -p $O
cd $R
timeout 300 scripts/ubench/valu_issue.bin > $O/valu_issue.jsonl 2> $O/valu_issue.err
cat $O/valu_issue.jsonl
timeout 600 python scripts/c4_ab.py "" > $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json; tail -3 $O/c4_ab.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
tail -c 1500 $O/bench_20_5.json
