#!/bin/bash
# round 5: SQ instruction counters of the large-scan kernels (separate launches, --overlap 0), one run per variant
# usage: r05_sq.sh <tag> "<--set a=b ...>" [<tag2> "<...>"] ...
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
while [ $# -ge 2 ]; do
  TAG=$1; ARGS=$2; shift 2
  bash scripts/pmc_sq.sh $TAG --workload c4 $ARGS > gpurun_out/pmc_sq_$TAG.txt 2>&1
  cat gpurun_out/pmc_sq_$TAG.txt | tail -4
done
