#!/bin/bash
# HBM-side traffic per launch of a workload's kernels: the calibrated request-size counters, one --pmc pass per set
# (scripts/pmc_traffic.py turns them into bytes).  usage: pmc_traffic_run.sh <tag> <out.json> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; OUT=$2; shift 2
O=$R/gpurun_out/pmct_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum")
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --no-host-legs --no-large --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 4 "$@" > $O/p$i.log 2>&1 || tail -2 $O/p$i.log
done
python3 $R/scripts/pmc_traffic.py $TAG $O $OUT
