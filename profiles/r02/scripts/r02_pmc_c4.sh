#!/bin/bash
# traffic counters of the configs[3] fused launch only (quick A/B)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum")
tag=$1; shift
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$tag/p$i -o p -- python3 $R/bench.py --workload c4 --no-host-legs --no-large --no-cpu-baseline --steps 60 --warmup 10 --profile-steps 6 "$@" > $O/pmc_$tag.p$i.log 2>&1 || tail -2 $O/pmc_$tag.p$i.log
done
python3 $R/scripts/pmc_traffic.py $tag $O/pmc_$tag $O/pmc_traffic_ab.json
