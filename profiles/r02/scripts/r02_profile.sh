#!/bin/bash
# Round-2 measurement pass on the GPU box: rocprofv3 kernel-trace stats of the bench commands, PMC traffic
# passes (separate --pmc runs, no tracing) for the fused launch and for the per-phase table of configs[3].
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) kernel-trace stats of the default bench command (without the PCIe legs: they launch the same kernels on
#     pinned host memory and would pull the per-kernel average away from the timed region) and of configs[3]
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_c2 -o c2 -- python3 $R/bench.py --no-host-legs --no-cpu-baseline > $O/rocprof_c2.json 2> $O/rocprof_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_c4 -o c4 -- python3 $R/bench.py --workload c4 --no-host-legs --no-cpu-baseline --steps 2000 --warmup 200 > $O/rocprof_c4.json 2> $O/rocprof_c4.err
# (2) traffic counters
SETS=("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum")
pmc() {  # tag, bench args...
  tag=$1; shift
  i=0
  for SET in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$tag/p$i -o p -- python3 $R/bench.py --no-host-legs --no-large --no-cpu-baseline --steps 60 --warmup 10 --profile-steps 6 "$@" > $O/pmc_$tag.p$i.log 2>&1 || tail -2 $O/pmc_$tag.p$i.log
  done
  python3 $R/scripts/pmc_traffic.py $tag $O/pmc_$tag $O/pmc_traffic.json
}
pmc c2
pmc c4 --workload c4
# per-phase table of configs[3]: the two kernels on their own, then with the measurement-only early exits
pmc c4_split --workload c4 --overlap 0
pmc c4_bin_loads_arith --workload c4 --overlap 0 --set dbg_no_atomics=2
pmc c4_upd_rows --workload c4 --overlap 0 --set dbg_upd=1
pmc c4_upd_fold --workload c4 --overlap 0 --set dbg_upd=2
cd $R
ls $O | head -50
