#!/bin/bash
# One full measurement round on the GPU box: tests, bench, rocprofv3 kernel trace of the bench
# command, PMC passes (separate runs, --pmc only).  Outputs under gpurun_out/round/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/round
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --workload c4 --no-large --no-cpu-baseline --steps 300 --warmup 30 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload c3 --no-large --no-cpu-baseline --steps 1000 --warmup 100 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload c5 --steps 100 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err
# PCIe-inclusive host entry points (opt-in legs; separate runs so the default command's kernel averages stay clean)
for W in c2 c3 c4; do
  python bench.py --workload $W --no-large --no-cpu-baseline --steps 200 --warmup 20 2> $O/bench_host_$W.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'workload': '$W', 'points_per_scan': d['config']['points_per_scan'], **{k: round(v, 4) for k, v in d.items() if k.startswith('host_')}}))" >> $O/bench_host.jsonl
done
cd /tmp && export TMPDIR=/tmp
for W in c2 c4; do
  EXTRA=""; [ $W = c4 ] && EXTRA="--workload c4 --steps 200 --warmup 20"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_$W -o $W -- python3 $R/bench.py --no-large --no-cpu-baseline $EXTRA > $O/rocprof_$W.log 2>&1
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$W/p$i -o p -- python3 $R/bench.py --no-large --no-cpu-baseline --steps 100 --warmup 10 --profile-steps 10 $EXTRA > $O/pmc_$W.p$i.log 2>&1
  done
  python3 $R/scripts/pmc_traffic.py $W $O/pmc_$W $O/pmc_traffic.json
done
cd $R
# the widened rows: stage timings + kernel traces
for W in c2 c3 c4; do python scripts/ray_bench.py $W --steps 20 --cpu-iters 2 >> $O/ray_bench.jsonl 2>> $O/ray_bench.err; done
for W in c2 c4; do python scripts/stage_bench.py $W --iters 20 --cpu-iters 1 2>> $O/stage_bench.err | grep '^{' >> $O/stage_bench.jsonl; done  # unprofiled wall times
bash scripts/prof_ray.sh c2 c4 > $O/prof_ray.txt 2>&1
bash scripts/prof_stages.sh c2 c4 > $O/prof_stages.txt 2>&1
for W in c2 c4; do
  cp gpurun_out/prof_ray_$W/${W}_kernel_stats.csv $O/rocprof_ray_${W}_kernel_stats.csv
  cp gpurun_out/prof_stage_$W/${W}_kernel_stats.csv $O/rocprof_stages_${W}_kernel_stats.csv
done
tail -c 600 $O/bench_c2.json; echo; cat $O/pytest_gpu.txt
