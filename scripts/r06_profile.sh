#!/bin/bash
# Round-6 measurement pass on the GPU box (SURVEY.md §8d for EVERY config): rocprofv3 kernel-trace stats of the bench
# commands, calibrated PMC traffic passes (separate --pmc runs, no tracing; every pass under its own `timeout`: a
# counter set the hardware refuses makes rocprofv3 abort and then hang), SQ counters, block timelines.
# Outputs under gpurun_out/r06ev/; the summaries are copied into profiles/r05/ by scripts/collect_r05.sh.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06ev
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
args_of() {
  case $1 in
    c2) echo "" ;;
    c3) echo "--workload c3 --steps 2000 --warmup 200" ;;
    c4) echo "--workload c4 --steps 1000 --warmup 100" ;;
    c5) echo "--workload c5 --routed 0 --steps 500 --warmup 50" ;;
  esac
}
# (1) kernel-trace stats of the bench commands (no host legs / CPU baseline / large leg: the timed region's kernels only)
for W in c2 c3 c4 c5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_$W -o $W -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large $(args_of $W) > $O/rocprof_$W.json 2> $O/rocprof_$W.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_c5_routed -o c5r -- python3 $R/bench.py --workload c5 --steps 100 --warmup 10 > $O/rocprof_c5_routed.json 2> $O/rocprof_c5_routed.err
# (2) traffic counters (request-size counters: exact for every access pattern of this engine, profiles/r02/pmc_calibration.json)
SETS=("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum")
for W in c2 c3 c4 c5; do
  i=0
  for SET in "${SETS[@]}"; do
    i=$((i+1))
    A=$(args_of $W | sed 's/--steps [0-9]* --warmup [0-9]*//')
    ST="--steps 64 --warmup 16 --profile-steps 6"; [ $W = c2 ] && ST="--steps 640 --warmup 64 --profile-steps 6"
    timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$W/p$i -o p -- python3 $R/bench.py --no-host-legs --no-large --no-cpu-baseline $A $ST > $O/pmc_$W.p$i.log 2>&1 || tail -2 $O/pmc_$W.p$i.log
  done
  python3 $R/scripts/pmc_traffic.py $W $O/pmc_$W $O/pmc_traffic.json
done
# (3) SQ counters of the two dominant launches
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/sq_c2/p$i -o p -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large --steps 640 --warmup 64 --profile-steps 4 > $O/sq_c2.p$i.log 2>&1 || tail -2 $O/sq_c2.p$i.log
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/sq_c4/p$i -o p -- python3 $R/bench.py --workload c4 --no-host-legs --no-cpu-baseline --no-large --steps 64 --warmup 16 --profile-steps 4 > $O/sq_c4.p$i.log 2>&1 || tail -2 $O/sq_c4.p$i.log
done
cd $R
python3 - $O <<'PY' > $O/pmc_sq.txt
import collections, csv, glob, sys
for tag, keys in (("sq_c2", ("k_mbatch",)), ("sq_c4", ("k_tupdate_tbin", "k_tbin", "k_tupdate"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/{tag}/p*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            for key in keys:
                if "fdm::" + key + "<" in row["Kernel_Name"] or "fdm::" + key + "(" in row["Kernel_Name"]:
                    agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    break
    for k, d in agg.items():
        print(tag, k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
cat $O/pmc_sq.txt | cut -c1-400
# (4) block timelines
timeout 200 python3 scripts/timeline_batch.py > $O/timeline_c2_batch.json 2> $O/timeline_c2_batch.err
timeout 300 python3 scripts/timeline.py c4 > $O/timeline_c4_fused.json 2> $O/timeline_c4_fused.err
# (5) the default bench line + the other configs' lines, unprofiled
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
for W in c3 c4; do timeout 300 python3 bench.py $(args_of $W) --no-large --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err; done
timeout 300 python3 bench.py --workload c5 --routed 0 --steps 500 --warmup 50 --no-large --no-cpu-baseline > $O/bench_c5_plain.json 2> $O/bench_c5_plain.err
timeout 300 python3 bench.py --workload c5 --steps 100 --warmup 10 > $O/bench_c5_routed_1rank.json 2> $O/bench_c5_routed_1rank.err
ls $O | head -60
cat $O/pmc_traffic.json | python3 -c "import json,sys; d=json.load(sys.stdin); print({w: {k: round(v.get('hbm_bytes_per_launch', 0)) for k, v in e.items()} for w, e in d.items()})"
# (6) raycasting: stage bench per workload (single-scan stage, batch call), kernel stats of the batch call with raycasting
rm -f $O/ray_bench.jsonl
for W in c2 c3 c4; do timeout 600 python3 scripts/ray_bench.py $W --cpu-iters 2 2>/dev/null >> $O/ray_bench.jsonl; done
cat $O/ray_bench.jsonl | cut -c1-330
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_ray_batch -o rb -- python3 $R/scripts/ray_batch_run.py 640 > $O/ray_batch_c2.json 2> $O/ray_batch_c2.err)
# (7) soak of the end-of-round code: engine against engine, bit for bit, every profile
rm -f $O/soak.jsonl
for P in small p2 tiled ray rayp2 walk rayw; do timeout 400 python3 scripts/soak_r04.py ${SOAK_S:-60} 5 no $P 2>/dev/null | tail -1 >> $O/soak.jsonl; done
cat $O/soak.jsonl
# (8) round 6: the stage bench (stencils), the dependent-round-trip ubench, the perf-guard figures
timeout 600 python3 scripts/stage_bench.py > $O/stage_bench.jsonl 2>/dev/null; tail -3 $O/stage_bench.jsonl | cut -c1-300
timeout 120 scripts/ubench/dep_latency.bin > $O/dep_latency.jsonl 2>/dev/null
timeout 600 python3 tests/test_perf_guard_gpu.py 2>/dev/null | tail -1 > $O/perf_guard.json; cat $O/perf_guard.json
