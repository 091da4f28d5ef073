#!/bin/bash
# batch pipeline iteration: parity tests, batch-size probe, kernel trace of the default bench
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03_batch_iter
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -15
python scripts/batch_probe.py "$@" | tee $O/probe.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o c2 -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large --steps 4000 --warmup 400 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_nofuse -o c2 -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large --steps 4000 --warmup 400 --set batch_fuse=0 > $O/bench_nofuse.json 2> $O/bench_nofuse.err
python3 - $O <<'PY'
import csv, sys
for d in ("rocprof", "rocprof_nofuse"):
    print(d)
    for r in csv.DictReader(open(f"{sys.argv[1]}/{d}/c2_kernel_stats.csv")):
        n = r["Name"]
        if "k_m" in n or "k_update_bin" in n:
            print("  ", n.split("(")[0][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
