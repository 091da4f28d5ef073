#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command (timed region only: no host legs, no CPU baseline, no large leg)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03_prof_c2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o c2 -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large --steps 4000 --warmup 400 > $O/bench.json 2> $O/bench.err
head -12 $O/rocprof/*kernel_stats.csv | cut -c1-200
tail -c 600 $O/bench.json
