#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_rayprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o rb -- python3 $R/scripts/ray_batch_run.py 640 ${RB_OPTS:-} > $O/run.json 2> $O/run.err
cat $O/run.json
f=$(find $O/prof -name "rb_kernel_stats.csv" | head -1); cp $f $O/rb_kernel_stats.csv; cut -c1-100 $O/rb_kernel_stats.csv | head -14; python3 - $O/rb_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Calls'], r['AverageNs'], r['Percentage'], r['Name'][:70])
PY
