#!/bin/bash
# tests/test_post_gpu.py, then scripts/post_ab.py plain and under rocprofv3 (kernel times of the stencil stages); run on the GPU box
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests/test_post_gpu.py -q -x 2>&1 | tail -3
python scripts/post_ab.py c4 2>&1 | grep "^{"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_post_ab -o ab -- python3 $R/scripts/post_ab.py c4 --iters 20 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_post_ab/ab_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("k_fusion", "k_features", "copy_strided", "k_copy")):
        print("  %-60s calls=%5s avg=%9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3))
PY
