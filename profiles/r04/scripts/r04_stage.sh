#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_stage
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_post_gpu.py -m gpu -q -x 2>&1 | tail -3
for W in c2 c4; do timeout 900 python scripts/stage_bench.py $W --cpu-iters 1 2>$O/err_$W.txt | grep '^{' >> $O/stage_bench.jsonl || tail -3 $O/err_$W.txt; done
cat $O/stage_bench.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], d['stage'][:60], d.get('gpu_ms'), d.get('GBps'), d.get('ns_per_cell'))"
