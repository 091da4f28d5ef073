#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_c5probe
mkdir -p $O
cd $R
export FDM_BENCH_TRACE=1
for i in 1 2 3 4; do
timeout 300 python3 bench.py --workload c5 --steps 100 --warmup 10 > $O/t_$i.json 2> $O/t_$i.err
python3 - $i $O <<'PY'
import json,sys
i,O=sys.argv[1],sys.argv[2]
d=json.loads([l for l in open(f'{O}/t_{i}.json') if l.startswith('{')][-1]); print('run', i, 'ms_per_step', round(d['ms_per_step'],4))
PY
grep trace $O/t_$i.err
done
