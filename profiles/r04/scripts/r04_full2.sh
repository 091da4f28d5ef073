#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_full2
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
for i in 1 2 3; do
timeout 300 python3 bench.py --workload c5 --steps 100 --warmup 10 > $O/bench_c5_routed_$i.json 2> $O/bench_c5_$i.err
python3 - $i <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r04_full2/bench_c5_routed_{sys.argv[1]}.json') if l.startswith('{')][-1]); print('c5 routed 1 rank', d['ms_per_step'], d['value'])
PY
done
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
