#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_trace20
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-host-legs --no-large --no-cpu-baseline > $O/run.json 2> $O/run.err
python3 - $O <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+'/prof/**/t_kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the timed region: the last run of >= 15 consecutive k_mbatch launches
idx=[i for i,r in enumerate(rows) if 'k_mbatch' in r['Kernel_Name']]
d=json.loads([l for l in open(sys.argv[1]+'/run.json') if l.startswith('{')][-1]); print('ms_per_step', d['ms_per_step'], 'total_us', d['ms_per_step']*20*1e3)
# print the last 30 kernels of the process's batch phase
last=idx[-1]
t0=None
for r in rows[max(0,last-26):last+4]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print(round((s-t0)/1e3,1), round((e-s)/1e3,1), r['Kernel_Name'][:60], r.get('Grid_Size_X','') , r.get('Grid_Size_Y',''))
PY
