R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ovl -o t -- python3 $R/bench.py --workload c4 --no-large --no-cpu-baseline --steps 50 --warmup 10 --overlap 1 > $R/gpurun_out/ovl.log 2>&1
cd $R
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/ovl/t_kernel_trace.csv")) if "k_bin" in r["Kernel_Name"] or "k_update" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[100]["Start_Timestamp"])
for r in rows[100:112]:
    print(r["Kernel_Name"][:24], "queue", r["Queue_Id"], "start", (int(r["Start_Timestamp"])-t0)/1e3, "end", (int(r["End_Timestamp"])-t0)/1e3)
PY
