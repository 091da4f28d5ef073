#!/bin/bash
# round 5: block timeline of the fused large-scan launch for several update-block counts
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_tl
mkdir -p $O
cd $R
for U in "$@"; do
  timeout 300 python scripts/timeline.py c4 --set upd_blocks=$U > $O/tl_$U.json 2> $O/tl_$U.err || tail -3 $O/tl_$U.err
  python3 - $O/tl_$U.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("blocks", "update_groups", "span_us", "update_dur_us_pct", "update_end_us_pct", "bin_start_us_pct", "bin_dur_us_pct", "bin_end_us_pct")})
print([ (r["t"], r["update"], r["bin"]) for r in d["resident_by_us"][::3]])
PY
done
