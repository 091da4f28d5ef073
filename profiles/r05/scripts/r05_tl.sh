#!/bin/bash
# round 5: block timeline of the fused large-scan launch: r05_tl.sh "<opt=val,opt=val>" ...
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_tl
mkdir -p $O
cd $R
i=0
for V in "$@"; do
  i=$((i+1))
  SETS=""; for kv in ${V//,/ }; do SETS="$SETS --set $kv"; done
  timeout 300 python scripts/timeline.py c4 $SETS > $O/tl_$i.json 2> $O/tl_$i.err || tail -3 $O/tl_$i.err
  python3 - $O/tl_$i.json "$V" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], {k: d[k] for k in ("blocks", "update_groups", "span_us", "update_dur_us_pct", "update_end_us_pct", "bin_start_us_pct", "bin_dur_us_pct", "bin_end_us_pct")})
print([ (r["t"], r["update"], r["bin"]) for r in d["resident_by_us"][::3]])
print("longest update blocks", d["longest_update_groups"][:4])
PY
done
