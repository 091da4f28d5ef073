#!/usr/bin/env python3
"""bench.py with every engine's option ray_overlap set to argv[1] (-1 / 0 / 1): its large raycasting leg inside the bench's OWN
process (behind the host legs, with the other engines alive) — where the overlap loses: 339 -> 365 us per scan.
    python scripts/bench_overlap_ab.py 0; python scripts/bench_overlap_ab.py -1"""
import sys, os, json, io, contextlib
sys.path.insert(0, os.getcwd())
import bench
ov = int(sys.argv[1])
orig = bench.Resident.__init__
def init(self, *a, **k):
    orig(self, *a, **k)
    self.eng.set_option("ray_overlap", ov)
bench.Resident.__init__ = init
sys.argv = ["bench.py", "--no-cpu-baseline"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print(ov, d["large"]["raycasting_on"]["us_per_scan_hip_events"], d["raycasting_on"]["us_per_scan_hip_events"])
