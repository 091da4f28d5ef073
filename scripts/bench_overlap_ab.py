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
