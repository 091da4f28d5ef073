"""configs[3] with raycasting on, streamed through the batch entry point: option ray_overlap automatic (-1) against off (0).
    python scripts/ray_overlap_ab.py [distinct scans] [other engines alive]"""
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch
from fastdem_amd import synth
import bench
wl = synth.make("c4", n_scans=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
# other engines alive in the process (each holds a stream of its own: HIP maps streams onto a few hardware queues)
others = [bench.Resident(synth.make("c2"), 0) for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 0)]
if len(sys.argv) > 3 and sys.argv[3] == "host_legs":   # what bench.py has done in the process before its large raycasting leg
    big = bench.Resident(wl, 0)
    for k in range(8):
        big.step(k)
    big.eng.sync()
    legs = bench.host_legs(big, wl, 60, iters=12, stream_iters=40)
for ov in (-1, 0, -1, 0):
    r = bench.Resident(wl, 0)
    cfg = r.eng.cfg; cfg.raycast_enabled = 1; r.eng.set_config(cfg)
    r.eng.set_option("ray_overlap", ov)
    w, _ = r.batch(0, 4)
    assert r.eng.integrate_device_batch(w) == 0
    r.eng.sync()
    out = []
    for n in (12, 12, 32):
        b, _ = r.batch(4, n)
        assert r.eng.integrate_device_batch_timed(b) == 0
        out.append(round(r.eng.timer_ms() / n * 1e3, 1))
    print(json.dumps({"ray_overlap": ov, "us_per_scan_12_12_32": out}))
    del r
