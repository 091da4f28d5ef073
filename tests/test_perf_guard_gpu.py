"""Performance guards (GPU): device-side time per scan of every BASELINE config against a GENEROUS ceiling — about
1.5 x what profiles/r05 holds.  Not a benchmark (bench.py is): a tripwire for an order-of-magnitude slip that no parity
test sees.  Round 5 shipped the reason for it: a tile-walk change left every layer bit-identical and configs[4] at
176 us per scan instead of 36, and only the end-of-round evidence pass noticed."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def device_us_per_scan(gpu, wl, warm, timed, raycast=0, **options):
    import bench
    res = bench.Resident(wl, 0)
    for k, v in options.items():
        res.eng.set_option(k, v)
    if raycast:
        cfg = res.eng.cfg
        cfg.raycast_enabled = 1
        res.eng.set_config(cfg)
    w, _ = res.batch(0, warm)
    assert res.eng.integrate_device_batch(w) == 0
    res.eng.sync()
    best = float("inf")
    for rep in range(3):  # (the best of three: a noisy neighbour must not fail the suite)
        b, _ = res.batch(warm + rep * timed, timed)
        assert res.eng.integrate_device_batch_timed(b) == 0
        best = min(best, res.eng.timer_ms() / timed * 1e3)
    return best


# (workload, scans, ceiling in us per scan, measured in profiles/r05)
CASES = [
    ("c2", dict(n_scans=16), 160, 2.0, "1.05"),
    ("c3", dict(n_scans=8), 96, 8.0, "4.8"),
    ("c4", dict(n_scans=4), 40, 48.0, "31"),
    ("c5", dict(n_scans=4), 40, 55.0, "36"),
]


@pytest.mark.parametrize("name,kw,timed,ceiling,measured", CASES, ids=[c[0] for c in CASES])
def test_integrate_stays_within_reach_of_the_measured_time(gpu, name, kw, timed, ceiling, measured):
    if gpu.Engine.default_options:
        pytest.skip("the engine's own pipeline choice only (the other fixture variant forces slower paths on purpose)")
    wl = gpu.synth.make(name, **kw)
    us = device_us_per_scan(gpu, wl, 32, timed)
    assert us < ceiling, f"{name}: {us:.1f} us per scan on the device (profiles/r05: {measured}; ceiling {ceiling})"


@pytest.mark.parametrize("name,kw,timed,ceiling,measured", [
    ("c2", dict(n_scans=16), 96, 16.0, "9.3"),
    ("c3", dict(n_scans=4), 16, 170.0, "100"),
    ("c4", dict(n_scans=3), 8, 520.0, "350"),
], ids=["c2", "c3", "c4"])
def test_raycasting_stays_within_reach_of_the_measured_time(gpu, name, kw, timed, ceiling, measured):
    if gpu.Engine.default_options:
        pytest.skip("the engine's own pipeline choice only")
    wl = gpu.synth.make(name, **kw)
    us = device_us_per_scan(gpu, wl, 8, timed, raycast=1)
    assert us < ceiling, f"{name} with raycasting: {us:.1f} us per scan (profiles/r05: {measured}; ceiling {ceiling})"
