"""bench.py's command line (no GPU): what a step is for each workload, and the defaults the driver relies on
(`python bench.py` alone must pick N = 1 and a K / W that finish within minutes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(argv):
    import bench
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench.parse()
    finally:
        sys.argv = old


def test_defaults():
    a = parse([])
    assert (a.gpus, a.workload, a.steps, a.warmup) == (1, "c2", 625, 64)  # 625 steps x 16 scans = 10 000 scans
    assert a.scans_per_step == 0  # = per workload: 16 for the small-scan workloads, 1 for the large-scan ones


def test_driver_arguments_are_taken_literally():
    a = parse(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup) == (1, 20, 5)


def test_large_scan_workloads_step_per_scan():
    a = parse(["--workload", "c4"])
    assert (a.steps, a.warmup) == (10000, 1000)
    b = parse(["--workload", "c2", "--scans-per-step", "1"])
    assert (b.steps, b.warmup) == (10000, 1000)
