"""bench.py's command line (no GPU): what a step is for each workload, and the defaults the driver relies on
(`python bench.py` alone must pick N = 1 and a K / W that finish within minutes)."""
import os
import subprocess
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


def test_gpus_follows_the_launcher_when_it_is_not_given(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert parse([]).gpus == 4
    assert parse(["--gpus", "2"]).gpus == 2  # (main() then refuses: the launcher started another number of ranks)
    monkeypatch.delenv("WORLD_SIZE")
    assert parse([]).gpus == 1


def test_gpus_n_without_a_launcher_spawns_n_ranks_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus 8` the way the driver runs `--gpus 1`: no WORLD_SIZE in the environment, so bench.py
    itself starts the ranks through torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and leaves
    with the launcher's exit code — before it imports torch, let alone touches the GPU."""
    import bench
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 7

    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    sys.modules.pop("torch", None) if "torch" in sys.modules and False else None
    import pytest
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7  # a failing rank -> non-zero exit
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_must_match_gpus(monkeypatch):
    import bench
    import pytest
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.main()


def test_watchdog_hands_a_stall_to_its_callback():
    """tiling.Watchdog: a rank that makes no progress leaves — through `on_stall` when one is given (bench.py's riding
    global-map leg prints the replicas' line there and leaves with code 0), else with code 3."""
    code = ("import os, sys, time; sys.path.insert(0, %r); from fastdem_amd.tiling import Watchdog;"
            "w = Watchdog(0.3, what='t', on_stall=(lambda m: (print('STALL', m), sys.stdout.flush(), os._exit(0))) if sys.argv[1] == 'cb' else None);"
            "time.sleep(5); print('not reached')") % ROOT
    r = subprocess.run([sys.executable, "-c", code, "cb"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "STALL" in r.stdout and "not reached" not in r.stdout, (r.returncode, r.stdout, r.stderr)
    r = subprocess.run([sys.executable, "-c", code, "plain"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "no progress" in r.stderr, (r.returncode, r.stdout, r.stderr)
