"""bench.py end to end on the GPU box with N > 1: `--gpus 2` with NO launcher around it must start
its two ranks itself and print ONE JSON line that says n_gpus 2.  There is one GPU here and RCCL refuses two ranks on
one device, so both ranks compute on device 0 and the collectives are host-staged (gloo) — the rank plumbing, the tile
plan, the routing / pack / unpack kernels and the exchange loop are the ones the RCCL run uses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, timeout=900, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=env, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_2_global_map_unwrapped():
    """`python3 bench.py --gpus 2 --workload c5`: configs[4], one 8000 x 8000 global map cut into 1 x 2 tiles."""
    r = run_bench("--gpus", "2", "--workload", "c5", "--backend", "gloo", "--devices", "0,0", "--steps", "3", "--warmup", "1")
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["workload"].startswith("global_400x400m") and r["config"]["scans_per_step"] == 2
    assert r["roofline"] and r["roofline"]["achieved"] > 0 and 0 < r["roofline"]["frac"] < 1


def test_one_rank_rccl_line_certifies_its_communicator():
    """`--workload c5` with ONE rank runs its collectives over RCCL (a 1-rank communicator): the line says what RCCL saw."""
    r = run_bench("--workload", "c5", "--steps", "3", "--warmup", "1", "--global-size-m", "100")
    who = r["ranks"]
    assert who["backend"] == "nccl" and who["rccl_ranks"] == 1 and who["rccl_allreduce_sum"] == 1, who


def test_gpus_2_default_workload_prints_replicas_and_the_global_map():
    r = run_bench("--gpus", "2", "--backend", "gloo", "--devices", "0,0", "--steps", "4", "--warmup", "2",
                  "--no-cpu-baseline", "--no-host-legs")
    assert r["n_gpus"] == 2 and "replicas" in r["config"]["parallelism"]
    g = r["global_map"]
    assert g["n_gpus"] == 2 and g["roofline"]["achieved"] > 0 and g["value"] > 0
    # who ran the line: two ranks, both on the ONE device of this box (an 8-GPU run must read distinct_devices 8)
    who = r["ranks"]
    assert who["world_size"] == 2 and [x["rank"] for x in who["ranks"]] == [0, 1] and who["distinct_devices"] == 1
    assert all(x["cus"] == 256 and x["uuid"] for x in who["ranks"])


def test_a_failing_global_map_leg_does_not_take_the_replicas_line_with_it():
    """The global-map leg behind the replicas has never run on two devices.  Whatever happens in it — here rank 1 raises
    at its start (`--fail-global-rank 1`) — rank 0 still prints the replicas' line, with the error in `global_map`, the
    top-level `global_map_ok` false, and every rank leaves with code 0."""
    r = run_bench("--gpus", "2", "--backend", "gloo", "--devices", "0,0", "--steps", "4", "--warmup", "2",
                  "--no-cpu-baseline", "--no-host-legs", "--collective-timeout", "20", "--fail-global-rank", "1")
    assert r["n_gpus"] == 2 and r["value"] > 0 and "replicas" in r["config"]["parallelism"]
    assert "error" in r["global_map"] and r["global_map_ok"] is False


def test_gpus_8_global_map_plan_and_line_shape():
    """The command the driver runs on an 8-GPU node, with eight processes on the ONE GPU here (host-staged exchange)
    and a reduced global map: the 2 x 4 tile plan, eight scans per step, a roofline object — the rank plumbing, the
    plan, the routing kernels and the exchange loop of the first real 8-GPU run."""
    r = run_bench("--gpus", "8", "--workload", "c5", "--backend", "gloo", "--devices", "0,0,0,0,0,0,0,0", "--steps", "2",
                  "--warmup", "1", "--global-size-m", "100", timeout=1500)
    assert r["n_gpus"] == 8 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["tile_plan"] == "2x4" and r["config"]["scans_per_step"] == 8
    assert r["roofline"] and r["roofline"]["achieved"] > 0 and 0 < r["roofline"]["frac"] < 1
    m = r["rank0_routing_matrix_last_step"]
    assert m is not None and len(m) == 8  # every source's shares by owner


def test_gpus_8_default_workload_replicas_and_global_map():
    """`bench.py --gpus 8` (the default workload): eight replicas of configs[1] + the global-map leg behind them."""
    r = run_bench("--gpus", "8", "--backend", "gloo", "--devices", "0,0,0,0,0,0,0,0", "--steps", "4", "--warmup", "2",
                  "--no-cpu-baseline", "--no-host-legs", "--global-size-m", "100", timeout=1500)
    assert r["n_gpus"] == 8 and "replicas" in r["config"]["parallelism"] and r["value"] > 0
    assert r["global_map_ok"] is True
    g = r["global_map"]
    assert g["n_gpus"] == 8 and g["config"]["tile_plan"] == "2x4" and g["roofline"]["achieved"] > 0


def test_one_rank_of_the_n_rank_path_reproduces_the_default_line():
    """N = 1 through the launcher-less `--gpus 1` path prints the same figure as the plain command (within 5 %: the
    driver's `--steps 20 --warmup 5` region is repeated and the median reported)."""
    a = run_bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-host-legs", "--no-large")
    b = run_bench("--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-host-legs", "--no-large")
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["repeats"] >= 25 and b["repeats"] >= 25
    assert b["ranks"]["world_size"] == 1 and b["ranks"]["distinct_devices"] == 1 and b["ranks"]["ranks"][0]["cus"] == 256
    assert abs(a["value"] - b["value"]) / b["value"] < 0.05, (a["value"], b["value"])
