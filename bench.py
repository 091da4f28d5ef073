#!/usr/bin/env python3
"""bench.py — M points/s integrated into the ElevationMap on MI355X, with roofline + CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]

A "step" is ONE pass of the hot path over one batch of synthetic input whose SoA channels are already
resident in HBM (through the C ABI; enqueue-only, no host sync inside the timed region).  For the small-scan
workloads (c2, c3) the hot path is the batch launch — fdm_engine_integrate_device_batch bins sixteen scans per
launch (fastdem_amd/csrc/fdm_multi.hpp) — so a step hands it SIXTEEN FastDEM::integrate() calls (config.scans_per_step);
the large-scan workloads (c4, c5) launch once per scan and a step is one scan.  `value` is points per second either
way, ms_per_step is per step; the single-scan latency path is reported beside it (latency_path).
Default workload = BASELINE.json configs[1]: VLP-16 ~30 K-pt scan into a 15x15 m @ 0.1 m LOCAL map, Kalman estimator.  N>1 (launched by torch.distributed.run): LOCAL
maps do not shard so every rank runs an independent replica (weak scaling, no
data-path collective); `--workload c5` instead tiles ONE global map across the ranks with an
RCCL halo exchange per scan (fastdem_amd/tiling.py).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     — dominant kernel: algorithmic bytes per launch / HIP-event duration vs 8 TB/s
  cpu_baseline — the CPU oracle ("port" of the reference path, 1 thread) timed on this host
  kernels      — both kernels' event-timed durations and algorithmic bytes
  large        — the same roofline measurement on configs[3] (2 M-pt scan), where an HBM
                 fraction is physically meaningful (a 30 K-pt scan is launch-latency bound)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MEASURED_READ_GBS = 6400.0  # what a plain 16 B/lane read kernel reaches on this machine (profiles/r01/hbm_bw.jsonl)
# distinct 2 M-point scans of the configs[3] legs: 9 x 33.5 MB = 302 MB of input cycle through the
# timed region, more than the 256 MiB Infinity Cache can hold, so the read stream comes from HBM
LARGE_SCANS = 9
STREAM_BYTES = 300 * 1024 * 1024  # input the small-scan workloads' timed region cycles through (> the 256 MiB Infinity Cache)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE when a launcher set it, else 1).  With N > 1 and no "
                         "launcher, bench.py starts the N ranks itself (torch.distributed.run) before it touches the GPU")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 / c5 runs: nccl = RCCL over xGMI; gloo = host-staged (tests: "
                         "several ranks on ONE device, which RCCL refuses)")
    ap.add_argument("--devices", default="",
                    help="comma-separated device of every local rank (default: local rank r -> device r); '0,0' = two ranks on device 0")
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 625 steps of 16 scans for the small-scan workloads, 10000 scans otherwise)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--scans-per-step", type=int, default=0,
                    help="scans one step hands to the hot path: 0 = 16 for the workloads the batch pipeline takes (c2, c3: a "
                         "step is ONE launch over one batch of 16 scans), 1 for the large-scan workloads (one launch per scan)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--native-routed", type=int, default=1, help="c5 routed: 1 = the step as one C call (libfdm_halo), 2 = that call pipelined over consecutive scans (one rank), 0 = the Python loop")
    ap.add_argument("--order", default="azimuth", choices=["azimuth", "ring"])
    ap.add_argument("--scans", type=int, default=0,
                    help="distinct synthetic scans resident in HBM (0 = per workload: enough that the 2 M-point "
                         "workloads stream > 256 MiB of input, i.e. cannot sit in the Infinity Cache)")
    ap.add_argument("--no-host-legs", action="store_true",
                    help="skip the PCIe-inclusive host entry points (reported beside `value`, never as it): they launch "
                         "the same kernels on host memory, which pulls a rocprofv3 per-kernel average of this "
                         "command away from the timed region")
    ap.add_argument("--global-size-m", type=float, default=400.0,
                    help="side of the GLOBAL map of configs[4] / the global_map leg (tests: a reduced map for N processes on one GPU)")
    ap.add_argument("--global-n-az", type=int, default=2048, help="azimuth steps of the reduced global map's scans")
    ap.add_argument("--fail-global-rank", type=int, default=-1,
                    help="tests: the global-map leg raises at its start on this rank (default: never)")
    ap.add_argument("--trace-steps", action="store_true", help="c5: host time of every routed step's parts on stderr (measurement)")
    ap.add_argument("--repeats", type=int, default=25,
                    help="a timed region shorter than 5 ms is repeated this many times; ms_per_step / value are the median")
    ap.add_argument("--profile-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true")
    ap.add_argument("--no-global-map", action="store_true",
                    help="N > 1 on a LOCAL-map workload: skip the second object of the line (`global_map`: BASELINE configs[4], "
                         "ONE global map tiled over the N GPUs with routed scans and a halo exchange per step)")
    ap.add_argument("--wave-merge", type=int, default=1)
    ap.add_argument("--set", action="append", default=[], help="engine option key=value (A/B switch)")
    ap.add_argument("--overlap", type=int, default=1, help="bin(t+1) || update(t) on two streams (A/B switch)")
    ap.add_argument("--routed", type=int, default=1,
                    help="c5: 1 = the tiled / routed multi-GPU path (also with one rank: a 1-rank RCCL communicator); 0 with "
                         "--gpus 1 = the whole 8000 x 8000 map on one plain engine (the single-GPU roofline of configs[4])")
    ap.add_argument("--collective-timeout", type=int, default=120,
                    help="seconds a collective may take before RCCL aborts it (a rank that failed leaves with a non-zero "
                         "exit code at once; its peers follow when this expires)")
    a = ap.parse_args()
    if a.gpus is None:
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    small = a.workload in ("c2", "c3") and (a.scans_per_step or 16) > 1
    if a.steps is None:
        a.steps = 625 if small else 10000
    if a.warmup is None:
        a.warmup = 64 if small else 1000
    return a


def colmajor16(T):
    import numpy as np
    a = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).T).reshape(16)
    return (C.c_double * 16)(*a.tolist())


class Resident:
    """One workload with its scans resident in HBM and a device-side engine."""

    def __init__(self, wl, device, wave_merge=1, overlap=1):
        import torch
        from fastdem_amd import Engine, capi
        self.wl = wl
        self.eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()),
                          device=device)
        self.eng.set_option("wave_merge", wave_merge)
        self.eng.set_option("overlap", overlap)
        self.dev = []
        for s in wl.scans:
            d = {k: (torch.from_numpy(v).to(f"cuda:{device}") if v is not None else None)
                 for k, v in s.items()}
            self.dev.append(d)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        self.args = [(d["x"].numel(), p(d["x"]), p(d["y"]), p(d["z"]), p(d["intensity"]), p(d["rgb"]))
                     for d in self.dev]
        self.tbs = colmajor16(wl.T_base_sensor)
        self.poses = {}
        self.n = wl.n_points

    def pose(self, k):
        if k not in self.poses:
            self.poses[k] = colmajor16(self.wl.pose(k))
        return self.poses[k]

    def step(self, k):
        a = self.args[k % len(self.args)]
        rc = self.eng.integrate_device_raw(a[0], a[1], a[2], a[3], self.tbs, self.pose(k), a[4], a[5])
        if rc != 0:
            raise RuntimeError(f"integrate_device failed: {rc}")
        return a[0]

    def batch(self, k0, count):
        """`count` consecutive steps k0.. as one fdm_device_scan array (built outside any timed region)."""
        from fastdem_amd import capi
        arr = (capi.FdmDeviceScan * count)()
        vp = lambda p: None if p is None else p.value  # noqa: E731
        for i in range(count):
            a = self.args[(k0 + i) % len(self.args)]
            d = arr[i]
            d.n = a[0]
            d.x, d.y, d.z, d.intensity, d.rgb, d.sigma_z2 = vp(a[1]), vp(a[2]), vp(a[3]), vp(a[4]), vp(a[5]), None
            d.T_base_sensor = self.tbs
            d.T_world_base = self.pose(k0 + i)
        return arr, sum(self.args[(k0 + i) % len(self.args)][0] for i in range(count))

    def bytes_per_point(self):
        s = self.wl.scans[0]
        return 12 + (4 if s["intensity"] is not None else 0) + (4 if s["rgb"] is not None else 0)

    def bytes_per_cell(self):
        s = self.wl.scans[0]
        b = 124 if self.wl.estimation_type == 1 else 72
        return b + (8 if s["intensity"] is not None else 0) + (4 if s["rgb"] is not None else 0)


def measure_kernels(res, k0, steps, tag=None, overlap=1):
    """HIP-event durations of the launches (events recorded on the engine's stream around each
    launch), averaged over `steps` scans, plus the algorithmic bytes each launch moves.
    Phase A times k_bin / k_update on their own (hold-back off): the per-kernel breakdown.
    Phase B (overlap on) times what the timed region of main() actually runs: ONE launch per scan,
    k_update_bin = update of scan t + bin of scan t+1."""
    eng = res.eng
    eng.set_option("overlap", 0)
    eng.enable_profile(True)
    t_bin = t_upd = 0.0
    touched = 0
    pts = 0
    for i in range(steps):
        pts += res.step(k0 + i)
        b, u = eng.last_kernel_ms()
        rc, st = eng.last_stats()
        t_bin += b
        t_upd += u
        touched += st["n_cells_touched"]
    ms_bin, ms_upd = t_bin / steps, t_upd / steps
    n_per = pts / steps
    cells_total = eng.s_rows * eng.s_cols
    bytes_bin = n_per * res.bytes_per_point()
    # the per-scan dense term (4 B per map cell: the obstacle layer rewritten every scan) belongs to
    # LOCAL rolling maps.  On a GLOBAL map the engine sweeps only the tiles a scan stamped, so nothing map-sized has
    # to move: the dense term is left out (a lower bound — the reported fraction errs on the low side).
    bytes_upd = (touched / steps) * res.bytes_per_cell() + (0 if res.wl.mode == 1 else cells_total * 4)
    out = {
        "k_bin": {"ms": ms_bin, "alg_bytes": bytes_bin, "GBps": bytes_bin / (ms_bin * 1e-3) / 1e9},
        "k_update": {"ms": ms_upd, "alg_bytes": bytes_upd, "GBps": bytes_upd / (ms_upd * 1e-3) / 1e9},
        "touched_cells_per_scan": touched / steps, "points_per_scan": n_per,
        "alg_bytes_per_scan": bytes_bin + bytes_upd,
    }
    dom = "k_bin" if ms_bin >= ms_upd else "k_update"
    if overlap:
        eng.set_option("overlap", 1)
        res.step(k0 + steps)  # opens the chain: a plain bin launch, its update is held back
        t_f = 0.0
        for i in range(steps):
            res.step(k0 + steps + 1 + i)
            t_f += eng.last_kernel_ms()[0]  # (does not flush: the chain stays intact)
        ms_f = t_f / steps
        out["k_update_bin"] = {"ms": ms_f, "alg_bytes": bytes_bin + bytes_upd,
                               "GBps": (bytes_bin + bytes_upd) / (ms_f * 1e-3) / 1e9}
        dom = "k_update_bin"
    eng.enable_profile(False)
    eng.set_option("overlap", overlap)
    tiled = eng.last_pipeline() == 1  # maps of >= 240 tiles: k_tbin / k_tupdate
    real = {"k_bin": "k_tbin", "k_update": "k_tupdate", "k_update_bin": "k_tupdate_tbin"} if tiled else {}
    out["kernel_names"] = {k: real.get(k, k) for k in ("k_bin", "k_update", "k_update_bin") if k in out}
    names = {"k_update_bin": "k_update_bin (one launch: update of scan t + bin of scan t+1)",
             "k_tupdate_tbin": "k_tupdate_tbin (one launch: update of scan t + bin of scan t+1, per-tile record pools)"}
    # `bound`: the roofline the fraction is taken against (the contract's "hbm": no contraction on this path).
    # `limited_by`: what the counters say the launch actually waits for (profiles/r05: the large-scan launch is
    # instruction-issue / dependent-round-trip bound at ~20 % of the HBM roofline; the small-scan kernels are latency chains)
    roof = {"bound": "hbm", "limited_by": "issue" if tiled else "latency",
            "kernel": names.get(real.get(dom, dom), real.get(dom, dom)), "achieved": out[dom]["GBps"],
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": out[dom]["GBps"] / HBM_PEAK_GBS,
            "traffic": pmc_traffic(tag, real.get(dom, dom)),
            "avg_kernel_us": out[dom]["ms"] * 1e3,
            "alg_bytes_per_launch": out[dom]["alg_bytes"],
            # which terms of SURVEY.md §8d the algorithmic bytes hold (a GLOBAL map has no per-scan whole-layer clear to
            # count: its fraction is a lower bound and is NOT comparable with a LOCAL map's)
            "alg_bytes_formula": ("points x %d B + touched cells x %d B" % (res.bytes_per_point(), res.bytes_per_cell())) +
                                 ("" if res.wl.mode == 1 else " + map cells x 4 B (obstacle layer rewritten per scan)"),
            "dense_term_included": res.wl.mode != 1}
    return out, roof


def pmc_traffic(tag, kernel):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, made by
    scripts/r02_profile.sh + scripts/pmc_traffic.py: read bytes from the request-size counters
    128 * RDREQ_128B + 64 * RDREQ_64B + 32 * RDREQ_32B — calibrated on known-byte kernels,
    profiles/r02/pmc_calibration.json — plus WRITE_SIZE).  PMC cannot be collected from inside this process,
    so this is null when no profile of the workload has been committed."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return d[tag][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def parity_gate(wl, R):
    """SURVEY.md §8d "parity gate run with every benchmark": the engine that was just timed against the
    CPU oracle (the checker) on the same workload after 1, 2 and 10 scans — cell ids bit-exact for every
    point, every layer NaN-pattern identical and within 1e-5 relative (abs floor 1e-7)."""
    import numpy as np
    from fastdem_amd import Engine, capi
    from __graft_entry__ import compare_layers
    eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()))
    ref = R.RefEngine(wl.width, wl.height, wl.resolution, wl.apply_to(R.default_config()))
    eng.enable_cell_ids()
    ref.enable_cell_ids()
    worst, checked = 0.0, []
    for k in range(10):
        s = wl.scan(k)
        kw = {c: s[c] for c in ("intensity", "rgb") if s.get(c) is not None}
        rc_e, st_e = eng.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), **kw)
        rc_r, st_r = ref.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), **kw)
        if rc_e != rc_r or st_e != st_r:
            raise SystemExit(f"parity gate: status/statistics differ at scan {k}: {rc_e} {st_e} vs {rc_r} {st_r}")
        n = s["x"].size
        if not np.array_equal(eng.last_cell_ids(n), ref.last_cell_ids(n)):
            raise SystemExit(f"parity gate: cell ids differ at scan {k}")
        if k + 1 in (1, 2, 10):
            worst = max(worst, compare_layers(eng, ref))  # asserts NaN pattern + tolerance
            checked.append(k + 1)
    # ... and the path that was TIMED: scans resident in HBM, enqueue-only, one fused launch per scan with the
    # LEAN bin half (no cell ids asked for), against the same oracle state after the 10 scans
    import torch
    res = Resident(wl, torch.cuda.current_device())
    for k in range(10):
        res.step(k)
    worst_timed_path = compare_layers(res.eng, ref)
    # ... and as ONE fdm_engine_integrate_device_batch call (small scans: the batch launches of fdm_multi.hpp)
    res2 = Resident(wl, torch.cuda.current_device())
    arr, _ = res2.batch(0, 10)
    if res2.eng.integrate_device_batch(arr) != 0:
        raise SystemExit("parity gate: integrate_device_batch failed")
    batched = res2.eng.last_batch()
    worst_batch = compare_layers(res2.eng, ref)
    return {"after_scans": checked, "cell_ids": "bit-exact", "layers_max_rel_err": worst, "rtol": 1e-5,
            "layers": len(ref.layers()), "timed_path_after_10_scans_max_rel_err": worst_timed_path,
            "batch_call_after_10_scans_max_rel_err": worst_batch, "batch_call_scans_per_launch": batched}


PROFILER_VARS = ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY", "HSA_TOOLS_LIB", "ROCP_TOOL_LIB")


def under_profiler():
    """rocprofv3 preloads its tool library into this process (and into anything it spawns): a make -> sh -> g++
    chain behind it is exactly the exec-behind-a-GPU-initialised-process pattern the pool forbids."""
    return bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower()


def clean_env():
    return {k: v for k, v in os.environ.items() if k not in PROFILER_VARS}


def cpu_baseline(wl, target_s=12.0):
    """The CPU oracle (single-threaded port of the reference path, -O3 no -march: the reference's
    Release flags) timed on this host on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fdm_ref_py as R
    ref = R.RefEngine(wl.width, wl.height, wl.resolution, wl.apply_to(R.default_config()))
    s = wl.scan(0)
    poses = [wl.pose(k) for k in range(64)]
    kw = dict(intensity=s["intensity"], rgb=s["rgb"])
    ref.time_integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, poses, 5, **kw)  # warm-up
    t1 = ref.time_integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, poses, 5, **kw) / 5
    iters = max(5, min(20000, int(target_s / max(t1, 1e-6))))
    # SURVEY.md §8d policy (nanoPCL benchmark_common.hpp:56-60): warm-up, then 50 timed samples ->
    # median, mean +- CI95; a sample = iters/50 consecutive integrate() calls
    per = max(1, iters // 50)
    samples, stages = [], None
    for _ in range(50):
        d, st = ref.time_integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, poses, per, stages=True, **kw)
        samples.append(d / per)
        stages = st if stages is None else [a + b for a, b in zip(stages, st)]
    iters = per * 50
    dt = float(sum(samples)) * per
    samples.sort()
    mean = sum(samples) / len(samples)
    sd = (sum((v - mean) ** 2 for v in samples) / (len(samples) - 1)) ** 0.5
    n = int(s["x"].size)
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    names = ["sensor_cov", "transform_filter", "cov_transform", "rasterize", "map_update"]
    # courtesy row: the same source built with -march=native
    native = None
    try:
        # -march=native means THIS host: rebuild the library here (8 s of g++), never trust a shipped one
        import subprocess
        if under_profiler():
            raise RuntimeError("no child processes behind a profiler preload")
        subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "_build/libfdm_ref_native.so"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, env=clean_env())
        rn = R.RefEngine(wl.width, wl.height, wl.resolution, wl.apply_to(R.default_config()), native=True)
        rn.time_integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, poses, 5, **kw)
        it2 = max(5, iters // 4)
        native = n * it2 / rn.time_integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, poses, it2, **kw) / 1e6
    except Exception:
        pass
    return {"parity": parity_gate(wl, R),
            "value": n * iters / dt / 1e6, "unit": "Mpts/s", "cores": 1, "kind": "port",
            "march_native_value": native,  # same source, -march=native, built on this host just now (None: no g++)
            "ms_per_scan": dt / iters * 1e3,
            "ms_per_scan_median": samples[len(samples) // 2] * 1e3,
            "ms_per_scan_ci95": 1.96 * sd / len(samples) ** 0.5 * 1e3,
            "sample": f"{iters} integrate() calls of the {n}-pt scan ({dt:.1f} s), "
                      f"oracle/libfdm_ref.so -O3 no -march, 1 thread",
            "stage_ms": {k: float(v) / iters * 1e3 for k, v in zip(names, stages)},
            "host_cpu": cpu_model, "host_nproc": os.cpu_count()}


def host_legs(res, wl, k, iters=50, stream_iters=200):
    """the reference's own call shape — host arrays in — on the same workload.
    PCIe-inclusive, reported beside `value`, never as it: synchronous fdm_engine_integrate from pageable / pinned /
    pooled-pinned arrays (median of `iters` calls) and the steady stream fdm_engine_integrate_async from pinned memory."""
    import numpy as np
    import torch
    from fastdem_amd import host_array
    s = wl.scan(0)
    out = {}

    def median_ms(fn):  # (a median: one stalled call must not move a 0.1 ms figure)
        ts = []
        for i in range(iters):
            t0 = time.perf_counter()
            fn(i)
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    out["host_buffers_ms_per_scan"] = median_ms(
        lambda i: res.eng.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k + 1000 + i),
                                    intensity=s["intensity"], rgb=s["rgb"]))
    pin = {c: torch.from_numpy(s[c]).pin_memory() for c in ("x", "y", "z", "intensity", "rgb") if s.get(c) is not None}
    pn = {c: t.numpy() for c, t in pin.items()}
    # the same synchronous call on PINNED arrays: read in place by the bin kernel, no copy commands
    for i in range(3):  # first GPU touch of freshly pinned pages is not what is measured
        res.eng.integrate(pn["x"], pn["y"], pn["z"], wl.T_base_sensor, wl.pose(k + 1000 + i),
                          intensity=pn.get("intensity"), rgb=pn.get("rgb"))
    # ... and on arrays from the engine's own pinned pool (fdm_host_alloc: what every channel of the C++ mirror's
    # nanopcl::PointCloud is made of): their device aliases come from the pool's table
    pool = {c: host_array(s[c], np.uint32 if c == "rgb" else np.float32)
            for c in ("x", "y", "z", "intensity", "rgb") if s.get(c) is not None}
    pa = {c: h.array for c, h in pool.items()}
    for i in range(3):
        res.eng.integrate(pa["x"], pa["y"], pa["z"], wl.T_base_sensor, wl.pose(k + 1000 + i),
                          intensity=pa.get("intensity"), rgb=pa.get("rgb"))
    out["host_buffers_pool_ms_per_scan"] = median_ms(
        lambda i: res.eng.integrate(pa["x"], pa["y"], pa["z"], wl.T_base_sensor, wl.pose(k + 1000 + i),
                                    intensity=pa.get("intensity"), rgb=pa.get("rgb")))
    out["host_buffers_pinned_ms_per_scan"] = median_ms(
        lambda i: res.eng.integrate(pn["x"], pn["y"], pn["z"], wl.T_base_sensor, wl.pose(k + 1000 + i),
                                    intensity=pn.get("intensity"), rgb=pn.get("rgb")))
    # the reference's own cloud layout: {x, y, z, 1} records (nanopcl::PointCloud::points()), pinned pool memory,
    # through fdm_engine_integrate_points4 — no host AoS -> SoA loop; 16 B per point over PCIe instead of 12
    n_pts = int(s["x"].size)
    aos = host_array(np.stack([s["x"], s["y"], s["z"], np.ones(n_pts, np.float32)], axis=1).reshape(-1), np.float32)
    a4 = aos.array.reshape(n_pts, 4)
    for i in range(3):
        res.eng.integrate_points4(a4, wl.T_base_sensor, wl.pose(k + 1000 + i), intensity=pa.get("intensity"), rgb=pa.get("rgb"))
    out["host_points4_pool_ms_per_scan"] = median_ms(
        lambda i: res.eng.integrate_points4(a4, wl.T_base_sensor, wl.pose(k + 1000 + i), intensity=pa.get("intensity"),
                                            rgb=pa.get("rgb")))
    # N integrate() calls on host clouds as ONE call (fdm_engine_integrate_host_batch: what the C++ mirror's
    # FastDEM::integrateBatch does): 64 scans from the pinned pool, read in place by the batch launches, one wait at the end
    from fastdem_amd import capi
    nb = 64
    harr = (capi.FdmDeviceScan * nb)()
    for i in range(nb):
        d = harr[i]
        d.n = int(s["x"].size)
        d.x, d.y, d.z = pa["x"].ctypes.data, pa["y"].ctypes.data, pa["z"].ctypes.data
        d.intensity = pa["intensity"].ctypes.data if "intensity" in pa else None
        d.rgb = pa["rgb"].ctypes.data if "rgb" in pa else None
        d.sigma_z2 = None
        d.T_base_sensor = res.tbs
        d.T_world_base = res.pose(k + 3000 + i)
    res.eng.integrate_host_batch(harr)  # (first call: buffers)
    ts = []
    for _ in range(max(3, iters // 8)):
        t0 = time.perf_counter()
        res.eng.integrate_host_batch(harr)
        ts.append((time.perf_counter() - t0) / nb)
    out["host_batch_pool_ms_per_scan"] = sorted(ts)[len(ts) // 2] * 1e3
    # steady-state stream from PINNED host memory, no per-scan wait (SURVEY.md §8d iii): PCIe-inclusive
    hp = {c: C.c_void_p(t.data_ptr()) for c, t in pin.items()}
    for i in range(64):  # pose matrices are host work that does not belong to the stream's rate
        res.pose(k + 2000 + i)
    for i in range(min(20, stream_iters)):
        res.eng.integrate_async_raw(s["x"].size, hp["x"], hp["y"], hp["z"], res.tbs, res.pose(k + 2000 + i),
                                    hp.get("intensity"), hp.get("rgb"))
    res.eng.sync()
    t0 = time.perf_counter()
    for i in range(stream_iters):
        res.eng.integrate_async_raw(s["x"].size, hp["x"], hp["y"], hp["z"], res.tbs, res.pose(k + 2000 + i % 60),
                                    hp.get("intensity"), hp.get("rgb"))
    res.eng.sync()
    out["host_stream_pinned_ms_per_scan"] = (time.perf_counter() - t0) / stream_iters * 1e3
    return out


def rank_identity(args, rank, world, device):
    """Who ran this line — so that the first run on N devices certifies itself: per rank the device it
    computed on (index, PCI bus id, UUID, name) and host; `rccl_ranks` = the size RCCL ITSELF reports for a communicator
    built over the job's ranks (ncclCommCount on fastdem_amd.halo.make_comm's communicator — the one libfdm_halo's
    collectives use), `rccl_allreduce_sum` = a device all-reduce of ones through torch's process group (== N iff every
    rank's GPU took part), `distinct_devices` = distinct UUIDs over the ranks.  Collective: every rank calls it."""
    import socket
    import torch
    import torch.distributed as dist
    p = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "device": int(device), "name": p.name, "uuid": str(getattr(p, "uuid", "")),
          "pci_bus_id": f"{getattr(p, 'pci_domain_id', 0):04x}:{getattr(p, 'pci_bus_id', 0):02x}:{getattr(p, 'pci_device_id', 0):02x}",
          "cus": int(p.multi_processor_count), "host": socket.gethostname()}
    out = {"backend": args.backend if world > 1 or dist.is_initialized() else "none", "world_size": world}
    ranks = [me]
    if dist.is_initialized():
        box = [None] * world
        dist.all_gather_object(box, me)
        ranks = box
        if args.backend == "nccl":
            ones = torch.ones(1, dtype=torch.int32, device=f"cuda:{device}")
            dist.all_reduce(ones)
            out["rccl_allreduce_sum"] = int(ones.item())
            try:
                import ctypes as C
                from fastdem_amd import halo
                comm = halo.make_comm(rank, world, dist)
                n = C.c_int(-1)
                nccl = halo._rccl()
                nccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
                rc = nccl.ncclCommCount(comm, C.byref(n))
                out["rccl_ranks"] = int(n.value) if rc == 0 else f"ncclCommCount failed: {rc}"
                halo.destroy_comm(comm)
            except Exception as e:  # (the line is printed whatever happens here)
                out["rccl_ranks"] = f"unavailable: {type(e).__name__}: {e}"
    out["distinct_devices"] = len({(r["host"], r["uuid"] or r["pci_bus_id"]) for r in ranks})
    out["ranks"] = ranks
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here — one process per GPU through
    torch.distributed.run, rendezvous on 127.0.0.1 — BEFORE this process has touched the GPU (it never does: it only
    waits).  Rank 0's JSON line is the children's stdout; any rank that fails makes the launcher, and this process,
    exit non-zero."""
    import socket
    import subprocess
    if under_profiler():
        raise SystemExit("bench.py --gpus N under a profiler: launch the ranks with torch.distributed.run yourself "
                         "(no child processes behind a profiler preload)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    device = local_rank
    if args.devices:
        devs = [int(d) for d in args.devices.split(",")]
        if len(devs) < world:
            raise SystemExit(f"bench.py: --devices names {len(devs)} devices for {world} ranks")
        device = devs[local_rank]
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(device)
    local_rank = device  # (from here on: the device this rank computes on)
    routed = args.workload == "c5" and (world > 1 or args.routed)
    if world > 1 or routed:
        # (c5 runs its collectives over RCCL also with ONE rank: the N-rank code path, a 1-rank communicator)
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")  # a dead peer aborts the collective instead of hanging it
        kw_pg = {"device_id": torch.device(f"cuda:{device}")} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=args.collective_timeout), **kw_pg)
        if args.backend == "gloo":
            args.native_routed = 0  # (libfdm_halo's routed step talks RCCL: the host-staged Python loop instead)
        # The first collective builds the communicator (RCCL: ~50 ms of device-side set-up that the call only enqueues):
        # it happens HERE, not behind the barrier in front of a timed region (measured: the first engine call after an
        # un-warmed barrier blocked for 50 ms — 100 one-rank c5 steps read 0.54 ms instead of 0.037 ms per step)
        dist.barrier()
        torch.cuda.synchronize()
    from fastdem_amd import synth

    if routed:
        from fastdem_amd import tiling
        args.stall_timeout = args.collective_timeout + 60
        result = tiling.bench_global(args, rank, local_rank, world)
    else:
        kw = {"order": args.order} if args.workload in ("c2", "c4") else {}
        if args.scans or args.workload in ("c4", "c5"):
            kw["n_scans"] = args.scans or LARGE_SCANS
        else:
            # the small-scan workloads: enough DISTINCT scans that the timed region's input (> 256 MiB) cannot sit in the
            # Infinity Cache — `value` is measured on inputs that stream from HBM (with four scans cycling
            # through a batch the input lived in L2 / MALL); the cache-resident figure is reported beside it
            probe = synth.make(args.workload, n_scans=1, **kw)
            per_scan = probe.n_points * (12 + (4 if probe.scans[0]["intensity"] is not None else 0) +
                                         (4 if probe.scans[0]["rgb"] is not None else 0))
            kw["n_scans"] = int(min(640, max(4, -(-STREAM_BYTES // per_scan))))
        wl = synth.make(args.workload, **kw)
        res = Resident(wl, local_rank, args.wave_merge, args.overlap)
        for kv in args.set:
            res.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))

        def barrier():
            res.eng.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()  # (the barrier's own device work is done before a timed region starts)

        # host-side pose matrices for every step are built BEFORE the timed region (numpy 4x4
        # products cost more than the two kernel launches of a 30 K-point scan)
        # A STEP = one pass of the hot path over one batch of synthetic input.  For the small-scan workloads the hot
        # path is the batch launch (fdm_multi.hpp: k_mbatch over 16 scans), so a step hands it 16 scans; the large-scan
        # workloads launch once per scan.  `value` is points per second either way; ms_per_step is per STEP.
        sps = args.scans_per_step or (16 if args.workload in ("c2", "c3") else 1)
        n_timed, n_warm = args.steps * sps, args.warmup * sps

        def timed_region(r, k0):
            """warm-up, then the K timed steps as ONE call across the language boundary (fdm_engine_integrate_device_batch:
            K x fdm_engine_integrate_device in C++; with a Python / ctypes call per 6 us scan the region would measure the
            interpreter), K scans + the last scan's held-back update between two HIP events on the engine's stream.
            A region shorter than 5 ms (the driver's `--steps 20 --warmup 5` is 0.4 ms at configs[1]) is one launch latency
            and one sync away from noise: the IDENTICAL K-step region — barrier + synchronize on both sides each time —
            is then repeated `args.repeats` times on fresh scans and the MEDIAN is reported, with min / max beside it.
            Returns (wall seconds incl. the final sync [median], points, HIP-event us per scan, scans per launch, next k,
            the list of wall seconds of every repeat)."""
            reps_max = max(1, args.repeats)
            for kk in range(k0, k0 + n_warm + n_timed * reps_max + 8):
                r.pose(kk)
            k = k0
            if n_warm > 0:  # the warm-up steps take the timed region's own entry point (its first call is not free)
                wbatch, _ = r.batch(k, n_warm)
                if r.eng.integrate_device_batch_timed(wbatch) != 0:
                    raise RuntimeError("integrate_device_batch (warm-up) failed")
                k += n_warm

            def one(k_):
                batch, pts_ = r.batch(k_, n_timed)
                r.eng.sync()
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                    torch.cuda.synchronize()
                t0_ = time.perf_counter()
                rc_ = r.eng.integrate_device_batch_timed(batch)
                torch.cuda.synchronize()  # (device-wide: covers the engine's stream)
                dt_ = time.perf_counter() - t0_
                if rc_ != 0:
                    raise RuntimeError(f"integrate_device_batch failed: {rc_}")
                return dt_, pts_, r.eng.timer_ms() / n_timed * 1e3, max(1, r.eng.last_batch())

            dt_, pts_, us_, bs_ = one(k)
            k += n_timed
            wall, dev = [dt_], [us_]
            again = 1 if (dt_ < 5e-3 and reps_max > 1) else 0
            if world > 1:  # every rank repeats or none does
                flag = torch.tensor([again], dtype=torch.int32, device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                again = int(flag.item())
            if again:
                for _ in range(reps_max - 1):
                    d_, _, u_, _ = one(k)
                    k += n_timed
                    wall.append(d_)
                    dev.append(u_)
            return float(np.median(wall)), pts_, float(np.median(dev)), bs_, k, wall

        for kk in range(args.profile_steps + 8):
            res.pose(n_warm + n_timed + kk)
        dt, pts, timed_launch_us, batch_scans, k, wall_reps = timed_region(res, 0)
        if world > 1:  # the MAX over ranks of every repeat, then the median
            dist.barrier()
            t = torch.tensor(wall_reps, dtype=torch.float64, device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall_reps = [float(v) for v in t.tolist()]
            dt = float(np.median(wall_reps))
        rc, st = res.eng.last_stats()
        assert rc == 0 and st["n_in_map"] > 0, (rc, st)
        total_pts = pts * world
        result = {
            "metric": "M points/s integrated into ElevationMap",
            "value": total_pts / dt / 1e6, "unit": "Mpts/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            # (a region under 5 ms is repeated: ms_per_step / value are the MEDIAN over `repeats` identical K-step regions)
            "repeats": len(wall_reps), "ms_per_step_min": min(wall_reps) / args.steps * 1e3,
            "ms_per_step_max": max(wall_reps) / args.steps * 1e3,
            # the same K steps by the GPU's own clock (HIP events on the engine's stream around the region);
            # `value` is the host wall clock incl. the final sync — for very short regions it is launch-bound
            "device_value": pts / (timed_launch_us * n_timed * 1e-6) / 1e6,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl.name, "points_per_scan": wl.n_points,
                       "map_cells": res.eng.rows * res.eng.cols, "point_order": args.order,
                       "parallelism": "replicas only (LOCAL map does not shard)" if world > 1 else "1 gpu",
                       "inputs": (f"SoA float32 resident in HBM; `value` cycles through {len(wl.scans)} distinct scans = "
                                  f"{len(wl.scans) * res.bytes_per_point() * wl.n_points / 2**20:.0f} MiB of input"
                                  + (" (more than the 256 MiB Infinity Cache: the reads come from HBM); `cache_resident` = the "
                                     "same region over 4 scans that stay in L2 / Infinity Cache"
                                     if len(wl.scans) * res.bytes_per_point() * wl.n_points > 2**28 else "")),
                       "wave_merge": args.wave_merge,
                       "distinct_scans": len(wl.scans), "scans_per_launch": batch_scans, "scans_per_step": sps,
                       "step": (f"one launch of the batch pipeline over {sps} scans (fdm_engine_integrate_device_batch)"
                                if sps > 1 else "one scan")},
        }
        if rank == 0:
            kern, roof = measure_kernels(res, k, args.profile_steps, args.workload, args.overlap)
            if args.overlap and "k_update_bin" in kern and batch_scans == 1:
                # the timed region is nothing but back-to-back k_update_bin launches, one per scan:
                # its HIP-event time / steps IS that kernel's average duration in the run that was
                # measured (rocprofv3 --stats of the same command shows the same average); the
                # event-pair-per-launch figures of measure_kernels() time launches in isolation.
                kern["k_update_bin"]["ms_isolated"] = kern["k_update_bin"]["ms"]
                kern["k_update_bin"]["ms"] = timed_launch_us * 1e-3
                gbps = kern["k_update_bin"]["alg_bytes"] / (timed_launch_us * 1e-6) / 1e9
                kern["k_update_bin"]["GBps"] = gbps
                roof.update({"achieved": gbps, "frac": gbps / HBM_PEAK_GBS, "avg_kernel_us": timed_launch_us,
                             "measured": "HIP events on the engine stream around the timed region / steps"})
            if batch_scans > 1:
                # The timed region was the BATCH pipeline: back-to-back k_mbatch launches, each the bin of 16 scans, the
                # map update of the previous 16 and the scout blocks of the next 16.  Algorithmic bytes per launch = 16 x the
                # per-scan figure; duration = HIP events around the region / launches (gaps between launches included).
                launches = -(-n_timed // batch_scans)
                launch_us = timed_launch_us * n_timed / launches
                alg = kern["k_update_bin"]["alg_bytes"] * n_timed / launches
                gbps = alg / (launch_us * 1e-6) / 1e9
                kern["k_mbatch"] = {"ms": launch_us * 1e-3, "alg_bytes": alg, "GBps": gbps, "scans_per_launch": batch_scans,
                                    "launches": launches}
                kern["kernel_names"]["k_mbatch"] = "k_mbatch"
                # what one scan costs on its own (one fused launch per scan, the path a 10 Hz caller takes)
                result["latency_path"] = {"kernel": roof["kernel"], "us_per_scan": kern["k_update_bin"]["ms"] * 1e3,
                                          "Mpts_per_s": wl.n_points / (kern["k_update_bin"]["ms"] * 1e-3) / 1e6,
                                          "note": "fdm_engine_integrate_device scan by scan (event pair per launch)"}
                roof = {"bound": "hbm", "limited_by": "latency",
                        "kernel": "k_mbatch (one launch per 16 scans: update of batch b-1 | bin of batch b | scout blocks of batch b+1)",
                        "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
                        "traffic": pmc_traffic(args.workload, "k_mbatch"), "avg_kernel_us": launch_us,
                        "alg_bytes_per_launch": alg, "scans_per_launch": batch_scans,
                        "alg_bytes_formula": "%d scans x (%s)" % (batch_scans, roof["alg_bytes_formula"]),
                        "dense_term_included": roof["dense_term_included"],
                        "measured": "HIP events on the engine stream around the timed region / launches"}
            roof["frac_of_measured_read_bw"] = roof["achieved"] / MEASURED_READ_GBS
            result["roofline"] = roof
            result["kernels"] = kern
            if world == 1 and len(wl.scans) > 4 and args.workload in ("c2", "c3") and not args.scans:
                # round 3's protocol beside it: four distinct scans cycling through the batches (input in L2 / MALL)
                wl4 = synth.make(args.workload, n_scans=4, **{k_: v for k_, v in kw.items() if k_ != "n_scans"})
                res4 = Resident(wl4, local_rank, args.wave_merge, args.overlap)
                for kv in args.set:
                    res4.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
                dt4, pts4, us4, bs4, _, _ = timed_region(res4, 0)
                result["cache_resident"] = {"value": pts4 / dt4 / 1e6, "unit": "Mpts/s", "device_value": wl4.n_points / (us4 * 1e-6) / 1e6,
                                            "us_per_scan_hip_events": us4, "distinct_scans": 4, "scans_per_launch": bs4,
                                            "roofline_frac": roof["alg_bytes_per_launch"] / max(1, roof.get("scans_per_launch", 1)) /
                                                             (us4 * 1e-6) / 1e9 / HBM_PEAK_GBS}
                del res4
            result["timed_region_us_per_scan_hip_events"] = timed_launch_us
            if not args.no_host_legs:
                # end-to-end with host staging (PCIe-inclusive) for DESIGN.md — never `value`
                result.update(host_legs(res, wl, k))
            if world == 1 and not args.no_host_legs and args.workload in ("c2", "c3"):
                # the same stream with the shipped YAML's raycasting switch on (fastdem/config/default.yaml:40-41;
                # SURVEY.md §8 f1): small scans carry the stage inside the batches (fdm_rbatch.hpp)
                ray = Resident(wl, local_rank, args.wave_merge, args.overlap)
                for kv in args.set:
                    ray.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
                cfg_r = ray.eng.cfg
                cfg_r.raycast_enabled = 1
                ray.eng.set_config(cfg_r)
                n_ray = 320 if wl.n_points <= 65536 else 24
                wr, _ = ray.batch(0, 48 if wl.n_points <= 65536 else 8)
                if ray.eng.integrate_device_batch(wr) != 0:
                    raise RuntimeError("integrate_device_batch (raycasting leg warm-up) failed")
                ray.eng.sync()
                br, _ = ray.batch(48, n_ray)
                launches0 = sum(ray.eng.batch_launches())
                if ray.eng.integrate_device_batch_timed(br) != 0:
                    raise RuntimeError("integrate_device_batch (raycasting leg) failed")
                ray_us = ray.eng.timer_ms() / n_ray * 1e3
                result["raycasting_on"] = {"us_per_scan_hip_events": ray_us, "Mpts_per_s": wl.n_points / (ray_us * 1e-6) / 1e6,
                                           "scans": n_ray, "in_batch_launches": sum(ray.eng.batch_launches()) > launches0,
                                           "note": "fdm_engine_integrate_device_batch with raycast_enabled: voxel filter + "
                                                   "ray walks + ghost resolution of every scan; never `value`"}
                del ray
            if world == 1 and not args.no_large and args.workload != "c4":
                big = Resident(synth.lidar128(n_scans=LARGE_SCANS), local_rank, args.wave_merge, args.overlap)
                for kv in args.set:
                    big.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
                # the same protocol as the headline: warm-up, then ONE batch call timed by the wall clock and by HIP
                # events on the engine's stream; the roofline of the fused launch from that region
                n_big, n_warm = 1000, 200
                wb, _ = big.batch(0, n_warm)
                if big.eng.integrate_device_batch(wb) != 0:
                    raise RuntimeError("integrate_device_batch (large leg warm-up) failed")
                big.eng.sync()
                bb, bpts = big.batch(n_warm, n_big)
                t0 = time.perf_counter()
                rcb = big.eng.integrate_device_batch_timed(bb)
                torch.cuda.synchronize()
                dtb = time.perf_counter() - t0
                if rcb != 0:
                    raise RuntimeError(f"integrate_device_batch (large leg) failed: {rcb}")
                big_us = big.eng.timer_ms() / n_big * 1e3
                kb, rb = measure_kernels(big, n_warm + n_big, 20, "c4")
                if "k_update_bin" in kb:
                    kb["k_update_bin"]["ms_isolated"] = kb["k_update_bin"]["ms"]
                    kb["k_update_bin"]["ms"] = big_us * 1e-3
                    gb = kb["k_update_bin"]["alg_bytes"] / (big_us * 1e-6) / 1e9
                    kb["k_update_bin"]["GBps"] = gb
                    rb.update({"achieved": gb, "frac": gb / HBM_PEAK_GBS, "avg_kernel_us": big_us,
                               "measured": "HIP events on the engine stream around the timed region / steps"})
                # the fraction against what a plain read kernel reaches on this machine (scripts/ubench/hbm_bw.hip,
                # profiles/r01/hbm_bw.jsonl: 6.4 TB/s) beside the 8 TB/s of the data sheet
                rb["frac_of_measured_read_bw"] = rb["achieved"] / MEASURED_READ_GBS
                rb["measured_read_bw"] = MEASURED_READ_GBS
                big_legs = {} if args.no_host_legs else host_legs(big, big.wl, n_warm + n_big + 40, iters=12, stream_iters=40)
                if not args.no_host_legs:
                    # the shipped YAML's raycasting switch on (fastdem/config/default.yaml:40-41) at configs[3]: the stage
                    # runs scan by scan behind each scan's update (scans of this size do not ride the batches)
                    rayb = Resident(big.wl, local_rank, args.wave_merge, args.overlap)
                    cfg_b = rayb.eng.cfg
                    cfg_b.raycast_enabled = 1
                    rayb.eng.set_config(cfg_b)
                    wrb, _ = rayb.batch(0, 4)
                    if rayb.eng.integrate_device_batch(wrb) != 0:
                        raise RuntimeError("integrate_device_batch (large raycasting leg warm-up) failed")
                    rayb.eng.sync()
                    brb, _ = rayb.batch(4, 12)
                    if rayb.eng.integrate_device_batch_timed(brb) != 0:
                        raise RuntimeError("integrate_device_batch (large raycasting leg) failed")
                    ray_b_us = rayb.eng.timer_ms() / 12 * 1e3
                    big_legs["raycasting_on"] = {"us_per_scan_hip_events": ray_b_us, "scans": 12,
                                                 "Mpts_per_s": big.wl.n_points / (ray_b_us * 1e-6) / 1e6,
                                                 "in_batch_launches": False}
                    del rayb
                if not args.no_host_legs:
                    # the default-radius stencil stages on the map the scans above left: wall time per
                    # call (the calls only enqueue; a sync on both sides of 20 of them), best of three
                    def stage_ms(fn):
                        fn()
                        best = float("inf")
                        for _ in range(3):
                            big.eng.sync()
                            ts = time.perf_counter()
                            for _ in range(20):
                                fn()
                            big.eng.sync()
                            best = min(best, (time.perf_counter() - ts) / 20 * 1e3)
                        return best
                    big_legs["stencils_ms_per_call"] = {
                        "uncertainty_fusion_r0.15": stage_ms(lambda: big.eng.apply_uncertainty_fusion(True, 0.15, 0.05, 0.01, 0.99, 3)),
                        "feature_extraction_r0.3": stage_ms(lambda: big.eng.apply_feature_extraction(0.3, 4, 0.05, 0.95)),
                        "cells": int(big.eng.rows * big.eng.cols)}
                result["large"] = {"workload": big.wl.name, "value": bpts / dtb / 1e6, "steps": n_big,
                                   "device_value": bpts / (big_us * n_big * 1e-6) / 1e6,
                                   "unit": "Mpts/s", "ms_per_step": dtb / n_big * 1e3,
                                   "roofline": rb, "kernels": kb,
                                   # the three times of SURVEY.md §8d: device (above), end to end from host arrays
                                   # (synchronous call), steady stream from pinned memory — PCIe-inclusive, never `value`
                                   **big_legs}
                del big
            if world == 1 and not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(wl)
    if world > 1 and not routed and not args.no_global_map:
        # LOCAL maps do not shard: `value` above is N independent replicas.  The configuration in which the GPUs of a
        # node share ONE map is BASELINE configs[4]; its line for the same N rides along, so that a 1 / 2 / 4 / 8 sweep
        # of the default command has rows for both.
        from fastdem_amd import tiling
        args.stall_timeout = args.collective_timeout + 60
        # (this leg has never run on two devices: whatever happens in it, the replicas' line above is printed)
        args.global_map_fatal = False
        args.partial_result = result if rank == 0 else None
        g = tiling.bench_global(args, rank, local_rank, world)
        if rank == 0:
            result["global_map"] = g
            result["global_map_ok"] = bool(g) and "error" not in g
    try:
        ident = rank_identity(args, rank, world, local_rank)
    except Exception as e:  # (never in the way of the line)
        ident = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 or routed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        result["ranks"] = ident
        print(json.dumps(result))


if __name__ == "__main__":
    main()
