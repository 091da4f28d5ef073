"""The held-back update: a stream of small scans through fdm_engine_integrate_device leaves as ONE
launch per scan (update of scan t fused with the bin of scan t+1, k_update_bin).  Everything the
reference decides per scan (rolling move, "all filtered" / "nothing landed" gates, lazy layers) must
come out exactly as with one scan at a time, whatever is interleaved with the chain."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import assert_layers_equal, pair, same_geometry

pytestmark = pytest.mark.gpu
F32 = np.float32


def dev(s):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)).cuda() if v is not None else None) for k, v in s.items()}


def enqueue(eng, d, Tbs, Twb):
    eng.integrate_device(d["x"], d["y"], d["z"], Tbs, Twb, intensity=d.get("intensity"), rgb=d.get("rgb"))


def ref_step(ref, s, Tbs, Twb):
    kw = {k: s[k] for k in ("intensity", "rgb") if s.get(k) is not None}
    return ref.integrate(s["x"], s["y"], s["z"], Tbs, Twb, **kw)


@pytest.mark.parametrize("lean", [0, 1])
@pytest.mark.parametrize("overlap", [1, 0])
@pytest.mark.parametrize("name", ["vlp16", "rgbd_small"])
def test_chain_equals_oracle(gpu, R, name, overlap, lean):
    """lean = 1: nothing optional is asked of the bin kernel (no cell ids), which selects the LEAN fused
    kernels — the ones bench.py times."""
    wl = gpu.synth.vlp16(n_scans=12) if name == "vlp16" else gpu.synth.rgbd(n_scans=6)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.set_option("overlap", overlap)
    if lean:
        eng.enable_cell_ids(False)
    keep = []
    for k in range(12):
        s = wl.scan(k)
        if name != "vlp16":  # a 40 K-point subset keeps the scan on the one-point-per-thread bin kernel
            sel = slice(0, 40000)
            s = {c: (v[sel] if v is not None else None) for c, v in s.items()}
        d = dev(s)
        keep.append(d)
        enqueue(eng, d, wl.T_base_sensor, wl.pose(k))
        rc_r, st_r = ref_step(ref, s, wl.T_base_sensor, wl.pose(k))
    rc, st = eng.last_stats()
    assert rc == rc_r and st == st_r
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())
    assert eng.geometry().start_row != 0


@pytest.mark.parametrize("name", ["vlp16", "lidar128"])
def test_chain_on_a_stamp_gated_map(gpu, R, name):
    """Non-dense (stamp-gated) maps hold the update back too: the fused launch sweeps the tiles stamped
    by scan t, by scan t+1 (the bin kernel sharing the launch may already have re-stamped them) or by the
    last updating scan.  Moving window, scans that land nowhere and re-landing in between."""
    wl = gpu.synth.vlp16(n_scans=10) if name == "vlp16" else gpu.synth.lidar128(n_scans=5, n_az=1024)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.set_option("dense", 0)
    if name == "lidar128":
        eng.enable_cell_ids(False)  # LEAN stamped k_update_bin4
    keep = []
    n = len(wl.scans)
    for k in range(n):
        s = wl.scan(k)
        T = wl.pose(k).copy()
        if k in (3, 4):  # two scans far away: nothing lands, the obstacle cells of scan 2 must survive them
            T[0, 3] += 500.0
        d = dev(s)
        keep.append(d)
        enqueue(eng, d, wl.T_base_sensor, T)
        ref_step(ref, s, wl.T_base_sensor, T)
        if k == n // 2:
            assert_layers_equal(eng, ref)
    rc, st = eng.last_stats()
    assert rc == 0
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())


def test_chain_with_interleaved_calls(gpu, R):
    """Reads, writes, config changes and layer additions in the middle of a chain flush the
    held-back update first; the chain then restarts from the committed geometry."""
    wl = gpu.synth.vlp16(n_scans=10)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    rng = np.random.default_rng(5)
    keep = []
    for k in range(10):
        s = wl.scan(k)
        if k in (3, 4):  # drop the intensity channel for two scans (other kernels' variant), bring it back
            s = dict(s, intensity=None)
        d = dev(s)
        keep.append(d)
        enqueue(eng, d, wl.T_base_sensor, wl.pose(k))
        ref_step(ref, s, wl.T_base_sensor, wl.pose(k))
        if k == 1:
            assert_layers_equal(eng, ref)                      # downloads mid-chain
        if k == 2:
            for o in (eng, ref):
                o.add("user", 0.25)
        if k == 5:
            tag = rng.normal(size=(eng.rows, eng.cols)).astype(F32)
            for o in (eng, ref):
                o.set_layer("elevation_max", tag)
        if k == 6:
            for o in (eng, ref):
                o.move(wl.pose(k)[0, 3] + 0.7, wl.pose(k)[1, 3] - 0.4)
        if k == 7:
            ce, cr = eng.cfg, ref.cfg
            ce.kalman_process_noise = cr.kalman_process_noise = 1e-4
            eng.set_config(ce)
            ref.set_config(cr)
        if k == 8:
            for o in (eng, ref):
                o.clear("variance")
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())


def test_chain_through_the_reference_gates(gpu, R):
    """Scans that are entirely cropped (no move, fastdem.cpp:138), scans that land outside the map
    (move but no update, elevation_mapping.cpp:118) and empty clouds inside a chain."""
    wl = gpu.synth.vlp16(n_scans=1)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    s = wl.scan(0)
    high = dict(s, z=s["z"] + F32(50.0))        # every point fails cropZ
    far = dict(s, x=s["x"] * F32(0.05), y=s["y"] * F32(0.05))  # inside range_min -> all cropped too
    T = np.eye(4)
    seq = []
    for k in range(9):
        Twb = T.copy()
        Twb[0, 3], Twb[1, 3] = 0.35 * k, -0.2 * k
        cloud = (s, high, s, far, s, s, high, s, s)[k]
        seq.append((cloud, Twb))
    keep = []
    for cloud, Twb in seq:
        d = dev(cloud)
        keep.append(d)
        enqueue(eng, d, wl.T_base_sensor, Twb)
        rc_r, st_r = ref_step(ref, cloud, wl.T_base_sensor, Twb)
    empty = {k: torch.empty(0, device="cuda") for k in ("x", "y", "z")}
    eng.integrate_device(empty["x"], empty["y"], empty["z"], wl.T_base_sensor, T)  # no-op in the chain
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())
    # a last gated scan, checked through the stats path
    d = dev(high)
    enqueue(eng, d, wl.T_base_sensor, seq[-1][1])
    rc_r, st_r = ref_step(ref, high, wl.T_base_sensor, seq[-1][1])
    rc, st = eng.last_stats()
    assert rc == rc_r == 2 and st == st_r


def test_chain_mixes_small_and_large_scans(gpu, R):
    """k_bin (fusable) and k_bin4 (not) alternate: the held-back update leaves alone before a large scan."""
    small = gpu.synth.vlp16(n_scans=3)
    eng, ref = pair(gpu, R, 60.0, 60.0, 0.1, small.apply_to)
    eng.set_option("tiled", 0)  # (a 600 x 600 map would go through the per-tile pipeline by itself)
    big = gpu.synth.lidar128(n_scans=2, n_az=4096)   # 524 K points -> k_bin4 (the engine's choice from ~400 K up)
    keep = []
    for k in range(6):
        wl, idx = (small, k // 2) if k % 2 == 0 else (big, k // 2 % 2)
        s = wl.scan(idx)
        d = dev(s)
        keep.append(d)
        enqueue(eng, d, wl.T_base_sensor, small.pose(k))
        ref_step(ref, s, wl.T_base_sensor, small.pose(k))
    assert_layers_equal(eng, ref)


@pytest.mark.parametrize("first", ["tiled", "scratch"])
def test_pipeline_switch_on_a_scan_that_observes_nothing(gpu, R, first):
    """The two pipelines keep separate books on which tiles hold obstacle cells, so the first UPDATING scan behind a switch
    clears the layer as a whole (the reference's `map_.clear(obstacle)`, elevation_mapping.cpp:144-146).  If the scan AT
    the switch observes nothing — every point filtered — the reference clears nothing and the debt passes on to the next
    scan that does.  Found by a 240-s soak in round 5: 1 stale obstacle cell after 352 K scans."""
    small = gpu.synth.vlp16(n_scans=4)
    eng, ref = pair(gpu, R, 60.0, 60.0, 0.1, small.apply_to)   # 600 x 600: scans of >= 2 048 points take the record pools
    rng = np.random.default_rng(5)

    def cloud(n, lift=0.0, x0=0.0):
        x = rng.uniform(-8, 8, n).astype(F32) + F32(x0)
        y = rng.uniform(-8, 8, n).astype(F32)
        z = (rng.uniform(-0.5, 0.5, n) + lift).astype(F32)
        return {"x": x, "y": y, "z": z, "intensity": rng.uniform(0, 1, n).astype(F32), "rgb": None}

    big, little = 6000, 900  # (tiled_min is 2 048 points)
    a, b = (big, little) if first == "tiled" else (little, big)
    # pipeline A observes (obstacle cells around x = -4), the switch to B happens on a scan that is filtered away as a
    # whole (z far above z_max), then B observes somewhere else, twice
    seq = [cloud(a, x0=-4.0), cloud(a, x0=-4.0), cloud(b, lift=50.0), cloud(b, x0=4.0), cloud(b, x0=4.0)]
    Tbs = small.T_base_sensor
    keep = []
    for k, s in enumerate(seq):
        d = dev(s)
        keep.append(d)
        enqueue(eng, d, Tbs, small.pose(0))
        ref_step(ref, s, Tbs, small.pose(0))
    assert_layers_equal(eng, ref)
    assert np.isfinite(eng.layer("obstacle")).sum() > 0


def test_p2_and_per_layer_storage_in_a_chain(gpu, R):
    wl = gpu.synth.vlp16(n_scans=8)

    def fill(c):
        wl.apply_to(c)
        c.estimation_type = 1
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, fill)
    per = gpu.Engine(wl.width, wl.height, wl.resolution, eng.cfg)
    per.set_option("records", 0)
    keep = []
    for k in range(8):
        d = dev(wl.scan(k))
        keep.append(d)
        enqueue(eng, d, wl.T_base_sensor, wl.pose(k))
        enqueue(per, d, wl.T_base_sensor, wl.pose(k))
        ref_step(ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
    assert_layers_equal(eng, ref)
    assert_layers_equal(per, ref)


def test_destroy_with_a_held_back_update(gpu, R):
    wl = gpu.synth.vlp16(n_scans=2)
    eng = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    d = dev(wl.scan(0))
    enqueue(eng, d, wl.T_base_sensor, wl.pose(0))
    eng.close()  # nothing read back: the held-back closure is dropped with the engine


@pytest.mark.parametrize("lean", [0, 1])
@pytest.mark.parametrize("n_az,scans,variant", [(2048, 6, 0), (2048, 6, 4), (16384, 3, 0)])
def test_chain_of_large_scans(gpu, R, n_az, scans, lean, variant):
    """Bin launches of large scans (k_bin by the engine's choice at 262 K points, k_bin4 forced; the per-tile
    pipeline at 2 M) carrying the previous scan's update, with an 8-cell rolling shift per scan."""
    wl = gpu.synth.lidar128(n_scans=scans, n_az=n_az)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.set_option("bin_variant", variant)
    if lean:
        eng.enable_cell_ids(False)
    keep = []
    for k in range(scans):
        d = dev(wl.scan(k))
        keep.append(d)
        enqueue(eng, d, wl.T_base_sensor, wl.pose(k))
        rc_r, st_r = ref_step(ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
    rc, st = eng.last_stats()
    assert rc == rc_r and st == st_r and st["shift_rows"] == -8
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())


def test_long_chain_of_large_scans_through_the_batch_entry(gpu, R):
    """48 scans of 262 K points in ONE fdm_engine_integrate_device_batch call (what bench.py times): the record
    pools alternate 24 times, the window rolls by 8 cells per scan until it has wrapped around, every update is
    held back into the next scan's launch — compared with the oracle at the end, bit for bit."""
    scans, distinct = 48, 6
    wl = gpu.synth.lidar128(n_scans=distinct, n_az=2048)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.enable_cell_ids(False)
    devs = [dev(wl.scan(k)) for k in range(distinct)]
    arr = (gpu.capi.FdmDeviceScan * scans)()
    for k in range(scans):
        d, a = devs[k % distinct], arr[k]
        a.n = wl.n_points
        a.x, a.y, a.z, a.intensity = (d[c].data_ptr() for c in ("x", "y", "z", "intensity"))
        a.rgb, a.sigma_z2 = None, None
        # (the ABI takes 4x4 matrices column-major, as Eigen stores them)
        a.T_base_sensor = (C.c_double * 16)(*np.asarray(wl.T_base_sensor, dtype=np.float64).T.ravel())
        a.T_world_base = (C.c_double * 16)(*np.asarray(wl.pose(k), dtype=np.float64).T.ravel())
    assert eng.integrate_device_batch(arr) == 0
    for k in range(scans):
        rc_r, st_r = ref_step(ref, wl.scan(k % distinct), wl.T_base_sensor, wl.pose(k))
    rc, st = eng.last_stats()
    assert rc == rc_r and st == st_r
    assert_layers_equal(eng, ref, rtol=0.0)
    assert same_geometry(eng.geometry(), ref.geometry())


@pytest.mark.parametrize("variant", [0, 4])
@pytest.mark.parametrize("lean", [0, 1])
def test_chain_rgbd_p2_colour_large(gpu, R, lean, variant):
    """configs[2] through the chain: P2 cell records (128 B), colour channel; k_bin (the engine's choice at
    272 K points) and k_bin4<false,true,256> (forced)."""
    wl = gpu.synth.rgbd(n_scans=5)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.set_option("bin_variant", variant)
    if lean:
        eng.enable_cell_ids(False)
    keep = []
    for k in range(5):
        d = dev(wl.scan(k))
        keep.append(d)
        enqueue(eng, d, wl.T_base_sensor, wl.pose(k))
        ref_step(ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
    assert_layers_equal(eng, ref)
    assert "color" in eng.layers()


@pytest.mark.parametrize("zero_copy", [0, 1 << 20])
def test_host_streaming_entry_point(gpu, R, zero_copy):
    """fdm_engine_integrate_async on PINNED host arrays, enqueue only.  zero_copy = 0: H2D copies into
    three rotating staging blocks + the scan; default: the bin kernel reads the pinned arrays in place
    and writes them through to the staging block the held-back update gathers from."""
    wl = gpu.synth.vlp16(n_scans=9)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.set_option("zero_copy", zero_copy)
    pinned = []
    for k in range(9):
        s = wl.scan(k)
        h = {c: torch.from_numpy(np.ascontiguousarray(s[c])).pin_memory() for c in ("x", "y", "z", "intensity")}
        pinned.append(h)  # arrays stay untouched until the final sync
        eng.integrate_async(h["x"].numpy(), h["y"].numpy(), h["z"].numpy(), wl.T_base_sensor, wl.pose(k),
                            intensity=h["intensity"].numpy())
        ref_step(ref, s, wl.T_base_sensor, wl.pose(k))
    rc, st = eng.last_stats()
    assert rc == 0 and st["n_input"] == 28800
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())


def test_host_streaming_pageable_arrays_are_staged(gpu, R):
    """Plain (pageable) numpy arrays are not device-visible: the same entry point falls back to copies."""
    wl = gpu.synth.vlp16(n_scans=4)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    for k in range(4):
        s = wl.scan(k)
        eng.integrate_async(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
        ref_step(ref, s, wl.T_base_sensor, wl.pose(k))
    assert_layers_equal(eng, ref)


@pytest.mark.parametrize("variant", [0, 4])
def test_host_streaming_rgbd_in_place(gpu, R, variant):
    """configs[2] from pinned memory: the bin kernel (k_bin by choice, k_bin4 forced) reads x/y/z in place, the
    colour channel (which only the update kernel consumes) is copied through by the bin kernel as well."""
    wl = gpu.synth.rgbd(n_scans=4)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.set_option("bin_variant", variant)
    pinned = []
    for k in range(4):
        s = wl.scan(k)
        h = {c: torch.from_numpy(np.ascontiguousarray(s[c])).pin_memory() for c in ("x", "y", "z", "rgb")}
        pinned.append(h)
        eng.integrate_async(h["x"].numpy(), h["y"].numpy(), h["z"].numpy(), wl.T_base_sensor, wl.pose(k),
                            rgb=h["rgb"].numpy())
        ref_step(ref, s, wl.T_base_sensor, wl.pose(k))
    assert_layers_equal(eng, ref)
    assert "color" in eng.layers()


def test_host_streaming_in_place_with_raycasting(gpu, R):
    """The in-place path with the ray stage on (not a plain scan: the update is not held back)."""
    wl = gpu.synth.vlp16(n_scans=4)

    def cfg(c):
        c = wl.apply_to(c)
        c.raycast_enabled = 1
        return c
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, cfg)
    pinned = []
    for k in range(4):
        s = wl.scan(k)
        h = {c: torch.from_numpy(np.ascontiguousarray(s[c])).pin_memory() for c in ("x", "y", "z", "intensity")}
        pinned.append(h)
        eng.integrate_async(h["x"].numpy(), h["y"].numpy(), h["z"].numpy(), wl.T_base_sensor, wl.pose(k),
                            intensity=h["intensity"].numpy())
        ref_step(ref, s, wl.T_base_sensor, wl.pose(k))
    assert_layers_equal(eng, ref)


def test_update_device_chain(gpu, R):
    """ElevationMapping::update (cloud already in the map frame, tests/test_dual_layer.cpp:71) through
    the enqueue-only entry point: LOCAL moves by the robot position, no crops, z variance channel."""
    rng = np.random.default_rng(8)
    eng, ref = pair(gpu, R, 12.0, 12.0, 0.1)
    keep = []
    for k in range(7):
        n = 6000
        rx, ry = 0.37 * k, -0.21 * k
        x = (rng.uniform(-5, 5, n) + rx).astype(F32)
        y = (rng.uniform(-5, 5, n) + ry).astype(F32)
        z = rng.normal(0.2, 0.3, n).astype(F32)
        var = rng.uniform(1e-4, 5e-3, n).astype(F32) if k % 2 == 0 else None
        d = {c: torch.from_numpy(v).cuda() for c, v in (("x", x), ("y", y), ("z", z))}
        dv = torch.from_numpy(var).cuda() if var is not None else None
        keep.append((d, dv))
        eng.update_device(d["x"], d["y"], d["z"], (rx, ry), z_var=dv)
        st_r = ref.update(x, y, z, (rx, ry), z_var=var)
    rc, st = eng.last_stats()
    assert st == st_r
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())


def test_fused_launch_timeline_tool(gpu):
    """fdm_engine_debug_timeline (option dbg_timeline): start / end ticks of every block of the last fused
    large-scan launch — update blocks first, bin blocks behind them, every block ends after it starts."""
    import torch
    wl = gpu.synth.lidar128(n_scans=3, n_az=2048)
    eng = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    with pytest.raises(gpu.EngineError):
        eng.debug_timeline()  # the option is off
    eng.set_option("dbg_timeline", 1)
    keep = []
    for k in range(3):
        s = wl.scan(k)
        d = {c: torch.from_numpy(np.ascontiguousarray(s[c])).cuda() for c in ("x", "y", "z", "intensity")}
        keep.append(d)
        eng.integrate_device(d["x"], d["y"], d["z"], wl.T_base_sensor, wl.pose(k), intensity=d["intensity"])
    assert eng.last_pipeline() == 1
    ticks, n_upd = eng.debug_timeline()
    # update blocks (four tile wavefronts each, 16 x 16-cell tiles; at most the option's default of 768) + bin blocks
    tiles = ((eng.rows + 15) // 16) * ((eng.cols + 15) // 16)
    assert n_upd == min((tiles + 3) // 4, 768) and len(ticks) == n_upd + (wl.n_points + 1023) // 1024
    assert (ticks[:, 1] >= ticks[:, 0]).all() and ticks[:, 0].min() > 0
    span_us = (int(ticks[:, 1].max()) - int(ticks[:, 0].min())) / 100.0
    assert 1.0 < span_us < 5000.0
    eng.set_option("dbg_timeline", 0)
    eng.sync()


def test_engine_stopwatch(gpu):
    """fdm_engine_timer_*: the device-side duration of a run of enqueue-only calls, the last scan's held-back
    update included (what bench.py reports as `device_value`)."""
    wl = gpu.synth.vlp16(n_scans=3)
    eng = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    devs = [dev(wl.scan(k)) for k in range(3)]
    for k in range(3):
        enqueue(eng, devs[k], wl.T_base_sensor, wl.pose(k))
    eng.timer_start()
    for k in range(3, 43):
        enqueue(eng, devs[k % 3], wl.T_base_sensor, wl.pose(k))
    eng.timer_stop()
    ms = eng.timer_ms()
    assert 40 * 0.002 < ms < 40 * 0.2, ms  # 40 scans of a few microseconds each
    rc, st = eng.last_stats()
    assert rc == 0 and st["n_in_map"] > 0
