"""Map egress of the HIP engine through the C ABI vs the oracle's restatement of
toPointCloud2Impl (bridge/ros/impl.hpp:28-166): field list, point count and every byte of the records,
in the reference's visiting order."""
import ctypes as C

import numpy as np
import pytest

from helpers import pair, run_both

pytestmark = pytest.mark.gpu
F32 = np.float32


def same_cloud(eng, ref, layer="elevation", sub=None):
    fe, se, de = eng.pack_cloud(layer, sub)
    fr, sr, dr = ref.pack_cloud(layer, sub)
    assert fe == fr and se == sr
    assert de.shape == dr.shape
    assert np.array_equal(de.view(np.uint32), dr.view(np.uint32)), \
        f"{(de.view(np.uint32) != dr.view(np.uint32)).sum()} words differ"
    return fe, de


def test_small_hand_map(gpu, R):
    eng, ref = pair(gpu, R, 2.0, 1.5, 0.5)
    el = np.full((4, 3), np.nan, dtype=F32)
    el[0, 0], el[1, 2], el[3, 1], el[2, 2] = 1.0, 3.0, -2.0, np.inf
    for o in (eng, ref):
        o.set_layer("elevation", el)
    f, d = same_cloud(eng, ref)
    assert d.shape[0] == 3 and np.array_equal(d[0, :3], np.array([0.75, 0.5, 1.0], dtype=F32))
    same_cloud(eng, ref, sub=(1, 1, 3, 2))
    same_cloud(eng, ref, "elevation_max")  # empty cloud


@pytest.mark.parametrize("name", ["vlp16", "rgbd"])
def test_after_scans_with_rolling_window(gpu, R, name):
    wl = getattr(gpu.synth, name)(n_scans=6)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    for k in range(6):
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
    f, d = same_cloud(eng, ref)
    assert d.shape[0] > 1000
    if name == "rgbd":
        assert f[-1] == "rgb"
    else:
        assert "intensity" in f and all(not n.startswith("_") for n in f)
    g = eng.geometry()
    assert (g.start_row, g.start_col) != (0, 0)
    same_cloud(eng, ref, sub=(g.rows - 5, g.cols - 7, 40, 33))   # wraps around both axes
    same_cloud(eng, ref, "elevation_min")


def test_with_raycasting_layers_and_creation_order(gpu, R):
    """getLayers() order == creation order in the reference: ghost_removal/raycasting appear after
    intensity when both are born in the same scan, and a later user layer comes last."""
    wl = gpu.synth.vlp16(n_scans=3)

    def fill(cfg):
        wl.apply_to(cfg)
        cfg.raycast_enabled = 1
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, fill)
    for k in range(3):
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
    for o in (eng, ref):
        o.add("user_layer", 0.5)
    assert eng.layers() == ref.layers()
    f, _ = same_cloud(eng, ref)
    assert f.index("intensity") < f.index("ghost_removal") < f.index("raycasting") < f.index("user_layer")


def test_intensity_born_after_raycasting(gpu, R):
    """First scans land outside the map (no intensity layer yet) while raycasting already runs;
    the intensity layer is created later -> it must come AFTER the raycasting layers."""
    wl = gpu.synth.vlp16(n_scans=1)

    def fill(cfg):
        wl.apply_to(cfg)
        cfg.mode = 1  # GLOBAL
        cfg.raycast_enabled = 1
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, fill)
    s = wl.scan(0)
    far = np.eye(4)
    far[2, 3] = 30.0  # base 30 m up: every return lies above z_max... keep x,y so the sensor is inside
    s_out = {"x": s["x"] + 100.0, "y": s["y"], "z": s["z"], "intensity": s["intensity"], "rgb": None}

    def wide(o):
        c = o.cfg
        c.range_max = 1e9
        o.set_config(c)
    for o in (eng, ref):
        wide(o)
    run_both(eng, ref, s_out, wl.T_base_sensor, np.eye(4))     # all points outside the map; rays run
    assert eng.exists("raycasting") and not eng.exists("intensity")
    run_both(eng, ref, s, wl.T_base_sensor, np.eye(4))
    assert eng.layers() == ref.layers()
    assert eng.layers().index("raycasting") < eng.layers().index("intensity")
    same_cloud(eng, ref)


def test_device_resident_records(gpu, R):
    import torch
    wl = gpu.synth.vlp16(n_scans=2)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    for k in range(2):
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
    d_ptr, n, step = eng.pack_cloud_device()
    fr, sr, dr = ref.pack_cloud()
    assert n == dr.shape[0] and step == sr
    host = np.empty((n, step // 4), dtype=F32)
    torch.cuda.synchronize()
    import fastdem_amd.capi as capi  # noqa: F401
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(d_ptr), host.nbytes, 2) == 0
    assert np.array_equal(host.view(np.uint32), dr.view(np.uint32))


def test_large_map_block_scan(gpu, R):
    """1200x1200 cells = 5625 count blocks: the scan kernel carries across 1024-entry chunks."""
    rng = np.random.default_rng(5)
    eng, ref = pair(gpu, R, 60.0, 60.0, 0.05)
    el = rng.normal(0, 1, (eng.rows, eng.cols)).astype(F32)
    el[rng.uniform(size=el.shape) < 0.6] = np.nan
    for o in (eng, ref):
        o.set_layer("elevation", el)
        o.move(3.35, -7.9)
    _, d = same_cloud(eng, ref)
    assert d.shape[0] > 100_000
