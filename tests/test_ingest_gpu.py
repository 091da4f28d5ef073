"""PointCloud2 ingest of the HIP engine through the C ABI vs the oracle's
restatement of nanopcl from_impl: decoded channels bit-exact and in message order; ingest + integrate
in one call equal to the oracle doing the same."""
import numpy as np
import pytest

from cloud2 import make_blob
from helpers import assert_layers_equal, pair, same_geometry

pytestmark = pytest.mark.gpu
F32 = np.float32


def lay_of(gpu, lay):
    return gpu.Engine.cloud2_layout(lay.point_step, lay.off_x, lay.off_y, lay.off_z, lay.off_intensity,
                                    lay.intensity_type, lay.off_rgb)


def same_decode(gpu, R, blob, n, lay):
    eng = gpu.Engine(4.0, 4.0, 0.5)
    got = eng.ingest_cloud2(blob, n, lay_of(gpu, lay))
    want = R.from_cloud2(blob, n, lay)
    for k in ("x", "y", "z", "intensity", "rgb"):
        if want[k] is None:
            assert got[k] is None, k
        else:
            assert got[k] is not None and np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    return got


def noisy_cloud(rng, n):
    x, y = (rng.uniform(-6, 6, n).astype(F32) for _ in range(2))
    z = rng.normal(0, 0.3, n).astype(F32)
    bad = rng.uniform(size=n)
    x[bad < 0.05] = np.nan
    y[(bad > 0.05) & (bad < 0.08)] = np.inf
    z[(bad > 0.08) & (bad < 0.10)] = -np.inf
    return x, y, z


@pytest.mark.parametrize("itype", [2, 4, 7, 8, 5])
def test_decode_intensity_types(gpu, R, itype):
    rng = np.random.default_rng(itype)
    n = 10_000
    x, y, z = noisy_cloud(rng, n)
    a = rng.integers(0, 250, n) if itype in (2, 4, 5) else rng.random(n) * 100
    blob, lay = make_blob(x, y, z, intensity=a, intensity_type=itype, rng=rng)
    got = same_decode(gpu, R, blob, n, lay)
    assert 0 < got["x"].size < n


def test_decode_rgb_and_padding_like_a_velodyne_message(gpu, R):
    rng = np.random.default_rng(9)
    n = 70_000
    x, y, z = noisy_cloud(rng, n)
    rgb = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    blob, lay = make_blob(x, y, z, intensity=rng.random(n), rgb=rgb,
                          offsets=dict(x=0, y=4, z=8, intensity=16, rgb=24), point_step=32, rng=rng)
    got = same_decode(gpu, R, blob, n, lay)
    assert (got["rgb"] >> 24).max() == 0


def test_decode_unaligned_records(gpu, R):
    rng = np.random.default_rng(10)
    n = 5000
    x, y, z = noisy_cloud(rng, n)
    for itype in (7, 8, 4):
        blob, lay = make_blob(x, y, z, intensity=rng.random(n) * 9, intensity_type=itype, rgb=rng.integers(0, 2 ** 24, n),
                              offsets=dict(x=1, y=6, z=11, intensity=17, rgb=27), point_step=33, rng=rng, lead=3)
        same_decode(gpu, R, blob, n, lay)


def test_edge_cases(gpu, R):
    x = np.ones(4, dtype=F32)
    blob, lay = make_blob(x, x, x)
    eng = gpu.Engine(4.0, 4.0, 0.5)
    assert eng.ingest_cloud2(blob, 0, lay_of(gpu, lay))["x"].size == 0           # empty message
    nan = np.full(300, np.nan, dtype=F32)
    b2, l2 = make_blob(nan, nan, nan)
    assert eng.ingest_cloud2(b2, 300, lay_of(gpu, l2))["x"].size == 0            # nothing finite
    rc, _ = eng.integrate_cloud2(b2, 300, lay_of(gpu, l2), np.eye(4), np.eye(4))
    assert rc == 1                                                               # FDM_SKIP_EMPTY_CLOUD
    lay.off_z = -1
    assert eng.ingest_cloud2(blob, 4, lay_of(gpu, lay))["x"].size == 0           # no z field
    lay.off_z = lay.point_step - 2
    with pytest.raises(gpu.EngineError):
        eng.ingest_cloud2(blob, 4, lay_of(gpu, lay))                             # field leaves the record


@pytest.mark.parametrize("name", ["vlp16", "rgbd"])
def test_integrate_cloud2_matches_oracle(gpu, R, name):
    wl = getattr(gpu.synth, name)(n_scans=4)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    rng = np.random.default_rng(3)
    for k in range(4):
        s = wl.scan(k)
        x = s["x"].copy()
        x[rng.uniform(size=x.size) < 0.02] = np.nan  # dropped returns, as real drivers publish them
        blob, lay = make_blob(x, s["y"], s["z"], intensity=s.get("intensity"), rgb=s.get("rgb"),
                              point_step=32, rng=rng)
        rc_e, st_e = eng.integrate_cloud2(blob, x.size, lay_of(gpu, lay), wl.T_base_sensor, wl.pose(k))
        rc_r, st_r = ref.integrate_cloud2(blob, x.size, lay, wl.T_base_sensor, wl.pose(k))
        assert rc_e == rc_r == 0 and st_e == st_r
        assert st_e["n_input"] == np.isfinite(x).sum()
    assert_layers_equal(eng, ref)


def test_blob_already_in_hbm(gpu, R):
    import torch
    wl = gpu.synth.vlp16(n_scans=1)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    s = wl.scan(0)
    blob, lay = make_blob(s["x"], s["y"], s["z"], intensity=s["intensity"], point_step=16)
    d = torch.from_numpy(np.ascontiguousarray(blob)).cuda()
    rc_e, st_e = eng.integrate_cloud2(None, s["x"].size, lay_of(gpu, lay), wl.T_base_sensor, wl.pose(0),
                                      on_device_ptr=d.data_ptr())
    rc_r, st_r = ref.integrate_cloud2(blob, s["x"].size, lay, wl.T_base_sensor, wl.pose(0))
    assert rc_e == rc_r == 0 and st_e == st_r
    assert_layers_equal(eng, ref)


def test_integrate_cloud2_all_points_non_finite_is_an_empty_cloud(gpu, R):
    """from_impl drops every point -> cloud.empty() -> integrate returns false before anything moves
    (fastdem.cpp:125-128); the one-pass path decides that on the device."""
    wl = gpu.synth.vlp16(n_scans=2)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    s = wl.scan(0)
    blob, lay = make_blob(s["x"], s["y"], s["z"], intensity=s["intensity"], point_step=16)
    for o, l in ((eng, lay_of(gpu, lay)), (ref, lay)):
        rc, st = o.integrate_cloud2(blob, s["x"].size, l, wl.T_base_sensor, wl.pose(0))
        assert rc == 0
    bad = np.full(5000, np.nan, dtype=F32)
    inf = np.full(5000, np.inf, dtype=F32)
    blob2, lay2 = make_blob(bad, inf, -inf, point_step=16)
    rc_e, st_e = eng.integrate_cloud2(blob2, bad.size, lay_of(gpu, lay2), wl.T_base_sensor, wl.pose(1))
    rc_r, st_r = ref.integrate_cloud2(blob2, bad.size, lay2, wl.T_base_sensor, wl.pose(1))
    assert rc_e == rc_r == gpu.capi.FDM_SKIP_EMPTY_CLOUD and st_e == st_r and st_e["n_input"] == 0
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())  # the window did not move
    assert eng.last_stats()[0] == gpu.capi.FDM_SKIP_EMPTY_CLOUD


def test_integrate_cloud2_infinite_coordinates_are_dropped_before_the_crops(gpu, R):
    """+Inf x passes cropRange with the default range_max (inf <= inf) when it reaches integrate();
    through a PointCloud2 it never does (from_impl: isfinite) and must not count as a filtered-in point."""
    eng, ref = pair(gpu, R, 8.0, 8.0, 0.1)
    rng = np.random.default_rng(5)
    n = 3000
    x, y = (rng.uniform(-3, 3, n).astype(F32) for _ in range(2))
    z = rng.normal(0, 0.1, n).astype(F32)
    x[::7] = np.inf
    y[3::11] = -np.inf
    blob, lay = make_blob(x, y, z, point_step=12)
    I = np.eye(4)
    rc_e, st_e = eng.integrate_cloud2(blob, n, lay_of(gpu, lay), I, I)
    rc_r, st_r = ref.integrate_cloud2(blob, n, lay, I, I)
    assert rc_e == rc_r == 0 and st_e == st_r
    assert st_e["n_input"] == st_e["n_after_filter"] == int((np.isfinite(x) & np.isfinite(y)).sum())
    assert_layers_equal(eng, ref)


def test_integrate_cloud2_pinned_message_is_decoded_in_place(gpu, R):
    wl = gpu.synth.rgbd(n_scans=2)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    for k in range(2):
        s = wl.scan(k)
        blob, lay = make_blob(s["x"], s["y"], s["z"], rgb=s["rgb"], point_step=32)
        msg = gpu.host_array(np.frombuffer(blob, dtype=np.uint8), dtype=np.uint8)
        assert msg.pinned
        rc_e, st_e = eng.integrate_cloud2(msg.array, s["x"].size, lay_of(gpu, lay), wl.T_base_sensor, wl.pose(k))
        rc_r, st_r = ref.integrate_cloud2(blob, s["x"].size, lay, wl.T_base_sensor, wl.pose(k))
        assert rc_e == rc_r == 0 and st_e == st_r
    assert_layers_equal(eng, ref)
