"""Pins the ingest part of the oracle (oracle/fdm_ref_ingest.hpp) with hand-derived
values of nanopcl/bridge/ros/impl.hpp:104-118,163-171,174-246 (the ROS bridge has no unit test)."""
import numpy as np

from cloud2 import Layout, make_blob

F32 = np.float32


def test_nonfinite_points_skipped_order_kept(R):  # impl.hpp:216-219
    x = np.array([1, np.nan, 3, 4, np.inf, 6], dtype=F32)
    y = np.array([1, 2, -np.inf, 4, 5, 6], dtype=F32)
    z = np.array([0, 0, 0, np.nan, 0, 0.5], dtype=F32)
    blob, lay = make_blob(x, y, z, intensity=np.arange(6, dtype=F32), rgb=np.arange(6) * 0x010203 + 0xFF000000)
    c = R.from_cloud2(blob, 6, lay)
    assert list(c["x"]) == [1.0, 6.0] and list(c["z"]) == [0.0, 0.5]
    assert list(c["intensity"]) == [0.0, 5.0]
    assert list(c["rgb"]) == [0, 5 * 0x010203]  # alpha byte dropped (impl.hpp:163-171)


def test_intensity_types(R):  # impl.hpp:104-118
    x = np.zeros(3, dtype=F32)
    for code, vals, want in [(2, [0, 7, 255], [0.0, 7.0, 255.0]), (4, [1, 300, 65535], [1.0, 300.0, 65535.0]),
                             (7, [0.5, -1.25, 3e3], [0.5, -1.25, 3e3]), (8, [0.1, 1e-3, 12345.678], None),
                             (5, [1, 2, 3], [0.0, 0.0, 0.0])]:  # INT32 is not a readIntensity case -> 0
        blob, lay = make_blob(x, x, x, intensity=np.array(vals), intensity_type=code)
        got = R.from_cloud2(blob, 3, lay)["intensity"]
        want = np.array(vals, dtype=np.float64).astype(F32) if want is None else np.array(want, dtype=F32)
        assert np.array_equal(got, want), (code, got)


def test_missing_xyz_or_empty_gives_empty_cloud(R):  # impl.hpp:178-186
    x = np.ones(4, dtype=F32)
    blob, lay = make_blob(x, x, x)
    assert R.from_cloud2(blob, 0, lay)["x"].size == 0
    lay.off_z = -1
    assert R.from_cloud2(blob, 4, lay)["x"].size == 0


def test_unaligned_records(R):
    rng = np.random.default_rng(1)
    x, y, z = (rng.normal(size=50).astype(F32) for _ in range(3))
    a = rng.random(50).astype(F32)
    blob, lay = make_blob(x, y, z, intensity=a, offsets=dict(x=1, y=6, z=11, intensity=17), point_step=23, lead=3)
    c = R.from_cloud2(blob, 50, lay)
    assert np.array_equal(c["x"], x) and np.array_equal(c["z"], z) and np.array_equal(c["intensity"], a)


def test_integrate_cloud2_equals_integrate_of_decoded(R):
    rng = np.random.default_rng(2)
    n = 2000
    x, y = (rng.uniform(-4, 4, n).astype(F32) for _ in range(2))
    z = rng.normal(0, 0.2, n).astype(F32)
    x[::17] = np.nan
    a = rng.random(n).astype(F32)
    blob, lay = make_blob(x, y, z, intensity=a, point_step=32)
    T = np.eye(4)
    e1 = R.RefEngine(10.0, 10.0, 0.1)
    e2 = R.RefEngine(10.0, 10.0, 0.1)
    rc1, st1 = e1.integrate_cloud2(blob, n, lay, T, T)
    keep = np.isfinite(x)
    rc2, st2 = e2.integrate(x[keep], y[keep], z[keep], T, T, intensity=a[keep])
    assert rc1 == rc2 == 0 and st1 == st2 and st1["n_input"] == keep.sum()
    for name in e1.layers():
        assert np.array_equal(e1.layer(name), e2.layer(name), equal_nan=True), name
