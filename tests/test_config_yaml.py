"""YAML configuration loading: the reference's loader tests
(fastdem/tests/test_config.cpp:36-223,335-344) re-expressed against fastdem_amd.config, plus a
cross-check that the C++ mirror's YAML-subset loader reads the shipped file to the same values."""
import os
import subprocess
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cfgmod():
    from fastdem_amd import config
    return config


def write(tmp_path, text, name="c.yaml"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_load_default_yaml(cfgmod):  # :36-43
    c = cfgmod.load_config()
    assert c.estimation_type == 0 and c.sensor_type == 1 and c.raycast_enabled == 1
    assert (c.z_min, c.z_max, c.range_min, c.range_max) == (-1.0, 2.0, 0.5, 20.0)
    assert abs(c.rc_clear_threshold + 1.0) < 1e-9 and abs(c.p2_dn[3] - 0.84) < 1e-6


def test_nonexistent_file_raises(cfgmod):  # :45-47
    with pytest.raises(RuntimeError):
        cfgmod.load_config("/nonexistent/path.yaml")


def test_empty_and_partial_yaml_keep_defaults(cfgmod, tmp_path):  # :49-74, :109-113, :149-158
    from fastdem_amd import capi
    d = capi.default_config()
    c = cfgmod.load_config(write(tmp_path, "# empty config\n"))
    assert (c.mode, c.estimation_type, c.sensor_type) == (d.mode, d.estimation_type, d.sensor_type)
    assert c.z_min == d.z_min and c.range_max == d.range_max and c.raycast_enabled == 0
    c = cfgmod.load_config(write(tmp_path, "mapping:\n  type: p2_quantile\n"))
    assert c.estimation_type == 1 and c.mode == d.mode and c.lidar_range_noise == d.lidar_range_noise


def test_enum_values(cfgmod, tmp_path):  # :76-107
    assert cfgmod.load_config(write(tmp_path, "mapping:\n  type: kalman_filter\n")).estimation_type == 0
    for name, code in (("lidar", 1), ("rgbd", 2), ("constant", 0), ("laser", 1), ("none", 0)):
        assert cfgmod.load_config(write(tmp_path, f"sensor_model:\n  type: {name}\n")).sensor_type == code
    assert cfgmod.load_config(write(tmp_path, "mapping:\n  mode: global\n")).mode == 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert cfgmod.load_config(write(tmp_path, "sensor_model:\n  type: bogus\n")).sensor_type == 1
        assert any("Unknown sensor_model.type" in str(x.message) for x in w)


def test_numeric_blocks(cfgmod, tmp_path):  # :115-147
    c = cfgmod.load_config(write(tmp_path, "mapping:\n  kalman:\n    min_variance: 0.001\n"
                                           "    max_variance: 0.05\n    process_noise: 0.001\n"))
    assert abs(c.kalman_min_variance - 0.001) < 1e-9 and abs(c.kalman_max_variance - 0.05) < 1e-8
    c = cfgmod.load_config(write(tmp_path, "point_filter:\n  z_min: -0.5\n  z_max: 2.0\n  range_min: 0.5\n"
                                           "  range_max: 20.0\n"))
    assert (c.z_min, c.z_max, c.range_min, c.range_max) == (-0.5, 2.0, 0.5, 20.0)


def test_validation(cfgmod, tmp_path):  # :160-223, :335-344
    with pytest.raises(ValueError):
        cfgmod.load_config(write(tmp_path, "mapping:\n  kalman:\n    min_variance: 0.1\n    max_variance: 0.01\n"))
    with pytest.raises(ValueError):
        cfgmod.load_config(write(tmp_path, "mapping:\n  p2:\n    dn0: 0.9\n    dn1: 0.1\n"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c = cfgmod.load_config(write(tmp_path, "sensor_model:\n  lidar:\n    range_noise: -1.0\n    angular_noise: -0.5\n"
                                               "  constant:\n    uncertainty: -1\nmapping:\n  kalman:\n    process_noise: -2\n"
                                               "  p2:\n    elevation_marker: 9\n"))
    assert abs(c.lidar_range_noise - 0.02) < 1e-9 and c.lidar_angular_noise == 0.0
    assert abs(c.constant_uncertainty - 0.1) < 1e-8 and c.kalman_process_noise == 0.0 and c.p2_elevation_marker == 4
    with pytest.raises(RuntimeError):
        cfgmod.load_config(write(tmp_path, "point_filter:\n  z_min: not_a_number\n"))


def test_cpp_loader_host_tests():
    """The C++ mirror's loadConfig over its YAML-subset parser (no GPU needed)."""
    binp = os.path.join(ROOT, "fastdem_amd", "cpp", "build", "fdm_cpp_tests")
    if not os.path.exists(binp):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastdem_amd", "cpp")])
    env = dict(os.environ, FDM_CONFIG_DIR=os.path.join(ROOT, "fastdem_amd", "config"))
    r = subprocess.run([binp, "ConfigLoad."], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout + r.stderr
