"""The C++17 host mirror of fastdem::FastDEM / ElevationMap (fastdem_amd/cpp) — the reference's
API-level gtests re-expressed in fastdem_amd/cpp/tests/test_fastdem_api.cpp and run as a binary."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "fastdem_amd", "cpp", "build", "fdm_cpp_tests")


def test_cpp_spec_tests_are_built():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastdem_amd", "cpp")])
    assert os.access(BIN, os.X_OK)


def test_cpp_api_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    # host-only tests (config validation, sensor-model classes) still pass ...
    r = subprocess.run([BIN, "Config."], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    # ... anything that needs the map raises: there is no CPU fallback behind the API
    r = subprocess.run([BIN, "ElevationMap.Default"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no HIP device" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_cpp_api_spec_tests_on_gpu():
    env = dict(os.environ, FDM_CONFIG_DIR=os.path.join(ROOT, "fastdem_amd", "config"))
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " 0 failures" in r.stdout
    # the .npz checkpoint the C++ side wrote (fastdem/io/npz.hpp) is a plain NumPy archive
    import tempfile
    import numpy as np
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdm_cpp_checkpoint.npz")
    assert os.path.exists(path), tempfile.gettempdir()
    z = np.load(path)
    meta = __import__("json").loads(bytes(z["meta"]).decode())
    assert meta["version"] == 1 and meta["size"] == [20, 20] and meta["frame_id"] == ""
    assert abs(meta["resolution"] - 0.5) < 1e-6 and meta["start_index"] != [0, 0]
    elev = z["elevation"]
    assert elev.dtype == np.float32 and elev.shape == (20, 20) and elev.flags.f_contiguous
    assert np.isfinite(elev).sum() == 16 and np.allclose(elev[np.isfinite(elev)], 1.25)
    assert {"variance", "n_points", "_kalman_p", "obstacle"} <= set(z.files)
    os.remove(path)


@pytest.mark.gpu
def test_cpp_multi_gpu_loops_with_a_one_rank_rccl_communicator():
    """INTEGRATION.md §D compiled and run: a C++ host (HIP + RCCL) driving libfdm_halo.so — the replicated-scan loop
    (ncclBroadcast + integrate + halo exchange) and the routed-scan loop (route + ncclAllGather + point exchange +
    integrate), each against a plain engine, bit for bit."""
    exe = os.path.join(ROOT, "fastdem_amd", "cpp", "build", "fdm_halo_loop")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastdem_amd", "cpp"), "halo_loop"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, cwd=ROOT)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "halo loop: ok" in r.stdout


def test_conformance_expected_is_current():
    """scripts/conformance/expected.json + vectors.inc are what make_expected.py generates from the oracle today."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts", "conformance"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_expected as M
    exp = json.load(open(os.path.join(ROOT, "scripts", "conformance", "expected.json")))
    assert exp["index"] == M.expect_index() and exp["colors"] == M.expect_colors() and exp["regions"] == M.expect_regions()
    assert exp["move_all"] == M.expect_move(0) and exp["move_basic"] == M.expect_move(1) and exp["move_all"] != exp["move_basic"]
    assert exp["pre"] == M.expect_pre()


@pytest.mark.gpu
@pytest.mark.parametrize("basic", [0, 1])
def test_conformance_probe_through_the_mirror(basic, tmp_path):
    """The maintainer-side probe (scripts/conformance/probe.cpp) compiled against THIS repo's C++ mirror and run on the
    engine: index arithmetic, move() — in both readings of which layers it clears —, the packed colour, the preprocessed
    cloud incl. R Sigma R^T and every layer of the map after two scans per sensor model and estimator must be what the
    oracle put into expected.json.  (What the probe prints about the REAL libraries is the maintainer's to run:
    INTEGRATION.md §C.)"""
    exe = os.path.join(ROOT, "fastdem_amd", "cpp", "build", "fdm_probe_mirror")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastdem_amd", "cpp")])
    r = subprocess.run([exe] + (["--move-clear-basic"] if basic else []), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = tmp_path / "probe.txt"
    out.write_text(r.stdout)
    c = subprocess.run(["python3", os.path.join(ROOT, "scripts", "conformance", "check.py"), str(out), "--skip", "regions"],
                       capture_output=True, text=True, timeout=120)
    print(c.stdout)
    assert c.returncode == 0, c.stdout + c.stderr
    assert ("BASIC layers only" in c.stdout) == bool(basic)
