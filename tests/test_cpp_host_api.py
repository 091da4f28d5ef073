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
