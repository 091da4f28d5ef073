"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/fdm_engine.h declares (no compute calls — there is no GPU here), the product never
touches the oracle, and creating an engine without a device fails loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("fdm_engine.h", "fdm_engine_debug.h")):
    """Every function the engine library's headers declare: the drop-in boundary (fdm_engine.h) and the measurement /
    debugging entry points kept apart from it (fdm_engine_debug.h)."""
    out = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(fdm_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_measurement_entry_points_stay_out_of_the_boundary_header():
    public = declared_symbols(("fdm_engine.h",))
    assert not [s for s in public if "debug" in s or s.endswith("_timed")], public
    assert "fdm_engine_debug_timeline" in declared_symbols(("fdm_engine_debug.h",))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("fdm_engine_create", "fdm_engine_integrate", "fdm_engine_integrate_device",
                 "fdm_engine_update", "fdm_engine_move", "fdm_engine_layer_download",
                 "fdm_engine_clear", "fdm_engine_get_geometry", "fdm_engine_sync"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from fastdem_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_prototypes_cover_the_header():
    from fastdem_amd import capi
    assert sorted(capi.PROTOTYPES) == declared_symbols()
    capi.load()  # resolves every prototype


def test_config_struct_layout_and_defaults():
    from fastdem_amd import capi
    cfg = capi.default_config()
    assert ctypes.sizeof(capi.FdmConfig) == 4 * 4 + 4 + 4 * 7 + 4 * 2 + 4 * 3 + 4 * 5 + 4 + 4 + 4 * 6
    # Config{} defaults (config/fastdem.hpp:23-28, mapping.hpp:24-48, sensor_model.hpp:19-37)
    assert cfg.mode == 0 and cfg.estimation_type == 0 and cfg.sensor_type == 1
    assert cfg.range_min == 0.0 and cfg.z_max > 3e38 and cfg.z_min < -3e38
    assert abs(cfg.kalman_min_variance - 1e-4) < 1e-9 and abs(cfg.kalman_max_variance - 1e-2) < 1e-9
    assert [round(v, 2) for v in cfg.p2_dn] == [0.01, 0.16, 0.5, 0.84, 0.99]
    assert cfg.p2_elevation_marker == 3


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from fastdem_amd import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(10.0, 10.0, 0.5)


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "fastdem_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inl", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"fdm_ref|fdmref|oracle/", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_generated_sorting_network_is_current():
    """fastdem_amd/csrc/fdm_sortnet32.inc is the verbatim output of scripts/gen_sortnet.py (which also
    verifies the network on sampled 0-1 inputs and permutations)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "scripts", "gen_sortnet.py"), "--check"]).returncode == 0
