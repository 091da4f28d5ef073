#!/usr/bin/env python3
"""Cold-start cost of the engine in a fresh process: library load, engine creation, first scan (module load +
first launches), second scan.  The C ABI is driven WITHOUT importing torch (a C++ host does not have it)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import numpy as np  # noqa: E402
t_np = time.perf_counter()
os.environ.setdefault("FDM_NO_TORCH", "1")
from fastdem_amd import capi, synth  # noqa: E402
lib = C.CDLL(capi.LIB_PATH, mode=C.RTLD_GLOBAL)
for name, (res, args) in capi.PROTOTYPES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
t_lib = time.perf_counter()
wl = synth.make(sys.argv[1] if len(sys.argv) > 1 else "c2", n_scans=2)
t_syn = time.perf_counter()
cfg = capi.FdmConfig()
lib.fdm_default_config(C.byref(cfg))
wl.apply_to(cfg)
geo = capi.FdmGeometry()
geo.length_x, geo.length_y, geo.resolution = float(np.float32(wl.width)), float(np.float32(wl.height)), float(np.float32(wl.resolution))
h = C.c_void_p()
t1 = time.perf_counter()
rc = lib.fdm_engine_create(C.byref(geo), C.byref(cfg), None, 0, C.byref(h))
assert rc == 0, lib.fdm_last_error()
t_create = time.perf_counter()


def integrate(k):
    s = wl.scan(k)
    tbs = np.ascontiguousarray(wl.T_base_sensor.T).reshape(16)
    twb = np.ascontiguousarray(wl.pose(k).T).reshape(16)
    st = capi.FdmScanStats()
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    t = time.perf_counter()
    rc = lib.fdm_engine_integrate(h, s["x"].size, p(s["x"]), p(s["y"]), p(s["z"]), p(s["intensity"]), p(s.get("rgb")), None,
                                  tbs.ctypes.data_as(C.POINTER(C.c_double)), twb.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))
    assert rc == 0, rc
    return time.perf_counter() - t


first, second, third = integrate(0), integrate(1), integrate(2)
print(json.dumps({"workload": wl.name, "import_numpy_s": round(t_np - t0, 3), "dlopen_engine_s": round(t_lib - t_np, 3),
                  "engine_create_s": round(t_create - t1, 3), "first_integrate_s": round(first, 4),
                  "second_integrate_s": round(second, 5), "third_integrate_s": round(third, 5),
                  "lib_MB": round(os.path.getsize(capi.LIB_PATH) / 1e6, 2)}))
