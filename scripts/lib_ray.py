#!/usr/bin/env python3
"""ray_bench.py with another build of the engine library: lib_ray.py <path/libfdm_engine_x.so> [ray_bench args...]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastdem_amd import capi
capi.LIB_PATH = sys.argv[1]
sys.argv = [os.path.join(ROOT, "scripts", "ray_bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
