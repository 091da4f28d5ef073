#!/usr/bin/env python3
"""bench.py with another build of the engine library: lib_bench.py <path/libfdm_engine_x.so> [bench args...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastdem_amd import capi
capi.LIB_PATH = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench
bench.main()
