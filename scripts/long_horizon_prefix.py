#!/usr/bin/env python3
"""Does the long-horizon generator (tests/test_long_horizon_gpu.py) reach the round-5 stale-obstacle bug?  Runs the test
against a build of the tree BEFORE the fix f53d0fc (fastdem_amd/lib/libfdm_engine_prefix.so: `git archive f53d0fc^
fastdem_amd/csrc include`, make) — it must FAIL there and pass on the shipped library.
    python scripts/long_horizon_prefix.py [path/to/lib.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastdem_amd import capi
capi.LIB_PATH = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fastdem_amd", "lib", "libfdm_engine_prefix.so")
import pytest
sys.exit(pytest.main([os.path.join(ROOT, "tests", "test_long_horizon_gpu.py"), "-q", "-m", "gpu", "-k", "default", "--no-header", "-x", "--tb=line"]))
