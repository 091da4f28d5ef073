#!/usr/bin/env python3
"""pytest against ANOTHER build of the engine library (same ABI): does a regression test fail where it should?
    python scripts/pytest_with_lib.py fastdem_amd/lib/libfdm_engine_x.so tests/test_x.py -k name ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastdem_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:] + ["-q", "-m", "gpu", "--no-header", "--tb=line"]))
