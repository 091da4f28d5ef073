"""fastdem_amd — MI355X-native elevation-map update engine behind the FastDEM integrate() path.

Layout (only what the path needs):
  csrc/        hand-written HIP kernels + the C ABI (include/fdm_engine.h) -> lib/libfdm_engine.so
  capi.py      ctypes declarations of that ABI (plumbing)
  engine.py    Python handle used by tests / bench.py
  synth.py     synthetic scans of the BASELINE.json configurations
  tiling.py    multi-GPU spatial tiling + RCCL halo exchange of one global map
  cpp/         C++17 host mirror of fastdem::FastDEM / ElevationMap over the C ABI
"""
from . import capi, synth  # noqa: F401
from .engine import Engine, EngineError, HostArray, host_array  # noqa: F401

__all__ = ["Engine", "EngineError", "HostArray", "host_array", "capi", "synth"]
