"""Feature extraction: fraction of bit-identical cells and worst differences, engine vs oracle, with the oracle on
the platform libm (trig_mode 0) and on correctly rounded trig (trig_mode 1).  Run on the GPU box."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fdm_ref_py as R
from fastdem_amd import Engine, capi
F32=np.float32
def terrain(rng, shape, holes=0.3, noise=0.02):
    r, c = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    z = 0.4 * np.sin(r * 0.11) * np.cos(c * 0.07) + 0.002 * r + rng.normal(0, noise, shape)
    z = z.astype(F32); z[rng.uniform(size=shape) < holes] = np.nan
    return z
for mode in (0,1):
  R.set_trig_mode(mode)
  for seed, size, radius in ((24, 20.0, 0.3), (7, 30.0, 0.35), (9, 60.0, 0.3)):
      rng = np.random.default_rng(seed)
      eng = Engine(size, size, 0.05, capi.default_config()); ref = R.RefEngine(size, size, 0.05, R.default_config())
      shape = eng.layer("elevation").shape
      el = terrain(rng, shape, holes=0.15, noise=0.01); el[:, shape[1]//2:] += F32(0.3)
      for o in (eng, ref):
          o.set_layer("elevation", el); o.apply_feature_extraction(radius, 4, 0.05, 0.95)
      for n in ("step","roughness","curvature","_normal_x","_normal_y","_normal_z","slope"):
          a, b = eng.layer(n), ref.layer(n)
          ok = np.isfinite(b)
          same = (a.view(np.uint32)[ok] == b.view(np.uint32)[ok])
          d = np.abs(a[ok].astype(np.float64)-b[ok])
          ulp = np.abs(a.view(np.int32)[ok].astype(np.int64)-b.view(np.int32)[ok].astype(np.int64))
          print(mode, seed, n, "cells", ok.sum(), "bit-identical %.5f" % same.mean(), "max abs %.3e" % d.max(), "max ulp", ulp.max(), "n>1ulp", (ulp>1).sum(), "n>4ulp", (ulp>4).sum())
  