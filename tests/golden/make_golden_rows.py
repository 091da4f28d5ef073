"""Golden fixture for the rows either side of the hot path: one small scene run
through ingest -> integrate with raycasting -> stencil post-processing -> egress by the CPU oracle
(the reference cannot be built here; the oracle is pinned by tests/test_oracle_*_spec.py).
Fixtures are DATA: inputs + expected outputs.  Usage: python tests/golden/make_golden_rows.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import fdm_ref_py as R  # noqa: E402
from cloud2 import make_blob  # noqa: E402
from fastdem_amd import synth  # noqa: E402


def main():
    rng = np.random.default_rng(4321)
    Tbs = synth.translate(0.0, 0.0, 0.6)
    poses = [synth.translate(x, y, 0.0) @ synth.rot_z(a) for x, y, a in
             [(0, 0, 0), (0.31, -0.12, 0.02), (0.62, -0.2, 0.04), (0.9, -0.33, 0.06)]]
    cfg = R.default_config()
    cfg.z_min, cfg.z_max, cfg.range_min, cfg.range_max = -1.0, 2.0, 0.5, 20.0
    cfg.raycast_enabled = 1
    cfg.rc_log_odds_ghost, cfg.rc_clear_threshold = 0.7, -1.0
    ref = R.RefEngine(8.0, 8.0, 0.1, cfg)
    out = {"width": 8.0, "height": 8.0, "resolution": 0.1, "T_base_sensor": Tbs, "poses": np.stack(poses)}
    for k, _ in R.RefConfig._fields_:
        v = getattr(cfg, k)
        out["cfg_" + k] = np.array(list(v) if k == "p2_dn" else v)
    for k, pose in enumerate(poses):
        s = synth._lidar_scan(rng, 16, -15.0, 15.0, 256, pose @ Tbs, 3.5, "azimuth")
        x = s["x"].copy()
        x[rng.uniform(size=x.size) < 0.03] = np.nan  # dropped returns: from_impl filters them
        blob, lay = make_blob(x, s["y"], s["z"], intensity=s["intensity"], point_step=24, rng=rng)
        out[f"blob_{k}"] = blob
        out["layout"] = np.array([lay.point_step, lay.off_x, lay.off_y, lay.off_z, lay.off_intensity,
                                  lay.intensity_type, lay.off_rgb])
        if k == 1:  # phantom obstacle on the rays' way
            e = ref.layer("elevation")
            e[30:38, 36:44] = 1.2
            ref.set_layer("elevation", e)
            out["ghost_rows"], out["ghost_cols"], out["ghost_value"] = np.array([30, 38]), np.array([36, 44]), 1.2
        rc, st = ref.integrate_cloud2(blob, x.size, lay, Tbs, pose)
        assert rc == 0
        out[f"ray_stats_{k}"] = np.array(list(ref.last_ray_stats().values()))
    assert sum(out[f"ray_stats_{k}"][4] for k in range(4)) > 0
    decoded = R.from_cloud2(out["blob_3"], out["blob_3"].size // 24, lay)
    for ch in ("x", "y", "z", "intensity"):
        out["decoded3_" + ch] = decoded[ch]
    ref.apply_uncertainty_fusion(True, 0.25, 0.1, 0.01, 0.99, 3)
    ref.apply_inpainting(3, 2)
    ref.apply_spatial_smoothing("elevation_inpainted", 3, 5)
    ref.apply_feature_extraction(0.3, 4, 0.05, 0.95)
    g = ref.geometry()
    out["geom"] = np.array([g.position_x, g.position_y, g.start_row, g.start_col])
    out["layer_names"] = np.array(ref.layers())
    for name in ref.layers():
        out["layer_" + name] = ref.layer(name)
    fields, step, data = ref.pack_cloud()
    out["cloud_fields"], out["cloud_step"], out["cloud_data"] = np.array(fields), step, data
    path = os.path.join(HERE, "widened_rows.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB", "layers:", len(ref.layers()), "cloud points:", data.shape[0])


if __name__ == "__main__":
    main()
