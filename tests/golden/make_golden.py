"""Regenerates the golden fixtures from the CPU oracle (the reference itself cannot be built or
imported here — SURVEY.md §8c — so the oracle, pinned by tests/test_oracle_reference_spec.py, is
the generator).  Fixtures are DATA: inputs + expected outputs.  Usage: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

import fdm_ref_py as R  # noqa: E402
from fastdem_amd import synth  # noqa: E402

SNAP_AT = (1, 2, 10)


def run(name, width, height, res, fill, scans, Tbs, poses):
    cfg = R.default_config()
    fill(cfg)
    ref = R.RefEngine(width, height, res, cfg)
    ref.enable_cell_ids()
    out = {"width": width, "height": height, "resolution": res, "T_base_sensor": Tbs,
           "poses": np.stack(poses), "n_scans": len(poses), "n_distinct": len(scans)}
    for k, _ in R.RefConfig._fields_:
        v = getattr(cfg, k)
        out["cfg_" + k] = np.array(list(v) if k == "p2_dn" else v)
    for i, s in enumerate(scans):
        for ch in ("x", "y", "z", "intensity", "rgb"):
            if s.get(ch) is not None:
                out[f"scan{i}_{ch}"] = s[ch]
    for k, pose in enumerate(poses):
        s = scans[k % len(scans)]
        rc, st = ref.integrate(s["x"], s["y"], s["z"], Tbs, pose, intensity=s.get("intensity"),
                               rgb=s.get("rgb"))
        out[f"status_{k + 1}"] = rc
        out[f"stats_{k + 1}"] = np.array([st[f] for f in ("n_input", "n_after_filter", "n_in_map",
                                                          "n_cells_touched", "shift_rows", "shift_cols")])
        if k + 1 in SNAP_AT:
            out[f"ids_{k + 1}"] = ref.last_cell_ids(s["x"].size)
            g = ref.geometry()
            out[f"geom_{k + 1}"] = np.array([g.position_x, g.position_y, g.start_row, g.start_col])
            for lname in ref.layers():
                out[f"layer_{k + 1}_{lname}"] = ref.layer(lname)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


def main():
    # 1) LiDAR model, Kalman, LOCAL with row, column and wrap-around shifts, intensity channel
    rng = np.random.default_rng(1234)
    Tbs = synth.translate(0.0, 0.0, 0.6)
    poses = [synth.translate(x, y, 0.0) @ synth.rot_z(a) for x, y, a in
             [(0, 0, 0), (0.26, 0, 0.01), (0.26, -0.33, 0.02), (-1.1, -0.33, 0.03), (-1.1, 2.4, 0.04),
              (3.3, 2.4, 0.05), (3.3, 2.45, 0.06), (3.9, 2.1, 0.07), (9.0, 2.1, 0.08), (9.05, 2.0, 0.09)]]
    scans = [synth._lidar_scan(rng, 16, -15.0, 15.0, 256, poses[k] @ Tbs, 7.0, "azimuth")
             for k in range(3)]

    def fill1(c):
        c.mode, c.estimation_type, c.sensor_type = 0, 0, 1
        c.z_min, c.z_max, c.range_min, c.range_max = -1.0, 2.0, 0.5, 20.0

    run("lidar_kalman_local", 8.0, 8.0, 0.1, fill1, scans, Tbs, poses)

    # 2) RGB-D model, P2 quantile, colour channel (reduced 160x120 image)
    wl = synth.rgbd(n_scans=3, width=160, height=120)
    # the reduced image keeps fx=386, so it covers a narrow patch; fine for a fixture

    def fill2(c):
        wl.apply_to(c)

    run("rgbd_p2_colour", 5.0, 5.0, 0.05, fill2, wl.scans, wl.T_base_sensor,
        [wl.pose(k) for k in range(10)])


if __name__ == "__main__":
    main()
