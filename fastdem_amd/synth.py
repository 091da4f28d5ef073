"""Synthetic scans for the five BASELINE.json configurations.

Deterministic numpy generators (PCG64, fixed seeds) — the same arrays feed the HIP engine,
the CPU oracle and the golden fixtures.  Scene: the sinusoidal terrain of the reference's
demo generator (z = 0.3 sin(0.5x) cos(0.5y), fastdem/examples/common/data_loader.hpp:47)
plus its two boxes (data_loader.hpp:87-99) and a cylindrical wall around the sensor.
Points are returned in the SENSOR frame as float32 SoA, which is what
FastDEM::integrate(cloud, T_base_sensor, T_world_base) consumes.
"""
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def translate(x, y, z):
    T = np.eye(4, dtype=np.float64)
    T[:3, 3] = (x, y, z)
    return T


def terrain_height(x, y):
    h = 0.3 * np.sin(0.5 * x) * np.cos(0.5 * y)
    box1 = (x >= 1.0) & (x <= 2.0) & (y >= 1.0) & (y <= 2.0)
    box2 = (x >= -3.0) & (x <= -2.0) & (y >= -1.0) & (y <= 0.0)
    h = np.where(box1, np.maximum(h, 0.5), h)
    h = np.where(box2, np.maximum(h, 0.8), h)
    return h


def _cast_rays(origin, dirs_world, wall_r):
    """Range along each world-frame unit ray to the terrain or the cylindrical wall."""
    dx, dy, dz = dirs_world[:, 0], dirs_world[:, 1], dirs_world[:, 2]
    hxy = np.sqrt(dx * dx + dy * dy)
    t_wall = wall_r / np.maximum(hxy, 1e-9)
    down = dz < -1e-6
    t = np.where(down, -origin[2] / np.where(down, dz, -1.0), t_wall)
    for _ in range(8):  # fixed-point refinement onto the height field
        hx = origin[0] + t * dx
        hy = origin[1] + t * dy
        t_new = (terrain_height(hx, hy) - origin[2]) / np.where(down, dz, -1.0)
        t = np.where(down, np.clip(t_new, 0.05, None), t)
    return np.where(down & (t < t_wall), t, t_wall)


@dataclass
class Workload:
    name: str
    width: float
    height: float
    resolution: float
    mode: int                 # 0 LOCAL, 1 GLOBAL
    estimation_type: int      # 0 Kalman, 1 P2
    sensor_type: int          # 0 Constant, 1 LiDAR, 2 RGBD
    z_min: float
    z_max: float
    range_min: float
    range_max: float
    T_base_sensor: np.ndarray
    pose: Callable[[int], np.ndarray]   # scan index -> T_world_base
    scans: list = field(default_factory=list)  # list of dicts x,y,z,intensity,rgb
    position: tuple = (0.0, 0.0)

    def scan(self, k):
        return self.scans[k % len(self.scans)]

    @property
    def n_points(self):
        return int(self.scans[0]["x"].size)

    def apply_to(self, cfg):
        """Fill an fdm_config-like ctypes struct (engine or oracle flavour)."""
        cfg.mode = self.mode
        cfg.estimation_type = self.estimation_type
        cfg.sensor_type = self.sensor_type
        cfg.z_min, cfg.z_max = self.z_min, self.z_max
        cfg.range_min, cfg.range_max = self.range_min, self.range_max
        return cfg


def _lidar_scan(rng, n_beams, elev_lo, elev_hi, n_az, T_ws, wall_r, order, intensity=True):
    elev = np.deg2rad(np.linspace(elev_lo, elev_hi, n_beams))
    az = np.arange(n_az) * (2.0 * np.pi / n_az)
    if order == "azimuth":   # firing order: all lasers of one azimuth step, then the next
        A, E = np.meshgrid(az, elev, indexing="ij")
    else:                    # ring-major (organised cloud, one row per laser)
        E, A = np.meshgrid(elev, az, indexing="ij")
    A, E = A.ravel(), E.ravel()
    d_s = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=1)
    R = T_ws[:3, :3]
    origin = T_ws[:3, 3]
    rng_true = _cast_rays(origin, d_s @ R.T, wall_r)
    r = rng_true + rng.normal(0.0, 0.02, size=rng_true.shape)
    p = (d_s * r[:, None]).astype(np.float32)
    out = {"x": np.ascontiguousarray(p[:, 0]), "y": np.ascontiguousarray(p[:, 1]),
           "z": np.ascontiguousarray(p[:, 2]), "intensity": None, "rgb": None}
    if intensity:
        out["intensity"] = rng.random(r.size, dtype=np.float32)
    return out


def vlp16(n_scans=4, seed=42, order="azimuth"):
    """C1/C2: VLP-16, 16 x 1800 = 28 800 pts, 15x15 m @ 0.1 m, Kalman, LiDAR model, LOCAL,
    default.yaml filters (z in [-1,2], range in [0.5,20]; fastdem/config/default.yaml:19-23)."""
    rng = np.random.default_rng(seed)
    Tbs = translate(0.0, 0.0, 0.6)

    def pose(k):
        return translate(0.05 * k, 0.0, 0.0) @ rot_z(np.deg2rad(0.2) * k)

    wl = Workload("vlp16_30k_15x15m_0.1m_kalman", 15.0, 15.0, 0.1, 0, 0, 1, -1.0, 2.0, 0.5, 20.0,
                  Tbs, pose)
    for k in range(n_scans):
        wl.scans.append(_lidar_scan(rng, 16, -15.0, 15.0, 1800, pose(k) @ Tbs, 7.0, order))
    return wl


def lidar128(n_scans=2, seed=44, order="azimuth", n_az=16384):
    """C4: 128 x 16 384 = 2 097 152 pts, 60x60 m @ 0.05 m, LOCAL rolling window, Kalman;
    the pose advances 0.4 m per scan so every scan triggers an 8-cell shift."""
    rng = np.random.default_rng(seed)
    Tbs = translate(0.0, 0.0, 1.8)

    def pose(k):
        return translate(0.4 * k, 0.0, 0.0) @ rot_z(np.deg2rad(0.2) * k)

    wl = Workload("lidar128_2m_60x60m_0.05m_kalman_rolling", 60.0, 60.0, 0.05, 0, 0, 1, -2.0, 5.0,
                  0.5, 40.0, Tbs, pose)
    for k in range(n_scans):
        wl.scans.append(_lidar_scan(rng, 128, -22.5, 22.5, n_az, pose(k) @ Tbs, 28.0, order))
    return wl


def global_map(n_scans=2, seed=45, size_m=400.0, n_az=16384, radius=150.0):
    """C5: GLOBAL fixed-origin map 400x400 m @ 0.05 m, C4-type scans, robot on a circle so
    successive scans cross tile borders."""
    rng = np.random.default_rng(seed)
    Tbs = translate(0.0, 0.0, 1.8)

    def pose(k):
        a = 0.4 * k / radius
        return translate(radius * np.cos(a), radius * np.sin(a), 0.0) @ rot_z(a + np.pi / 2)

    wl = Workload("global_400x400m_0.05m_kalman", size_m, size_m, 0.05, 1, 0, 1, -2.0, 5.0, 0.5,
                  40.0, Tbs, pose)
    for k in range(n_scans):
        wl.scans.append(_lidar_scan(rng, 128, -22.5, 22.5, n_az, pose(k) @ Tbs, 28.0, "azimuth"))
    return wl


def rgbd(n_scans=2, seed=43, width=640, height=480):
    """C3: 640x480 pinhole (fx=fy=386), optical frame, camera 0.8 m up pitched 35 deg down,
    depth outside [0.2, 3.25] m dropped (rgbd_model.hpp:43-48), colour channel, P2, LOCAL,
    10x10 m @ 0.05 m."""
    rng = np.random.default_rng(seed)
    fx = fy = 386.0
    cx, cy = width / 2.0, height / 2.0
    # optical (x right, y down, z forward) -> body (x forward, y left, z up)
    R_bo = np.array([[0, 0, 1, 0], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    Tbs = translate(0.0, 0.0, 0.8) @ rot_y(np.deg2rad(35.0)) @ R_bo

    def pose(k):
        return translate(0.02 * k, 0.0, 0.0) @ rot_z(np.deg2rad(0.3) * k)

    wl = Workload("rgbd_640x480_10x10m_0.05m_p2", 10.0, 10.0, 0.05, 0, 1, 2, -1.0, 2.0, 0.2, 5.0,
                  Tbs, pose)
    v, u = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    u, v = u.ravel().astype(np.float64), v.ravel().astype(np.float64)
    d_o = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=1)
    zscale = np.linalg.norm(d_o, axis=1)
    d_unit = d_o / zscale[:, None]
    for k in range(n_scans):
        T_ws = pose(k) @ Tbs
        t = _cast_rays(T_ws[:3, 3], d_unit @ T_ws[:3, :3].T, 6.0)
        depth = t / zscale
        # Nguyen et al. axial noise: sigma = a + b (z - c)^2 (config/sensor_model.hpp:27-32)
        depth = depth + rng.normal(0.0, 1.0, depth.shape) * (0.001 + 0.002 * (depth - 0.4) ** 2)
        ok = (depth >= 0.2) & (depth <= 3.25)
        p = (d_o[ok] * depth[ok, None]).astype(np.float32)
        hit = T_ws[:3, 3] + (d_unit[ok] * t[ok, None]) @ T_ws[:3, :3].T
        r = (127.5 * (1.0 + np.sin(3.0 * hit[:, 0]))).astype(np.uint32)
        g = (127.5 * (1.0 + np.cos(3.0 * hit[:, 1]))).astype(np.uint32)
        b = np.clip(255.0 * (hit[:, 2] + 0.5), 0, 255).astype(np.uint32)
        wl.scans.append({"x": np.ascontiguousarray(p[:, 0]), "y": np.ascontiguousarray(p[:, 1]),
                         "z": np.ascontiguousarray(p[:, 2]), "intensity": None,
                         "rgb": ((r << 16) | (g << 8) | b).astype(np.uint32)})
    return wl


WORKLOADS = {
    "c1": vlp16, "c2": vlp16, "vlp16": vlp16,
    "c3": rgbd, "rgbd": rgbd,
    "c4": lidar128, "lidar128": lidar128,
    "c5": global_map, "global": global_map,
}


def make(name, **kw):
    return WORKLOADS[name](**kw)
