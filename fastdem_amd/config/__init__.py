"""YAML configuration -> fdm_config, the Python twin of fastdem::loadConfig
(fastdem/src/config_fastdem.cpp:57-126 parse, :128-260 validate): every key optional, unknown enum
strings warn and fall back, min_variance >= max_variance and unsorted P2 markers raise ValueError,
other out-of-range values warn and are clamped.  Host-side plumbing only."""
import os
import warnings

import yaml

from .. import capi

DEFAULT_YAML = os.path.join(os.path.dirname(os.path.abspath(__file__)), "default.yaml")

_MODE = {"local": 0, "global": 1}
_EST = {"kalman_filter": 0, "p2_quantile": 1}
_SENSOR = {"constant": 0, "none": 0, "lidar": 1, "laser": 1, "rgbd": 2}


def _warn(msg):
    warnings.warn("[Config] " + msg, stacklevel=3)


def _enum(table, value, what, default_name):
    if value in table:
        return table[value]
    _warn(f"Unknown {what} '{value}', defaulting to {default_name}")
    return table[default_name]


def parse_config(root):
    """dict (parsed YAML) -> validated FdmConfig."""
    cfg = capi.default_config()
    root = root or {}
    if not isinstance(root, dict):
        raise RuntimeError("config root must be a mapping")

    def load(node, key, attr, conv=float):
        if isinstance(node, dict) and key in node and node[key] is not None:
            try:
                setattr(cfg, attr, conv(node[key]))
            except ValueError as exc:  # yaml-cpp: bad conversion -> YAML::Exception -> runtime_error
                raise RuntimeError(f"bad conversion of '{key}': {node[key]!r}") from exc

    m = root.get("mapping") or {}
    if m.get("mode"):
        cfg.mode = _enum(_MODE, m["mode"], "mapping mode", "local")
    if m.get("type"):
        cfg.estimation_type = _enum(_EST, m["type"], "estimation type", "kalman_filter")
    k = m.get("kalman") or {}
    load(k, "min_variance", "kalman_min_variance")
    load(k, "max_variance", "kalman_max_variance")
    load(k, "process_noise", "kalman_process_noise")
    p = m.get("p2") or {}
    for i in range(5):
        if p.get(f"dn{i}") is not None:
            cfg.p2_dn[i] = float(p[f"dn{i}"])
    load(p, "elevation_marker", "p2_elevation_marker", int)
    load(p, "max_sample_count", "p2_max_sample_count")
    f = root.get("point_filter") or {}
    for key in ("z_min", "z_max", "range_min", "range_max"):
        load(f, key, key)
    r = root.get("raycasting") or {}
    load(r, "enabled", "raycast_enabled", lambda v: int(bool(v)))
    for key in ("height_conflict_threshold", "log_odds_observed", "log_odds_ghost", "log_odds_max",
                "clear_threshold"):
        load(r, key, "rc_" + key)
    s = root.get("sensor_model") or {}
    if s.get("type"):
        cfg.sensor_type = _enum(_SENSOR, s["type"], "sensor_model.type", "lidar")
    load(s.get("lidar"), "range_noise", "lidar_range_noise")
    load(s.get("lidar"), "angular_noise", "lidar_angular_noise")
    for key in ("normal_a", "normal_b", "normal_c", "lateral_factor"):
        load(s.get("rgbd"), key, "rgbd_" + key)
    load(s.get("constant"), "uncertainty", "constant_uncertainty")
    return validate(cfg)


def validate(cfg):
    """detail::validate (config_fastdem.cpp:128-260)."""
    if cfg.kalman_min_variance >= cfg.kalman_max_variance:
        raise ValueError(f"mapping.kalman: min_variance ({cfg.kalman_min_variance}) >= max_variance "
                         f"({cfg.kalman_max_variance})")

    def clamp(attr, bad, to, rule):
        if bad(getattr(cfg, attr)):
            _warn(f"{attr} ({getattr(cfg, attr)}) must be {rule}, clamping to {to}")
            setattr(cfg, attr, to)

    if cfg.raycast_enabled:
        clamp("rc_height_conflict_threshold", lambda v: v <= 0, 0.05, "> 0")
        clamp("rc_log_odds_observed", lambda v: v <= 0, 0.4, "> 0")
        clamp("rc_log_odds_ghost", lambda v: v <= 0, 0.2, "> 0")
        clamp("rc_log_odds_max", lambda v: v <= 0, 2.0, "> 0")
        clamp("rc_clear_threshold", lambda v: v >= 0, -1.0, "< 0")
    clamp("kalman_min_variance", lambda v: v <= 0, 0.0001, "> 0")
    clamp("kalman_process_noise", lambda v: v < 0, 0.0, ">= 0")
    if not 0 <= cfg.p2_elevation_marker <= 4:
        _warn("mapping.p2.elevation_marker out of range [0, 4], clamping")
        cfg.p2_elevation_marker = min(max(cfg.p2_elevation_marker, 0), 4)
    for i in range(5):
        if not 0.0 <= cfg.p2_dn[i] <= 1.0:
            _warn(f"mapping.p2.dn{i} ({cfg.p2_dn[i]}) out of [0, 1], clamping")
            cfg.p2_dn[i] = min(max(cfg.p2_dn[i], 0.0), 1.0)
    if any(cfg.p2_dn[i] > cfg.p2_dn[i + 1] for i in range(4)):
        raise ValueError("mapping.p2: markers must be sorted (dn0 <= dn1 <= dn2 <= dn3 <= dn4)")
    clamp("lidar_range_noise", lambda v: v <= 0, 0.02, "> 0")
    clamp("lidar_angular_noise", lambda v: v < 0, 0.0, ">= 0")
    clamp("constant_uncertainty", lambda v: v <= 0, 0.1, "> 0")
    for key in ("rgbd_normal_a", "rgbd_normal_b", "rgbd_normal_c", "rgbd_lateral_factor"):
        clamp(key, lambda v: v < 0, 0.0, ">= 0")
    return cfg


def load_config(path=DEFAULT_YAML):
    """fastdem::loadConfig: RuntimeError when the file cannot be read or parsed."""
    try:
        with open(path) as fh:
            root = yaml.safe_load(fh)
    except (OSError, yaml.YAMLError) as exc:
        raise RuntimeError(f"Failed to load config: {path} - {exc}") from exc
    try:
        return parse_config(root)
    except (TypeError, AttributeError, RuntimeError) as exc:  # structure mismatch / bad conversion
        raise RuntimeError(f"Failed to load config: {path} - {exc}") from exc
