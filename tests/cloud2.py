"""Builds sensor_msgs/PointCloud2-style byte blobs for the ingest tests (test data, not product)."""
import numpy as np


class Layout:
    """Duck-typed fdm_cloud2_layout source (copied field by field into the ctypes structs)."""

    def __init__(self, point_step, off_x, off_y, off_z, off_intensity=-1, intensity_type=7, off_rgb=-1):
        self.point_step, self.off_x, self.off_y, self.off_z = point_step, off_x, off_y, off_z
        self.off_intensity, self.intensity_type, self.off_rgb = off_intensity, intensity_type, off_rgb


INTENSITY_DTYPES = {2: "<u1", 4: "<u2", 7: "<f4", 8: "<f8", 5: "<i4"}


def make_blob(x, y, z, intensity=None, intensity_type=7, rgb=None, offsets=None, point_step=None,
              rng=None, lead=0):
    """Pack channels into records.  offsets = dict(x=, y=, z=, intensity=, rgb=); the rest of each
    record (ring / time / padding in a real message) is random bytes.  `lead` bytes precede the
    first record (to make the whole blob unaligned)."""
    n = len(x)
    off = dict(x=0, y=4, z=8)
    cur = 12
    if intensity is not None:
        off["intensity"] = cur
        cur += np.dtype(INTENSITY_DTYPES[intensity_type]).itemsize
    if rgb is not None:
        off["rgb"] = cur
        cur += 4
    if offsets:
        off.update(offsets)
    step = point_step or max(cur, max(off.values()) + 8)
    rng = rng or np.random.default_rng(0)
    rec = rng.integers(0, 256, (n, step), dtype=np.uint8)

    def put(name, arr, dt):
        b = np.ascontiguousarray(np.asarray(arr).astype(dt)).view(np.uint8).reshape(n, -1)
        rec[:, off[name]:off[name] + b.shape[1]] = b

    put("x", x, "<f4"), put("y", y, "<f4"), put("z", z, "<f4")
    if intensity is not None:
        put("intensity", intensity, INTENSITY_DTYPES[intensity_type])
    if rgb is not None:
        put("rgb", rgb, "<u4")
    lay = Layout(step, off["x"], off["y"], off["z"], off.get("intensity", -1), intensity_type, off.get("rgb", -1))
    blob = np.concatenate([np.zeros(lead, np.uint8), rec.reshape(-1)])
    return blob[lead:], lay
