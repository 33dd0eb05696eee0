"""Seeded synthetic inputs shared by the golden-vector generator and the tests.

The golden generator (make_golden.py, runs only where /root/reference exists)
and the parity tests (run anywhere) must feed byte-identical inputs to the
reference and to the build, so both import the builders below.  Every fixture
stores a sha256 of its inputs; the tests re-derive the inputs from the seed and
check the digest before they compare outputs.
"""
import hashlib

import numpy as np


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def rot_y(angle: float) -> np.ndarray:
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=np.float64)


def rot_xyz(ax: float, ay: float, az: float) -> np.ndarray:
    cx, sx = np.cos(ax), np.sin(ax)
    cy, sy = np.cos(ay), np.sin(ay)
    cz, sz = np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return rz @ ry @ rx


def frame(seed: int, h: int, w: int, lo: float = 1.0, hi: float = 2.0,
          plant: bool = True):
    """Random depth in [lo, hi) f32 + random BGR u8, with planted invalid pixels.

    Planted values cover every branch of the reference's validity mask
    (depth_to_reconstruction.py:359-361): 0, negative, NaN, +inf, a value above
    any max_depth in use, and the exact float32 images of the 0.1 / 50 / 100
    thresholds and their float32 neighbours.
    """
    rng = np.random.default_rng(seed)
    depth = (lo + (hi - lo) * rng.random((h, w))).astype(np.float32)
    color = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    if plant and h * w >= 64:
        flat = depth.reshape(-1)
        idx = rng.choice(h * w, size=min(48, h * w // 2), replace=False)
        specials = np.array([
            0.0, -1.0, np.nan, np.inf, -np.inf, 1000.0,
            np.float32(0.1), np.nextafter(np.float32(0.1), np.float32(1)),
            np.nextafter(np.float32(0.1), np.float32(0)),
            np.float32(50.0), np.nextafter(np.float32(50.0), np.float32(0)),
            np.float32(100.0), np.nextafter(np.float32(100.0), np.float32(0)),
            np.float32(75.0), np.float32(0.08), np.float32(40.0),
        ], dtype=np.float32)
        flat[idx] = np.resize(specials, idx.shape)
        # pixel (0,0) is always sampled by every stride: make it an edge value
        flat[0] = np.float32(1.5)
    return depth, color


def pose(seed: int):
    rng = np.random.default_rng(1000 + seed)
    a = rng.uniform(-0.3, 0.3, size=3)
    r = rot_xyz(*a)
    t = rng.uniform(-0.5, 0.5, size=(3, 1))
    return r, t


# intrinsics in use
K_A = dict(fx=525.0, fy=525.0, cx=320.0, cy=240.0)            # 640x480
K_B = dict(fx=1719.0, fy=1719.0, cx=540.0, cy=960.0)          # 1080x1920 portrait (D2R:49-52)
K_S = dict(fx=70.0, fy=72.0, cx=39.5, cy=29.25)               # small frames, non-integer c
