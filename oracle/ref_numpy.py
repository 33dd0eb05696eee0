"""CPU ORACLE (numpy) -- TEST INFRASTRUCTURE ONLY.

Restates, on the CPU, the parts of the hot path that the reference pins
(SURVEY.md section 8a rows a3-a9).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product path
(textureless-3d-reconstruction_amd/) never does and fails loudly without its HIP
library.

Parity status: PINNED for back-projection / validity mask / pose transform / scale
estimation / vstack-merge / ASCII PLY by golden vectors captured from the reference
itself (tests/golden/make_golden.py imports depth_to_reconstruction.py and
depth_enhanced_reconstruction.py).  UNPINNED for the Open3D semantics restated in
voxel_centroid_open3d() and statistical_outlier_open3d(): Open3D is a third-party
dependency of the reference (pip `open3d`, version unpinned -- no requirements file
in the reference), absent from /root/reference and from this image, and the
reference holds no test or golden for it.  Those two functions restate Open3D's
published algorithm (VoxelDownSample: geometry/PointCloud.cpp; RemoveStatisticalOutliers:
geometry/PointCloud.cpp) at the call sites depth_to_reconstruction.py:406-418.

All file:line citations are relative to /root/reference.
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------------------
# a3  projection factors            depth_to_reconstruction.py:287-295, DER:545-552
# --------------------------------------------------------------------------------------
def projection_factors(h: int, w: int, fx: float, fy: float, cx: float, cy: float):
    """fp64 maps xf[v,u]=(u-cx)/fx, yf[v,u]=(v-cy)/fy for integer pixel indices."""
    xf_row = (np.arange(w, dtype=np.int64) - cx) / fx        # int64 - float -> fp64
    yf_col = (np.arange(h, dtype=np.int64) - cy) / fy
    xf = np.broadcast_to(xf_row[None, :], (h, w))
    yf = np.broadcast_to(yf_col[:, None], (h, w))
    return xf, yf


# --------------------------------------------------------------------------------------
# a4/a5  back-projection            depth_to_reconstruction.py:328-384, DER:554-613
# --------------------------------------------------------------------------------------
def backproject(depth, color, fx, fy, cx, cy, pose=None, scale=1.0, subsample=1,
                min_depth=0.1, max_depth=50.0):
    """Depth map -> (points f32[N,3] world, colors u8[N,3] RGB), row-major pixel order.

    Follows the reference step by step:
      stride subsample (D2R:349-353), depth*scale in numpy's promoted dtype (D2R:356; a
      Python float keeps f32, an np.float64 promotes to fp64), strict compares in that dtype
      (D2R:359-361), fp64 factors * z (D2R:364-366), P_w = R^T P_c - R^T t in fp64 (D2R:376),
      BGR->RGB (D2R:381-382), cast to f32 (D2R:384).
    The 3x3 transform is written as explicit column sums (no BLAS call) so the oracle gives the
    same bits on every host; vs the reference's `@` this differs only in the last fp64 bit and
    therefore in <1e-5 of the f32 outputs by 1 ulp (measured in tests/test_oracle_golden.py).
    """
    depth = np.asarray(depth)
    color = np.asarray(color)
    h, w = depth.shape
    s = int(subsample)
    xf_row = (np.arange(w, dtype=np.int64) - cx) / fx
    yf_col = (np.arange(h, dtype=np.int64) - cy) / fy
    if s > 1:
        depth = depth[::s, ::s]
        color = color[::s, ::s]
        xf_row = xf_row[::s]
        yf_col = yf_col[::s]
    d = depth * scale
    valid = (d > min_depth) & (d < max_depth) & np.isfinite(d)
    vv, uu = np.nonzero(valid)                      # row-major order of surviving pixels
    z = d[vv, uu]
    x = xf_row[uu] * z
    y = yf_col[vv] * z
    if pose is not None:
        r, t = pose
        r = np.asarray(r, dtype=np.float64)
        t = np.asarray(t, dtype=np.float64).reshape(3)
        c = np.array([r[0, i] * t[0] + r[1, i] * t[1] + r[2, i] * t[2] for i in range(3)])  # R^T t
        out = np.empty((z.shape[0], 3), dtype=np.float64)
        for i in range(3):
            out[:, i] = (r[0, i] * x + r[1, i] * y + r[2, i] * z) - c[i]
    else:
        out = np.stack([x, y, z], axis=-1) if z.size else np.zeros((0, 3))
    rgb = color[vv, uu][:, ::-1]
    return out.astype(np.float32), np.ascontiguousarray(rgb, dtype=np.uint8)


# --------------------------------------------------------------------------------------
# a6  depth scale                  depth_to_reconstruction.py:297-326, DER:659-697
# --------------------------------------------------------------------------------------
def estimate_scale(sparse_points, sparse_pts2d, depth_map, variant="d2r"):
    """Median of Z_sparse / depth[int(y), int(x)].

    variant "d2r": keep 0.001 < s < 1000 (D2R:315), < 3 samples -> 1.0 (D2R:318-320).
    variant "der": needs >= 5 input points (DER:673), no sanity clamp, < 3 samples -> 1.0.
    int() truncates toward zero, so -0.5 indexes column 0 (both files).
    """
    sparse_points = np.asarray(sparse_points)
    sparse_pts2d = np.asarray(sparse_pts2d)
    if variant == "der" and len(sparse_points) < 5:
        return 1.0
    h, w = depth_map.shape
    ratios = []
    for p3, p2 in zip(sparse_points, sparse_pts2d):
        px, py = int(p2[0]), int(p2[1])
        if not (0 <= px < w and 0 <= py < h):
            continue
        dn = depth_map[py, px]
        zs = p3[2]
        if not (dn > 0 and zs > 0):
            continue
        s = zs / dn
        if variant == "d2r" and not (0.001 < s < 1000):
            continue
        ratios.append(s)
    if len(ratios) < 3:
        return 1.0
    return np.median(ratios)


# --------------------------------------------------------------------------------------
# a8  pose algebra                 depth_to_reconstruction.py:543-546, 618-620, 633
# --------------------------------------------------------------------------------------
def compose_pose(r_rel, t_rel, r_prev, t_prev):
    """world->camera chain: R_c = R_rel R_prev, t_c = R_rel t_prev + t_rel (D2R:619-620)."""
    r_rel = np.asarray(r_rel, np.float64)
    return r_rel @ np.asarray(r_prev, np.float64), r_rel @ np.asarray(t_prev, np.float64).reshape(3, 1) + np.asarray(t_rel, np.float64).reshape(3, 1)


def camera_centre(r, t):
    """C = -R^T t (D2R:374-375, 740)."""
    return -(np.asarray(r, np.float64).T @ np.asarray(t, np.float64).reshape(3))


# --------------------------------------------------------------------------------------
# a7  merge                        depth_to_reconstruction.py:386-420, DER:615-645
# --------------------------------------------------------------------------------------
def merge_vstack(clouds):
    """The reference's branch when Open3D is absent: drop empty clouds, vstack (D2R:390-402)."""
    pts = [p for p, c in clouds if len(p) > 0]
    col = [c for p, c in clouds if len(p) > 0]
    if not pts:
        return np.array([]), np.array([])             # shape (0,), fp64 (D2R:398-399)
    return np.vstack(pts), np.vstack(col)


def voxel_centroid_open3d(points, colors_u8, voxel_size):
    """Open3D PointCloud::VoxelDownSample semantics [parity unpinned -- see module header].

    origin = min_bound - voxel/2; index = floor((p - origin)/voxel) in fp64; per voxel the fp64 mean
    of the points and of colours/255.  Output order here: ascending (ix, iy, iz) (Open3D's is the
    iteration order of an unordered_map, i.e. unspecified -- parity is set-based).
    Returns (points f64[M,3], colors f64[M,3] in [0,1], index i64[M,3], count i64[M], origin f64[3]).
    """
    p = np.asarray(points, dtype=np.float64)
    c = np.asarray(colors_u8, dtype=np.float64) / 255.0
    origin = p.min(axis=0) - voxel_size * 0.5
    idx = np.floor((p - origin) / voxel_size).astype(np.int64)
    uniq, inv, cnt = np.unique(idx, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    psum = np.zeros((len(uniq), 3))
    csum = np.zeros((len(uniq), 3))
    np.add.at(psum, inv, p)
    np.add.at(csum, inv, c)
    return psum / cnt[:, None], csum / cnt[:, None], uniq, cnt, origin


def colors_to_u8(colors01):
    """(c*255).astype(uint8): truncation, as the reference does after Open3D (D2R:418)."""
    return (np.asarray(colors01) * 255).astype(np.uint8)


def statistical_outlier_open3d(points, nb_neighbors=20, std_ratio=2.0):
    """Open3D RemoveStatisticalOutliers semantics [parity unpinned -- see module header].

    For every point: mean distance to its nb_neighbors nearest neighbours, the query point itself
    included (KNN search returns it at distance 0).  mu / sigma = mean / Bessel-corrected std of those
    means over the valid points; keep  mean_i > 0  and  mean_i < mu + std_ratio*sigma.
    Returns the boolean keep mask.
    """
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float64)
    n = len(p)
    if n == 0:
        return np.zeros(0, dtype=bool)
    k = min(nb_neighbors, n)
    dist, _ = cKDTree(p).query(p, k=k)
    dist = dist.reshape(n, k)
    mean_d = dist.mean(axis=1)
    ok = mean_d > 0
    m = int(ok.sum())
    if m <= 1:
        return ok
    mu = mean_d[ok].sum() / m
    sigma = np.sqrt(((mean_d[ok] - mu) ** 2).sum() / (m - 1))
    return ok & (mean_d < mu + std_ratio * sigma)


def merge_open3d(clouds, voxel_size, sor=True, nb_neighbors=20, std_ratio=2.0):
    """What depth_to_reconstruction.py:386-420 returns WITH Open3D present (sor=False: DER:615-645)."""
    pts, col = merge_vstack(clouds)
    if len(pts) == 0 or voxel_size <= 0:
        return pts, col
    p, c, _, _, _ = voxel_centroid_open3d(pts, col, voxel_size)
    if sor:
        keep = statistical_outlier_open3d(p, nb_neighbors, std_ratio)
        p, c = p[keep], c[keep]
    return p, colors_to_u8(c)


# --------------------------------------------------------------------------------------
# a9  PLY                          depth_to_reconstruction.py:689-701, DER:1296-1309
# --------------------------------------------------------------------------------------
def ply_ascii_text(points, colors):
    """The reference's ASCII fallback: header + one `x y z r g b` line per point.

    Coordinates print as Python's str() of the numpy scalar (repr-shortest of a float32 for f32
    input); colours as integers.
    """
    lines = ["ply", "format ascii 1.0", f"element vertex {len(points)}",
             "property float x", "property float y", "property float z",
             "property uchar red", "property uchar green", "property uchar blue", "end_header"]
    for p, c in zip(points, colors):
        lines.append(f"{p[0]} {p[1]} {p[2]} {int(c[0])} {int(c[1])} {int(c[2])}")
    return "\n".join(lines) + "\n"


def read_ply(path):
    """Minimal PLY reader (ascii / binary_little_endian; x y z [+ red green blue]) for tests."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt = None
        n = 0
        props = []
        while True:
            line = f.readline().decode().strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element vertex"):
                n = int(line.split()[2])
            elif line.startswith("property"):
                _, ty, name = line.split()
                props.append((name, ty))
            elif line == "end_header":
                break
        tymap = {"float": "<f4", "double": "<f8", "uchar": "u1", "float32": "<f4", "float64": "<f8", "uint8": "u1"}
        if fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float64, ndmin=2) if n else np.zeros((0, len(props)))
            cols = {name: data[:, i] for i, (name, _) in enumerate(props)}
        else:
            dt = np.dtype([(name, tymap[ty]) for name, ty in props])
            rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
            cols = {name: rec[name] for name, _ in props}
    pts = np.stack([cols["x"], cols["y"], cols["z"]], axis=1).astype(np.float64)
    if "red" in cols:
        col = np.stack([cols["red"], cols["green"], cols["blue"]], axis=1).astype(np.uint8)
    else:
        col = np.zeros((n, 3), np.uint8)
    return pts, col


# --------------------------------------------------------------------------------------
# set-based parity metric (vertex order is unspecified)
# --------------------------------------------------------------------------------------
def chamfer_mean(a, b):
    """Symmetric mean Chamfer distance: (mean_a min_b |a-b| + mean_b min_a |a-b|) / 2, in metres."""
    from scipy.spatial import cKDTree
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if len(a) == 0 or len(b) == 0:
        return float("inf")
    dab, _ = cKDTree(b).query(a)
    dba, _ = cKDTree(a).query(b)
    return 0.5 * (dab.mean() + dba.mean())
