"""The dense back end with the reference's call shapes, running on the MI355X.

    DenseReconstructor            depth_to_reconstruction.py:274-420
    DensePointCloudGenerator      depth_enhanced_reconstruction.py:535-645
    DepthScaleEstimator           depth_enhanced_reconstruction.py:652-697

Same method names, argument meaning, return types and corner-case behaviour, so the body of the
reference's reconstruct() could call these unchanged.  All per-pixel / per-point work happens in
libtl3d.so (back-projection, voxel accumulation, extraction, outlier filter); there is no CPU
fallback: without the library or without a GPU the constructor's first device call raises.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from . import _cabi as abi
from .config import CameraIntrinsics, ReconstructionConfig
from .fusion import FusionContext, GridSpec

MERGE_GRID_BUDGET_BYTES = 96 << 30      # dense centroid grid the merge may allocate (HBM is 288 GB)


def _is_f64_scalar(scale) -> bool:
    """numpy-2 promotion: depth(float32) * np.float64 -> float64, * Python float -> float32
    (depth_to_reconstruction.py:356 with :323)."""
    return isinstance(scale, np.floating) and not isinstance(scale, (np.float32, np.float16))


def estimate_scale_d2r(sparse_points, sparse_pts2d, depth_map) -> float:
    """DenseReconstructor.estimate_scale (D2R:297-326): median of Z_sparse / depth[int(y), int(x)], ratios outside
    (0.001, 1000) dropped, fewer than 3 samples -> 1.0.  n is 10^2..10^3 points: host work."""
    h, w = depth_map.shape
    scales = []
    for p3, p2 in zip(sparse_points, sparse_pts2d):
        px, py = int(p2[0]), int(p2[1])
        if 0 <= px < w and 0 <= py < h:
            dn, ds = depth_map[py, px], p3[2]
            if dn > 0 and ds > 0:
                s = ds / dn
                if 0.001 < s < 1000:
                    scales.append(s)
    if len(scales) < 3:
        print("Warning: Too few scale samples, using default scale=1.0")
        return 1.0
    scale = np.median(scales)
    print(f"Estimated depth scale: {scale:.6f} (from {len(scales)} samples)")
    return scale


class _DeviceDense:
    """Shared device plumbing: one back-projection context per frame size, one fusion context per merge."""

    def __init__(self, device: int = 0):
        self._device = device
        self._bp_ctx = {}

    def _ctx_for(self, h, w, fx, fy, cx, cy) -> FusionContext:
        key = (h, w, fx, fy, cx, cy)
        ctx = self._bp_ctx.get(key)
        if ctx is None:
            for old in self._bp_ctx.values():
                old.close()
            self._bp_ctx.clear()
            ctx = FusionContext(w, h, fx, fy, cx, cy, n_slots=1, grid=None, device=self._device)
            self._bp_ctx[key] = ctx
        return ctx

    def _backproject(self, depth, color, fx, fy, cx, cy, pose, scale, subsample, min_depth, max_depth):
        depth = np.asarray(depth)
        h, w = depth.shape
        ctx = self._ctx_for(h, w, float(fx), float(fy), float(cx), float(cy))
        f64 = _is_f64_scalar(scale)
        if depth.dtype == np.float64:
            # Happens when a caller pre-multiplies float32 depth by an np.float64 scale (DER:1135 under numpy 2).
            # The device holds float32 depth: if the fp64 values are float32-representable the fp64 compare path
            # reproduces the reference exactly; otherwise they are rounded to float32 first, which moves z by at
            # most half a float32 ulp (<= 3e-8 relative) -- pass (depth32, scale) separately for exact parity.
            depth32 = depth.astype(np.float32)
            f64 = f64 or bool(np.array_equal(depth32.astype(np.float64), depth, equal_nan=True))
            depth = depth32
        ctx.upload(0, np.ascontiguousarray(depth, dtype=np.float32), None if color is None else color)
        pts, col = ctx.backproject(0, pose=pose, scale=float(scale), subsample=int(subsample), min_depth=min_depth,
                                   max_depth=max_depth, scale_f64=f64)
        return pts.copy(), col.copy()

    def _merge(self, clouds, voxel_size, sor, nb_neighbors=20, std_ratio=2.0):
        pts = [np.asarray(p) for p, c in clouds if len(p) > 0]
        col = [np.asarray(c) for p, c in clouds if len(p) > 0]
        if not pts:
            return np.array([]), np.array([])                       # (0,) fp64, as D2R:398-399
        if not voxel_size > 0:
            return np.vstack(pts), np.vstack(col)
        v = float(voxel_size)
        # pass 1: bounds (Open3D: voxel origin = min_bound - voxel/2)
        boot = FusionContext(8, 8, 1.0, 1.0, 0.0, 0.0, n_slots=1, grid=None, device=self._device)
        try:
            mn = np.full(3, np.inf)
            mx = np.full(3, -np.inf)
            for p in pts:
                a, b = boot.points_bounds(np.ascontiguousarray(p, np.float32))
                mn, mx = np.minimum(mn, a), np.maximum(mx, b)
        finally:
            boot.close()
        origin = mn - 0.5 * v
        dims = np.floor((mx - origin) / v).astype(np.int64) + 1
        dims = ((dims + 7) // 8) * 8
        # Open3D's hash map holds the occupied voxels only, whatever the extent (D2R:404-410; the reference's defaults reach
        # 50 m at 5 mm: D2R:57, :64).  Here a grid's brick table is direct-indexed: up to 2^32 voxels per grid.  A larger lattice is
        # fused BLOCK BY BLOCK: blocks of at most 2^31 voxels (multiples of 8 per axis) of the one lattice that starts at `origin`
        # (tl3d_config.voxel_offset: indices are computed against `origin` and the block's offset subtracted, so the voxels are the
        # single-grid voxels exactly); every point list goes to every block, a block keeps the points that fall into it.
        blocks = _lattice_blocks(dims, 1 << 31)
        n_points = int(sum(len(p) for p in pts))
        out_xyz, out_rgb = [], []
        for off, bdims in blocks:
            nvox = int(bdims[0]) * int(bdims[1]) * int(bdims[2])
            # A cloud of n points occupies at most n voxels, i.e. at most n bricks; the pool is what the points can need, capped by the
            # memory budget.
            pool = 0
            if nvox * 32 > MERGE_GRID_BUDGET_BYTES // 4:
                pool = int(min(nvox // 512, max(4096, min(n_points, MERGE_GRID_BUDGET_BYTES // 16384))))
            spec = GridSpec(tuple(int(d) for d in bdims), tuple(float(o) for o in origin), v, 4 * v, abi.CH_CENTROID, pool_centroid=pool,
                            voxel_offset=tuple(int(o) for o in off))
            with FusionContext(8, 8, 1.0, 1.0, 0.0, 0.0, n_slots=1, grid=spec, device=self._device) as ctx:
                for p, c in zip(pts, col):
                    ctx.accumulate_points(p, c)
                if pool and ctx.stats()["pool_refused"]:
                    raise MemoryError(f"merge_pointclouds: the cloud occupies more than {pool} bricks of 8^3 voxels at voxel {v} m "
                                      f"({pool * 16384 / 2**30:.0f} GiB of records): raise voxel_size")
                xyz, rgb = ctx.extract(abi.EXTRACT_CENTROID)
            out_xyz.append(xyz)
            out_rgb.append(rgb)
        xyz = np.concatenate(out_xyz) if len(out_xyz) > 1 else out_xyz[0]
        rgb = np.concatenate(out_rgb) if len(out_rgb) > 1 else out_rgb[0]
        if sor and len(xyz) > 0:
            with FusionContext(8, 8, 1.0, 1.0, 0.0, 0.0, n_slots=1, grid=None, device=self._device) as ctx:
                keep = ctx.statistical_outlier(xyz, nb_neighbors, std_ratio, cell_size=2.0 * v)
            xyz, rgb = xyz[keep], rgb[keep]
        # the reference hands back what Open3D holds: fp64 points, uint8 colours (D2R:417-418)
        return xyz.astype(np.float64), rgb

    def close(self):
        for c in self._bp_ctx.values():
            c.close()
        self._bp_ctx.clear()


def _lattice_blocks(dims, max_voxels):
    """[(offset, dims)] of blocks that tile a lattice of `dims` voxels (multiples of 8), each of at most max_voxels: the longest axis
    is halved (at a multiple of 8) until every block fits."""
    out, todo = [], [(np.zeros(3, np.int64), np.asarray(dims, np.int64))]
    while todo:
        off, d = todo.pop()
        if int(d[0]) * int(d[1]) * int(d[2]) <= max_voxels:
            out.append((off, d))
            continue
        a = int(np.argmax(d))
        h = ((int(d[a]) // 2 + 7) // 8) * 8
        lo, hi = d.copy(), d.copy()
        lo[a], hi[a] = h, d[a] - h
        off_hi = off.copy()
        off_hi[a] += h
        todo += [(off_hi, hi), (off, lo)]
    return sorted(out, key=lambda b: (int(b[0][2]), int(b[0][1]), int(b[0][0])))


class DenseReconstructor(_DeviceDense):
    """depth_to_reconstruction.py:274-420."""

    def __init__(self, config: ReconstructionConfig):
        super().__init__(getattr(config, "device", 0))
        self.config = config
        self.K = config.K

    def estimate_scale(self, sparse_points, sparse_pts2d, depth_map) -> float:
        return estimate_scale_d2r(sparse_points, sparse_pts2d, depth_map)

    def depth_to_pointcloud(self, depth, color, pose=None, scale=1.0, subsample: int = 1):
        """(points float32[N,3] world, colors uint8[N,3] RGB) in row-major pixel order (D2R:328-384)."""
        c = self.config
        return self._backproject(depth, color, c.fx, c.fy, c.cx, c.cy, pose, scale, subsample, c.min_depth, c.max_depth)

    def merge_pointclouds(self, clouds, voxel_size: float = 0.005):
        """vstack + voxel centroid + statistical outlier removal (20 neighbours, 2 sigma), D2R:386-420."""
        return self._merge(clouds, voxel_size, sor=True)


class DensePointCloudGenerator(_DeviceDense):
    """depth_enhanced_reconstruction.py:535-645 (limits are per call, no scale argument, merge has no outlier filter)."""

    def __init__(self, intrinsics: CameraIntrinsics, device: int = 0):
        super().__init__(device)
        self.K = intrinsics

    def depth_to_pointcloud(self, depth, color, pose=None, min_depth: float = 0.1, max_depth: float = 100.0,
                            subsample: int = 1):
        k = self.K
        return self._backproject(depth, color, k.fx, k.fy, k.cx, k.cy, pose, 1.0, subsample, min_depth, max_depth)

    def merge_pointclouds(self, pointclouds, voxel_size: float = 0.01):
        return self._merge(pointclouds, voxel_size, sor=False)


class DepthScaleEstimator:
    """depth_enhanced_reconstruction.py:652-697."""

    @staticmethod
    def estimate_scale(sparse_points, sparse_pts2d, depth_map, K=None) -> float:
        if len(sparse_points) < 5:
            return 1.0
        h, w = depth_map.shape
        scales = []
        for p3, p2 in zip(sparse_points, sparse_pts2d):
            px, py = int(p2[0]), int(p2[1])
            if 0 <= px < w and 0 <= py < h:
                rel, met = depth_map[py, px], p3[2]
                if rel > 0 and met > 0:
                    scales.append(met / rel)
        if len(scales) < 3:
            return 1.0
        scale = np.median(scales)
        print(f"  Depth scale: {scale:.4f} (from {len(scales)} points)")
        return scale
