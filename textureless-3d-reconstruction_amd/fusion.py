"""FusionContext: one GPU's device-resident pipeline state (frame slots + grids) behind the C-ABI.

This is the object the reference-shaped classes in dense.py / pipeline.py drive.  It owns no numerics:
every method is one call into libtl3d.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _cabi as abi


@dataclass
class GridSpec:
    dims: Tuple[int, int, int]
    origin: Tuple[float, float, float]
    voxel_size: float = 0.005          # depth_to_reconstruction.py:64
    sdf_trunc: float = 0.02            # 4 voxels at the reference voxel size
    channels: int = abi.CH_TSDF | abi.CH_CENTROID
    # SPARSE grid: the channel keeps records for at most this many 8^3 bricks (4 KB / 16 KB each), handed out on first touch
    # through a brick table; 0 = dense.  dims may then describe a volume far larger than memory (what the reference's hash-map
    # merge gives for free, D2R:404-410).
    pool_tsdf: int = 0
    pool_centroid: int = 0
    # the grid is a BLOCK of a larger voxel lattice that starts at `origin`: index of its voxel (0, 0, 0) in that lattice (multiples of 8;
    # centroid channel only -- tl3d.h: tl3d_config.voxel_offset)
    voxel_offset: Tuple[int, int, int] = (0, 0, 0)

    @property
    def sparse(self) -> bool:
        return bool(self.pool_tsdf or self.pool_centroid)

    def device_bytes(self) -> int:
        """HBM the grid takes: record pools + brick tables + free-space counters."""
        nbr = self.nvox // 512
        t = (min(self.pool_tsdf, nbr) if self.pool_tsdf else nbr) * 4096 if self.channels & abi.CH_TSDF else 0
        c = (min(self.pool_centroid, nbr) if self.pool_centroid else nbr) * 16384 if self.channels & abi.CH_CENTROID else 0
        return t + c + 12 * nbr

    @property
    def nvox(self) -> int:
        return int(self.dims[0]) * int(self.dims[1]) * int(self.dims[2])

    @staticmethod
    def cube(n: int, voxel_size: float, centre=(0.0, 0.0, 0.0), sdf_trunc: Optional[float] = None,
             channels: int = abi.CH_TSDF | abi.CH_CENTROID) -> "GridSpec":
        half = 0.5 * n * voxel_size
        return GridSpec((n, n, n), tuple(float(c) - half for c in centre), voxel_size,
                        4.0 * voxel_size if sdf_trunc is None else sdf_trunc, channels)


class PinnedArray:
    """A numpy array in page-locked host memory (tl3d_pinned_alloc): the source of asynchronous uploads."""

    def __init__(self, shape, dtype):
        self.lib = abi.load()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        abi.check(self.lib.tl3d_pinned_alloc(self.nbytes, C.byref(p)))
        self._p = p
        buf = (C.c_char * self.nbytes).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self._p is not None:
            self.array = None
            self.lib.tl3d_pinned_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedCache:
    """Page-locked staging buffers kept between uses.  Locking and unlocking host memory costs ~1.5 ms per 10 MB each way (measured:
    a ring of 32 x 18.6 MB staging buffers 51 ms to set up, 74 ms to tear down -- a third of a 384-frame file-fed run), so the frame
    prefetcher takes its buffers from here and hands them back; at most `limit_bytes` stay cached, the rest is freed at once, and
    everything is freed when the interpreter exits.  Thread-safe (the decode workers take buffers side by side)."""
    limit_bytes = 1 << 30
    _free = {}
    _cached = 0
    _lock = None

    @classmethod
    def _lk(cls):
        if cls._lock is None:
            import atexit
            import threading
            cls._lock = threading.Lock()
            atexit.register(cls.clear)
        return cls._lock

    @classmethod
    def take(cls, shape, dtype) -> PinnedArray:
        key = (tuple(int(x) for x in shape), np.dtype(dtype).str)
        with cls._lk():
            lst = cls._free.get(key)
            if lst:
                pa = lst.pop()
                cls._cached -= pa.nbytes
                return pa
        return PinnedArray(shape, dtype)

    @classmethod
    def give(cls, pa: PinnedArray):
        if pa is None or pa._p is None:
            return
        key = (tuple(pa.array.shape), pa.array.dtype.str)
        with cls._lk():
            if cls._cached + pa.nbytes <= cls.limit_bytes:
                cls._free.setdefault(key, []).append(pa)
                cls._cached += pa.nbytes
                return
        pa.free()

    @classmethod
    def clear(cls):
        with cls._lk():
            lists, cls._free, cls._cached = list(cls._free.values()), {}, 0
        for lst in lists:
            for pa in lst:
                try:
                    pa.free()
                except Exception:
                    pass


def release_cached_memory():
    """Return what the library keeps between contexts to the system: the device frame slabs of destroyed contexts
    (tl3d_release_cached_memory) and the page-locked staging buffers of closed prefetchers (PinnedCache)."""
    PinnedCache.clear()
    abi.check(abi.load().tl3d_release_cached_memory())


# numpy images of tl3d_icp_pair / tl3d_icp_result (include/tl3d.h; sizes checked against the ctypes structures at import)
_ICP_PAIR_DT = np.dtype([("slot_src", "<i4"), ("slot_tgt", "<i4"), ("scale_src", "<f8"), ("T_init", "<f8", (16,))])
_ICP_RESULT_DT = np.dtype([("T", "<f8", (16,)), ("fitness", "<f8"), ("rmse", "<f8"), ("n_corr", "<i8"), ("n_src", "<i8"),
                           ("iters_run", "<i4"), ("status", "<i4"), ("scale", "<f8")])
assert _ICP_PAIR_DT.itemsize == C.sizeof(abi.IcpPair) and _ICP_RESULT_DT.itemsize == C.sizeof(abi.IcpResult)


class FusionContext:
    def __init__(self, width: int, height: int, fx: float, fy: float, cx: float, cy: float,
                 min_depth: float = 0.1, max_depth: float = 50.0, n_slots: int = 2,
                 grid: Optional[GridSpec] = None, device: int = 0, ext_tsdf=None, ext_centroid=None, stream=None):
        self._h = None
        lib = abi.load()
        if abi.device_count() <= 0:
            raise RuntimeError("libtl3d: no HIP device visible; the MI355X path has no CPU fallback")
        cfg = abi.Config()
        cfg.abi_version = abi.ABI_VERSION
        cfg.width, cfg.height = int(width), int(height)
        cfg.fx, cfg.fy, cfg.cx, cfg.cy = float(fx), float(fy), float(cx), float(cy)
        cfg.min_depth, cfg.max_depth = float(min_depth), float(max_depth)
        cfg.n_slots = int(n_slots)
        if grid is not None:
            cfg.channels = int(grid.channels)
            cfg.nx, cfg.ny, cfg.nz = (int(d) for d in grid.dims)
            cfg.origin = (C.c_double * 3)(*[float(o) for o in grid.origin])
            cfg.voxel_size = float(grid.voxel_size)
            cfg.sdf_trunc = float(grid.sdf_trunc)
            cfg.pool_bricks_tsdf, cfg.pool_bricks_centroid = int(grid.pool_tsdf), int(grid.pool_centroid)
            cfg.voxel_offset = (C.c_int64 * 3)(*[int(o) for o in grid.voxel_offset])
        cfg.ext_tsdf = abi.ptr(ext_tsdf)
        cfg.ext_centroid = abi.ptr(ext_centroid)
        cfg.stream = abi.ptr(stream)
        self._keep = (ext_tsdf, ext_centroid)
        h = C.c_void_p()
        abi.check(lib.tl3d_create(C.byref(cfg), int(device), C.byref(h)))
        self._h, self._lib = h, lib
        self.width, self.height = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.grid = grid
        self.device = int(device)
        self.n_slots = int(n_slots)

    # ---- lifetime --------------------------------------------------------------------------
    def close(self):
        if self._h is not None:
            self._lib.tl3d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        abi.check(self._lib.tl3d_sync(self._h))

    def stream_ptr(self) -> int:
        """The hipStream_t the context enqueues on, as an integer (torch.cuda.ExternalStream takes it)."""
        p = C.c_void_p()
        abi.check(self._lib.tl3d_get_stream(self._h, C.byref(p)))
        return int(p.value or 0)

    # ---- frames ----------------------------------------------------------------------------
    def upload(self, slot: int, depth, bgr=None):
        """depth: float32 [H,W] metres / relative units, or uint16 [H,W] millimetres (16-bit PNG, D2R:85-90)."""
        if isinstance(depth, np.ndarray):
            if depth.dtype == np.uint16:
                kind = abi.DEPTH_U16_MM
            else:
                depth = np.ascontiguousarray(depth, dtype=np.float32)
                kind = abi.DEPTH_F32_M
            depth = np.ascontiguousarray(depth)
            assert depth.shape == (self.height, self.width), f"depth {depth.shape} != {(self.height, self.width)}"
        else:                                   # torch tensor (host or device)
            import torch
            kind = abi.DEPTH_U16_MM if depth.dtype in (torch.uint16, torch.int16) else abi.DEPTH_F32_M
            assert tuple(depth.shape) == (self.height, self.width)
        if bgr is not None:
            if isinstance(bgr, np.ndarray):
                bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
            assert tuple(bgr.shape) == (self.height, self.width, 3), f"colour {tuple(bgr.shape)}"
        abi.check(self._lib.tl3d_upload_frame(self._h, int(slot), abi.ptr(depth), kind, abi.ptr(bgr)))

    def upload_async(self, slot: int, depth, bgr=None):
        """Enqueue the copies and return; `depth` / `bgr` (ideally pinned, see pinned_array) must stay untouched until
        slot_wait(slot).  Device work that reads the slot is ordered after the copy automatically."""
        kind = abi.DEPTH_U16_MM if depth.dtype == np.uint16 else abi.DEPTH_F32_M
        assert depth.flags["C_CONTIGUOUS"] and depth.shape == (self.height, self.width)
        assert bgr is None or (bgr.flags["C_CONTIGUOUS"] and bgr.shape == (self.height, self.width, 3) and bgr.dtype == np.uint8)
        abi.check(self._lib.tl3d_upload_frame_async(self._h, int(slot), abi.ptr(depth), kind, abi.ptr(bgr)))

    def slot_wait(self, slot: int):
        abi.check(self._lib.tl3d_slot_wait(self._h, int(slot)))

    def attach_grid(self, grid: "GridSpec", ext_tsdf=None, ext_centroid=None):
        """Give a grid-less context its fusion grid (frames stay resident across registration and fusion)."""
        cfg = abi.Config()
        cfg.abi_version = abi.ABI_VERSION
        cfg.channels = int(grid.channels)
        cfg.nx, cfg.ny, cfg.nz = (int(d) for d in grid.dims)
        cfg.origin = (C.c_double * 3)(*[float(o) for o in grid.origin])
        cfg.voxel_size, cfg.sdf_trunc = float(grid.voxel_size), float(grid.sdf_trunc)
        cfg.pool_bricks_tsdf, cfg.pool_bricks_centroid = int(grid.pool_tsdf), int(grid.pool_centroid)
        cfg.voxel_offset = (C.c_int64 * 3)(*[int(o) for o in grid.voxel_offset])
        cfg.ext_tsdf, cfg.ext_centroid = abi.ptr(ext_tsdf), abi.ptr(ext_centroid)
        abi.check(self._lib.tl3d_attach_grid(self._h, C.byref(cfg)))
        self._keep = (ext_tsdf, ext_centroid)
        self.grid = grid

    def download_depth(self, slot: int) -> np.ndarray:
        out = np.empty((self.height, self.width), np.float32)
        abi.check(self._lib.tl3d_download_depth(self._h, int(slot), abi.ptr(out)))
        return out

    # ---- a4/a5 -----------------------------------------------------------------------------
    @staticmethod
    def _pose_args(pose, flags):
        if pose is None:
            return abi.d9(np.eye(3)), abi.d3(np.zeros(3)), flags | abi.F_NO_POSE
        r, t = pose
        return abi.d9(r), abi.d3(t), flags

    def backproject(self, slot: int, pose=None, scale=1.0, subsample: int = 1, min_depth=None, max_depth=None,
                    scale_f64: bool = False):
        r, t, flags = self._pose_args(pose, abi.F_SCALE_F64 if scale_f64 else 0)
        mn = self.min_depth if min_depth is None else float(min_depth)
        mx = self.max_depth if max_depth is None else float(max_depth)
        cap = -(-self.height // subsample) * -(-self.width // subsample)
        xyz = np.empty((cap, 3), np.float32)
        rgb = np.empty((cap, 3), np.uint8)
        n = C.c_int64(0)
        abi.check(self._lib.tl3d_backproject(self._h, int(slot), abi.ptr(r), abi.ptr(t), float(scale), flags,
                                             int(subsample), mn, mx, abi.ptr(xyz), abi.ptr(rgb), cap, C.byref(n)))
        return xyz[:n.value], rgb[:n.value]

    def backproject_device(self, slot: int, out_xyz, out_rgb, out_n, pose=None, scale=1.0, subsample: int = 1,
                           min_depth=None, max_depth=None, scale_f64: bool = False, cap=None):
        """Asynchronous, device-only form: out_xyz (float32 [cap,3]), out_rgb (uint8 [cap,3]) and out_n (int64 [1]) are
        device tensors (anything with data_ptr()); nothing is read back, the call returns once its kernel is enqueued."""
        r, t, flags = self._pose_args(pose, abi.F_SCALE_F64 if scale_f64 else 0)
        mn = self.min_depth if min_depth is None else float(min_depth)
        mx = self.max_depth if max_depth is None else float(max_depth)
        cap = int(out_xyz.shape[0]) if cap is None else int(cap)
        abi.check(self._lib.tl3d_backproject_device(self._h, int(slot), abi.ptr(r), abi.ptr(t), float(scale), flags, int(subsample),
                                                    mn, mx, abi.ptr(out_xyz), abi.ptr(out_rgb), cap, abi.ptr(out_n)))

    def frame_bounds(self, slot: int, pose=None, scale=1.0, subsample: int = 1, min_depth=None, max_depth=None, scale_f64: bool = False):
        """(min[3], max[3]) of the points backproject() would return, computed on the device (+-inf when there are none)."""
        r, t, flags = self._pose_args(pose, abi.F_SCALE_F64 if scale_f64 else 0)
        mn_d = self.min_depth if min_depth is None else float(min_depth)
        mx_d = self.max_depth if max_depth is None else float(max_depth)
        lo, hi = np.zeros(3), np.zeros(3)
        abi.check(self._lib.tl3d_frame_bounds(self._h, int(slot), abi.ptr(r), abi.ptr(t), float(scale), flags, int(subsample), mn_d, mx_d,
                                              abi.ptr(lo), abi.ptr(hi), None))
        return lo, hi

    def frames_bounds(self, slots, poses, scales=None, subsample: int = 1, min_depth=None, max_depth=None):
        """(min[3], max[3]) over the clouds of many frames (poses: one (R, t) per frame), one read-back per 16 frames."""
        n = len(slots)
        sl = np.ascontiguousarray(slots, np.int32)
        R = np.ascontiguousarray(np.stack([np.asarray(p[0], np.float64).reshape(3, 3) for p in poses]))
        t = np.ascontiguousarray(np.stack([np.asarray(p[1], np.float64).reshape(3) for p in poses]))
        sc = np.ascontiguousarray(np.ones(n) if scales is None else np.asarray(scales, np.float64))
        mn_d = self.min_depth if min_depth is None else float(min_depth)
        mx_d = self.max_depth if max_depth is None else float(max_depth)
        lo, hi = np.zeros(3), np.zeros(3)
        abi.check(self._lib.tl3d_frames_bounds(self._h, n, abi.ptr(sl), abi.ptr(R), abi.ptr(t), abi.ptr(sc), 0, int(subsample), mn_d, mx_d,
                                               abi.ptr(lo), abi.ptr(hi)))
        return lo, hi

    def count_bricks(self, grid: "GridSpec", slots, poses, scales=None, centroid_subsample: int = 1, min_depth=None, max_depth=None):
        """(TSDF bricks, centroid bricks) a fusion of these frames into `grid` would give records to -- geometry only, nothing is
        allocated or written: what a sparse grid's pools must hold (tl3d.h: tl3d_count_bricks)."""
        n = len(slots)
        cfg = abi.Config()
        cfg.abi_version = abi.ABI_VERSION
        cfg.channels = int(grid.channels)
        cfg.nx, cfg.ny, cfg.nz = (int(d) for d in grid.dims)
        cfg.origin = (C.c_double * 3)(*[float(o) for o in grid.origin])
        cfg.voxel_size, cfg.sdf_trunc = float(grid.voxel_size), float(grid.sdf_trunc)
        sl = np.ascontiguousarray(slots, np.int32)
        R = np.ascontiguousarray(np.stack([np.asarray(p[0], np.float64).reshape(3, 3) for p in poses]))
        t = np.ascontiguousarray(np.stack([np.asarray(p[1], np.float64).reshape(3) for p in poses]))
        sc = np.ascontiguousarray(np.ones(n) if scales is None else np.asarray(scales, np.float64))
        mn_d = self.min_depth if min_depth is None else float(min_depth)
        mx_d = self.max_depth if max_depth is None else float(max_depth)
        nt, nc = C.c_int64(0), C.c_int64(0)
        abi.check(self._lib.tl3d_count_bricks(self._h, C.byref(cfg), n, abi.ptr(sl), abi.ptr(R), abi.ptr(t), abi.ptr(sc), int(centroid_subsample),
                                              mn_d, mx_d, C.byref(nt), C.byref(nc)))
        return int(nt.value), int(nc.value)

    # ---- fusion ----------------------------------------------------------------------------
    def accumulate_centroid(self, slot: int, pose=None, scale=1.0, subsample: int = 1, min_depth=None, max_depth=None,
                            scale_f64: bool = False):
        r, t, flags = self._pose_args(pose, abi.F_SCALE_F64 if scale_f64 else 0)
        mn = self.min_depth if min_depth is None else float(min_depth)
        mx = self.max_depth if max_depth is None else float(max_depth)
        abi.check(self._lib.tl3d_accumulate_centroid(self._h, int(slot), abi.ptr(r), abi.ptr(t), float(scale), flags,
                                                     int(subsample), mn, mx))

    def accumulate_points(self, xyz, rgb):
        if isinstance(xyz, np.ndarray):
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        n = int(xyz.shape[0])
        abi.check(self._lib.tl3d_accumulate_points(self._h, abi.ptr(xyz), abi.ptr(rgb), n))

    def points_bounds(self, xyz):
        if isinstance(xyz, np.ndarray):
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        mn, mx = np.zeros(3), np.zeros(3)
        abi.check(self._lib.tl3d_points_bounds(self._h, abi.ptr(xyz), int(xyz.shape[0]), abi.ptr(mn), abi.ptr(mx)))
        return mn, mx

    def integrate(self, slot: int, pose, scale=1.0):
        r, t = pose
        abi.check(self._lib.tl3d_integrate(self._h, int(slot), abi.ptr(abi.d9(r)), abi.ptr(abi.d3(t)), float(scale)))

    def fuse_frames(self, slots, poses, scales=None, centroid_subsample: int = 0, min_depth=None, max_depth=None):
        """integrate() (and accumulate_centroid() when centroid_subsample >= 1) of many frames in one call, in order."""
        n = len(slots)
        if n == 0:
            return
        sl = np.ascontiguousarray(slots, np.int32)
        R = np.ascontiguousarray(np.stack([np.asarray(p[0], np.float64).reshape(3, 3) for p in poses]))
        t = np.ascontiguousarray(np.stack([np.asarray(p[1], np.float64).reshape(3) for p in poses]))
        sc = np.ascontiguousarray(np.ones(n) if scales is None else np.asarray(scales, np.float64))
        mn_d = self.min_depth if min_depth is None else float(min_depth)
        mx_d = self.max_depth if max_depth is None else float(max_depth)
        abi.check(self._lib.tl3d_fuse_frames(self._h, n, abi.ptr(sl), abi.ptr(R), abi.ptr(t), abi.ptr(sc), 0, int(centroid_subsample), mn_d, mx_d))

    def fuse_frames_packed(self, slots_i32, R_f64, t_f64, scales_f64=None, centroid_subsample: int = 0):
        """fuse_frames() on arrays the caller has already packed (int32 [n], float64 [n,9], float64 [n,3], float64 [n]): a
        long-running loop pays the packing once."""
        n = int(slots_i32.shape[0])
        if n == 0:
            return
        sc = scales_f64 if scales_f64 is not None else np.ones(n)
        abi.check(self._lib.tl3d_fuse_frames(self._h, n, abi.ptr(slots_i32), abi.ptr(R_f64), abi.ptr(t_f64), abi.ptr(sc), 0,
                                             int(centroid_subsample), self.min_depth, self.max_depth))

    # ---- ICP -------------------------------------------------------------------------------
    def build_normals_many(self, slots, scales=None, depth_jump=0.05):
        n = len(slots)
        if n == 0:
            return
        sl = np.ascontiguousarray(slots, np.int32)
        sc = None if scales is None else np.ascontiguousarray(np.asarray(scales, np.float64))
        abi.check(self._lib.tl3d_build_normals_many(self._h, n, abi.ptr(sl), None if sc is None else abi.ptr(sc), float(depth_jump)))

    def set_normal_smoothing(self, radius: int):
        """Normal maps built from now on use the window-averaged depth (radius 1 = 3 x 3), and registrations read it as their
        source depth: robust to depth noise (tl3d.h: tl3d_set_normal_smoothing); 0 = off."""
        abi.check(self._lib.tl3d_set_normal_smoothing(self._h, int(radius)))

    def build_normals(self, slot: int, scale=1.0, depth_jump=0.05):
        abi.check(self._lib.tl3d_build_normals(self._h, int(slot), float(scale), float(depth_jump)))

    def download_normals(self, slot: int, out=None) -> np.ndarray:
        """The slot's normal map as a row-major [H][W][4] image (nx, ny, nz, depth); out: a host array or a device tensor of that
        shape to fill instead (the library keeps the map in phase-major rows and converts on the way out)."""
        if out is None:
            out = np.empty((self.height, self.width, 4), np.float32)
        abi.check(self._lib.tl3d_download_normals(self._h, int(slot), abi.ptr(out)))
        return out

    def icp(self, slot_src: int, slot_tgt: int, T_init=None, iters=10, stride=4, max_dist=0.05, damping=1e-6,
            eps=1e-9, scale_src=1.0, eig_rel=1e-4, estimate_scale=False):
        """Point-to-plane ICP; returns T (src camera -> tgt camera) and statistics.  With src = previous frame and
        tgt = current frame, (T[:3,:3], T[:3,3]) is (R_rel, t_rel) of depth_to_reconstruction.py:618-620."""
        T0 = np.ascontiguousarray(np.eye(4) if T_init is None else np.asarray(T_init, np.float64).reshape(4, 4))
        prm = abi.IcpParams(int(iters), int(stride), float(max_dist), float(damping), float(eps), float(eig_rel), 1 if estimate_scale else 0, 0)
        res = abi.IcpResult()
        abi.check(self._lib.tl3d_icp_p2plane(self._h, int(slot_src), float(scale_src), int(slot_tgt), abi.ptr(T0),
                                             C.byref(prm), C.byref(res)))
        return dict(T=np.array(res.T).reshape(4, 4), fitness=res.fitness, rmse=res.rmse, n_corr=res.n_corr,
                    n_src=res.n_src, iters_run=res.iters_run, status=res.status, scale=res.scale)

    def icp_enqueue(self, lane: int, slot_src: int, slot_tgt: int, T_init=None, iters=10, stride=4, max_dist=0.05,
                    damping=1e-6, eps=1e-9, scale_src=1.0, eig_rel=1e-4, estimate_scale=False):
        """Asynchronous form: up to abi.ICP_LANES independent registrations in flight (one per lane)."""
        T0 = np.ascontiguousarray(np.eye(4) if T_init is None else np.asarray(T_init, np.float64).reshape(4, 4))
        prm = abi.IcpParams(int(iters), int(stride), float(max_dist), float(damping), float(eps), float(eig_rel), 1 if estimate_scale else 0, 0)
        abi.check(self._lib.tl3d_icp_enqueue(self._h, int(lane), int(slot_src), float(scale_src), int(slot_tgt), abi.ptr(T0),
                                             C.byref(prm)))

    def icp_collect(self, lane: int):
        res = abi.IcpResult()
        abi.check(self._lib.tl3d_icp_collect(self._h, int(lane), C.byref(res)))
        return dict(T=np.array(res.T).reshape(4, 4), fitness=res.fitness, rmse=res.rmse, n_corr=res.n_corr,
                    n_src=res.n_src, iters_run=res.iters_run, status=res.status, scale=res.scale)

    def icp_batch_enqueue(self, pairs, levels, T_init=None, scales=None):
        """Register every (slot_src, slot_tgt) of `pairs` through all of `levels` in ONE launch (asynchronous).

        levels: sequence of dicts with the keyword arguments of icp() (iters, stride, max_dist, damping, eps, eig_rel),
        coarse to fine.  T_init: one 4x4 per pair (default identity); scales: metric scale of each pair's source depth."""
        n = len(pairs)
        # the request as one numpy record array laid out like tl3d_icp_pair (a Python loop over ctypes fields costs ~10 us per pair:
        # as much as the registration of a pair inside a large batch)
        arr = np.zeros(n, _ICP_PAIR_DT)
        pr = np.asarray(pairs, np.int64).reshape(n, 2)
        arr["slot_src"], arr["slot_tgt"] = pr[:, 0], pr[:, 1]
        arr["scale_src"] = 1.0 if scales is None else np.asarray(scales, np.float64)
        arr["T_init"] = np.eye(4).ravel()
        if T_init is not None:
            for i, T0 in enumerate(T_init):
                if T0 is not None:
                    arr["T_init"][i] = np.asarray(T0, np.float64).reshape(16)
        lv = (abi.IcpParams * len(levels))()
        for i, kw in enumerate(levels):
            lv[i] = abi.IcpParams(int(kw.get("iters", 10)), int(kw.get("stride", 4)), float(kw.get("max_dist", 0.05)),
                                  float(kw.get("damping", 1e-6)), float(kw.get("eps", 1e-9)), float(kw.get("eig_rel", 1e-4)),
                                  1 if kw.get("estimate_scale", False) else 0, 0)
        abi.check(self._lib.tl3d_icp_batch_enqueue(self._h, arr.ctypes.data_as(C.POINTER(abi.IcpPair)), n, lv, len(levels)))
        self._icp_batch_n = n

    def icp_batch_collect(self):
        n = getattr(self, "_icp_batch_n", 0)
        res = np.zeros(max(1, n), _ICP_RESULT_DT)
        abi.check(self._lib.tl3d_icp_batch_collect(self._h, res.ctypes.data_as(C.POINTER(abi.IcpResult)), n))
        self._icp_batch_n = 0
        T = res["T"].reshape(-1, 4, 4)
        cols = [res[k].tolist() for k in ("fitness", "rmse", "n_corr", "n_src", "iters_run", "status", "scale")]
        return [dict(T=T[i], fitness=cols[0][i], rmse=cols[1][i], n_corr=cols[2][i], n_src=cols[3][i], iters_run=cols[4][i],
                     status=cols[5][i], scale=cols[6][i]) for i in range(n)]

    def icp_batch(self, pairs, levels, T_init=None, scales=None):
        self.icp_batch_enqueue(pairs, levels, T_init, scales)
        return self.icp_batch_collect()

    # ---- grids -----------------------------------------------------------------------------
    def reset(self):
        abi.check(self._lib.tl3d_grid_reset(self._h))

    def grid_ptr(self, channel: int):
        p, nb = C.c_void_p(), C.c_size_t()
        abi.check(self._lib.tl3d_grid_device_ptr(self._h, int(channel), C.byref(p), C.byref(nb)))
        return p.value, nb.value

    def _channel_bytes(self, channel: int) -> int:
        return self.grid.nvox * (8 if channel == abi.CH_TSDF else 32)

    def download_grid(self, channel: int) -> np.ndarray:
        """The channel as a dense array in record order (a sparse grid is gathered through its brick table: the same image)."""
        nb = self._channel_bytes(channel)
        if channel == abi.CH_TSDF:
            out = np.empty((nb // 8, 2), np.int32)
        else:
            out = np.empty((nb // 32, 4), np.uint64)
        abi.check(self._lib.tl3d_grid_download(self._h, int(channel), abi.ptr(out), nb))
        return out

    def upload_grid(self, channel: int, arr):
        nb = self._channel_bytes(channel)
        if isinstance(arr, np.ndarray):
            arr = np.ascontiguousarray(arr)
            assert arr.nbytes == nb
        abi.check(self._lib.tl3d_grid_upload(self._h, int(channel), abi.ptr(arr), nb))

    def add_grid(self, channel: int, arr):
        nb = self._channel_bytes(channel)
        if isinstance(arr, np.ndarray):
            arr = np.ascontiguousarray(arr)
            assert arr.nbytes == nb
        abi.check(self._lib.tl3d_grid_add(self._h, int(channel), abi.ptr(arr), nb))

    # ---- sparse merge helpers (device tensors: anything with data_ptr()) ---------------------------------------------
    @property
    def n_bricks(self) -> int:
        return self.grid.nvox // 512

    def touched_bricks(self, map_dev, channels: int = 0):
        """map_dev[b] |= 1 (uint8, one per brick, device memory, zeroed by the caller) for every brick that holds anything.
        With abi.CH_FREE in `channels` (and in the channel of pack_bricks / unpack_bricks) pending free-space counts stay pending
        and mark nothing: they travel on their own, as grid_tensor(abi.CH_FREE).  With abi.CH_SUB the unit is a 4x4x4 sub-brick:
        the map has 8 bytes per brick, the lists of pack_bricks / unpack_bricks hold brick * 8 + sub-brick, rows are 64 records."""
        abi.check(self._lib.tl3d_grid_touched_bricks(self._h, int(channels), abi.ptr(map_dev), int(map_dev.shape[0])))

    def pack_bricks(self, channel: int, bricks_dev, packed_dev):
        abi.check(self._lib.tl3d_grid_pack_bricks(self._h, int(channel), abi.ptr(bricks_dev), int(bricks_dev.shape[0]), abi.ptr(packed_dev)))

    def unpack_bricks(self, channel: int, bricks_dev, packed_dev):
        abi.check(self._lib.tl3d_grid_unpack_bricks(self._h, int(channel), abi.ptr(bricks_dev), int(bricks_dev.shape[0]), abi.ptr(packed_dev)))

    def max_weight(self) -> int:
        """Largest number of observations any TSDF voxel holds (int32 headroom: _cabi.TSDF_MAX_WEIGHT)."""
        w = C.c_int64(0)
        abi.check(self._lib.tl3d_grid_max_weight(self._h, C.byref(w)))
        return int(w.value)

    # ---- multi-GPU merge through the library's own RCCL binding (hosts without torch.distributed) -------------------
    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        abi.check(abi.load().tl3d_rccl_unique_id(buf))
        return bytes(buf)

    def rccl_init(self, world: int, rank: int, unique_id: bytes):
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        abi.check(self._lib.tl3d_rccl_init(self._h, int(world), int(rank), buf))

    def allreduce_grid(self, channels: int = 0):
        abi.check(self._lib.tl3d_allreduce_grid(self._h, int(channels)))

    def grid_tensor(self, channel: int):
        """Zero-copy torch view of a grid channel (for torch.distributed all_reduce over RCCL)."""
        import torch
        p, nb = self.grid_ptr(channel)
        dt, item, typestr = (torch.int32, 4, "<i4") if channel in (abi.CH_TSDF, abi.CH_FREE) else (torch.int64, 8, "<i8")

        class _Iface:
            __cuda_array_interface__ = {"shape": (nb // item,), "typestr": typestr, "data": (p, False), "version": 2}
        t = torch.as_tensor(_Iface(), device=torch.device("cuda", self.device))
        assert t.data_ptr() == p and t.dtype == dt
        return t

    def extract(self, mode: int = abi.EXTRACT_CENTROID, min_count: int = 1, min_weight: int = 0,
                max_abs_tsdf: float = 1.0):
        n = C.c_int64(0)
        abi.check(self._lib.tl3d_extract(self._h, int(mode), int(min_count), int(min_weight), float(max_abs_tsdf),
                                         None, None, 0, C.byref(n)))
        xyz = np.empty((n.value, 3), np.float32)
        rgb = np.empty((n.value, 3), np.uint8)
        if n.value:
            abi.check(self._lib.tl3d_extract(self._h, int(mode), int(min_count), int(min_weight), float(max_abs_tsdf),
                                             abi.ptr(xyz), abi.ptr(rgb), n.value, C.byref(n)))
        return xyz, rgb

    def statistical_outlier(self, xyz, nb_neighbors=20, std_ratio=2.0, cell_size=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        keep = np.zeros(len(xyz), np.uint8)
        kept = C.c_int64(0)
        cell = float(cell_size if cell_size is not None else (self.grid.voxel_size * 2 if self.grid else 0.01))
        abi.check(self._lib.tl3d_statistical_outlier(self._h, abi.ptr(xyz), len(xyz), int(nb_neighbors), float(std_ratio),
                                                     cell, abi.ptr(keep), C.byref(kept)))
        return keep.astype(bool)

    # ---- measurement -----------------------------------------------------------------------
    def set_profile(self, count_records=False, time_kernels=False):
        abi.check(self._lib.tl3d_set_profile(self._h, int(count_records), int(time_kernels)))

    def set_tsdf_pairing(self, on: bool):
        """integrate() may update two consecutive overlapping frames per launch (default); off: one frame per launch."""
        abi.check(self._lib.tl3d_set_tsdf_pairing(self._h, 1 if on else 0))

    def stats(self) -> dict:
        s = abi.Stats()
        abi.check(self._lib.tl3d_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in abi.Stats._fields_}

    def reset_stats(self):
        abi.check(self._lib.tl3d_reset_stats(self._h))

    def event_record(self, which: int):
        abi.check(self._lib.tl3d_event_record(self._h, int(which)))

    def event_elapsed_ms(self) -> float:
        ms = C.c_float(0)
        abi.check(self._lib.tl3d_event_elapsed_ms(self._h, C.byref(ms)))
        return ms.value
