"""ctypes face of oracle/tl3d_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liborc.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "tl3d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class OrcCfg(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("min_depth", C.c_double), ("max_depth", C.c_double),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("origin", C.c_double * 3), ("voxel_size", C.c_double), ("sdf_trunc", C.c_double)]


class OrcIcpParams(C.Structure):
    _fields_ = [("iters", C.c_int32), ("stride", C.c_int32),
                ("max_dist", C.c_double), ("damping", C.c_double), ("eps", C.c_double), ("eig_rel", C.c_double),
                ("estimate_scale", C.c_int32), ("pad", C.c_int32)]


class OrcIcpResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("fitness", C.c_double), ("rmse", C.c_double),
                ("n_corr", C.c_int64), ("n_src", C.c_int64), ("iters_run", C.c_int32), ("status", C.c_int32),
                ("scale", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_backproject.restype = C.c_int64
        _lib.orc_extract.restype = C.c_int64
        _lib.orc_vox_index.restype = C.c_size_t
        _lib.orc_threads.restype = C.c_int
    return _lib


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (containers show every core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _p(a, ty=C.c_void_p):
    return None if a is None else a.ctypes.data_as(ty)


def _d9(r):
    return np.ascontiguousarray(np.asarray(r, dtype=np.float64).reshape(9))


def _d3(t):
    return np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(3))


class Oracle:
    """Holds the geometry (frame size, intrinsics, grid) and the numpy grids the C functions update."""

    def __init__(self, width, height, fx, fy, cx, cy, min_depth=0.1, max_depth=50.0,
                 dims=(0, 0, 0), origin=(0.0, 0.0, 0.0), voxel_size=0.005, sdf_trunc=0.02):
        self.cfg = OrcCfg(width, height, fx, fy, cx, cy, min_depth, max_depth,
                          dims[0], dims[1], dims[2], (C.c_double * 3)(*origin), voxel_size, sdf_trunc)
        self.nvox = int(dims[0]) * int(dims[1]) * int(dims[2])
        self.tsdf = np.zeros((self.nvox, 2), dtype=np.int32) if self.nvox else None
        self.centroid = np.zeros((self.nvox, 4), dtype=np.uint64) if self.nvox else None
        self.n_acc = C.c_uint64(0)
        self.n_drop = C.c_uint64(0)

    @property
    def threads(self):
        return lib().orc_threads()

    @staticmethod
    def set_threads(n: int):
        lib().orc_set_threads(C.c_int(int(n)))

    def _depth(self, depth):
        d = np.ascontiguousarray(depth, dtype=np.float32)
        assert d.shape == (self.cfg.height, self.cfg.width), d.shape
        return d

    def tsdf_integrate(self, depth, R, t, scale=1.0):
        d = self._depth(depth)
        lib().orc_tsdf_integrate(C.byref(self.cfg), _p(d), _p(_d9(R)), _p(_d3(t)), C.c_double(scale), _p(self.tsdf))

    def backproject(self, depth, bgr, R=None, t=None, scale=1.0, flags=0, subsample=1, min_depth=None, max_depth=None):
        d = self._depth(depth)
        bgr = None if bgr is None else np.ascontiguousarray(bgr, dtype=np.uint8)
        if R is None:
            flags |= 2
            R, t = np.eye(3), np.zeros(3)
        cap = ((self.cfg.height + subsample - 1) // subsample) * ((self.cfg.width + subsample - 1) // subsample)
        xyz = np.empty((cap, 3), np.float32)
        rgb = np.empty((cap, 3), np.uint8)
        n = lib().orc_backproject(C.byref(self.cfg), _p(d), _p(bgr), _p(_d9(R)), _p(_d3(t)), C.c_double(scale),
                                  C.c_uint32(flags), C.c_int(subsample),
                                  C.c_double(self.cfg.min_depth if min_depth is None else min_depth),
                                  C.c_double(self.cfg.max_depth if max_depth is None else max_depth), _p(xyz), _p(rgb))
        return xyz[:n].copy(), rgb[:n].copy()

    def centroid_accumulate(self, depth, bgr, R=None, t=None, scale=1.0, flags=0, subsample=1,
                            min_depth=None, max_depth=None):
        d = self._depth(depth)
        bgr = None if bgr is None else np.ascontiguousarray(bgr, dtype=np.uint8)
        if R is None:
            flags |= 2
            R, t = np.eye(3), np.zeros(3)
        lib().orc_centroid_accumulate(C.byref(self.cfg), _p(d), _p(bgr), _p(_d9(R)), _p(_d3(t)), C.c_double(scale),
                                      C.c_uint32(flags), C.c_int(subsample),
                                      C.c_double(self.cfg.min_depth if min_depth is None else min_depth),
                                      C.c_double(self.cfg.max_depth if max_depth is None else max_depth),
                                      _p(self.centroid), C.byref(self.n_acc), C.byref(self.n_drop))

    def centroid_points(self, xyz, rgb):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        lib().orc_centroid_points(C.byref(self.cfg), _p(xyz), _p(rgb), C.c_int64(len(xyz)), _p(self.centroid),
                                  C.byref(self.n_acc), C.byref(self.n_drop))

    def extract(self, mode=0, min_count=1, min_weight=0, max_abs_tsdf=1.0, use_tsdf=True, use_centroid=True):
        cap = self.nvox * (1 if mode == 0 else 3)
        cap = min(cap, 1 << 26)
        xyz = np.empty((cap, 3), np.float32)
        rgb = np.empty((cap, 3), np.uint8)
        n = lib().orc_extract(C.byref(self.cfg), C.c_int(mode), C.c_int(min_count), C.c_int(min_weight),
                              C.c_double(max_abs_tsdf), _p(self.tsdf if use_tsdf else None),
                              _p(self.centroid if use_centroid else None), _p(xyz), _p(rgb), C.c_int64(cap))
        assert 0 <= n <= cap, n
        return xyz[:n].copy(), rgb[:n].copy()

    def normals(self, depth, scale=1.0, depth_jump=0.05):
        d = self._depth(depth)
        nmap = np.empty((self.cfg.height, self.cfg.width, 4), np.float32)
        lib().orc_normals(C.byref(self.cfg), _p(d), C.c_double(scale), C.c_double(depth_jump), _p(nmap))
        return nmap

    def normals_smooth(self, depth, radius=2, scale=1.0, depth_jump=0.05):
        """(smoothed depth [H,W] in the units of `depth`, nmap [H,W,4]): harmonic window mean (mean of 1/z over the centre and the symmetric pixel pairs that are valid and within depth_jump of the
        centre, normals from it with a `radius`-pixel step (orc_normals_smooth)."""
        d = self._depth(depth)
        sd = np.empty((self.cfg.height, self.cfg.width), np.float32)
        nmap = np.empty((self.cfg.height, self.cfg.width, 4), np.float32)
        lib().orc_normals_smooth(C.byref(self.cfg), _p(d), C.c_double(scale), C.c_double(depth_jump), C.c_int(int(radius)), _p(sd), _p(nmap))
        return sd, nmap

    def icp(self, depth_src, nmap_tgt, T_init=None, iters=10, stride=4, max_dist=0.05, damping=1e-6, eps=1e-9,
            scale_src=1.0, eig_rel=1e-4, estimate_scale=False):
        d = self._depth(depth_src)
        nm = np.ascontiguousarray(nmap_tgt, dtype=np.float32)
        T0 = np.ascontiguousarray(np.eye(4) if T_init is None else np.asarray(T_init, np.float64).reshape(4, 4))
        prm = OrcIcpParams(iters, stride, max_dist, damping, eps, eig_rel, 1 if estimate_scale else 0, 0)
        res = OrcIcpResult()
        lib().orc_icp(C.byref(self.cfg), _p(d), C.c_double(scale_src), _p(nm), _p(T0), C.byref(prm), C.byref(res))
        return dict(T=np.array(res.T).reshape(4, 4), fitness=res.fitness, rmse=res.rmse, n_corr=res.n_corr,
                    n_src=res.n_src, iters_run=res.iters_run, status=res.status, scale=res.scale)

    def icp_sums(self, depth_src, nmap_tgt, T, stride=4, max_dist=0.05, scale_src=1.0):
        d = self._depth(depth_src)
        nm = np.ascontiguousarray(nmap_tgt, dtype=np.float32)
        T = np.ascontiguousarray(np.asarray(T, np.float64).reshape(16))
        out = np.zeros(29, np.float64)
        cnt, nsrc = C.c_int64(0), C.c_int64(0)
        lib().orc_icp_sums(C.byref(self.cfg), _p(d), C.c_double(scale_src), _p(nm), _p(T), C.c_int(stride),
                           C.c_double(max_dist), _p(out), C.byref(cnt), C.byref(nsrc))
        return out, cnt.value, nsrc.value

    def icp_sums_scale(self, depth_src, nmap_tgt, T, stride=4, max_dist=0.05, scale_src=1.0):
        """icp_sums with the Sim(3) column: 37 doubles = 21 + 6 + e + 0 + c[6] + cc + bc"""
        d = self._depth(depth_src)
        nm = np.ascontiguousarray(nmap_tgt, dtype=np.float32)
        T = np.ascontiguousarray(np.asarray(T, np.float64).reshape(16))
        out = np.zeros(37, np.float64)
        cnt, nsrc = C.c_int64(0), C.c_int64(0)
        lib().orc_icp_sums_scale(C.byref(self.cfg), _p(d), C.c_double(scale_src), _p(nm), _p(T), C.c_int(stride),
                                 C.c_double(max_dist), _p(out), C.byref(cnt), C.byref(nsrc))
        return out, cnt.value, nsrc.value

    def vox_index(self, i, j, k):
        return lib().orc_vox_index(C.byref(self.cfg), C.c_int(i), C.c_int(j), C.c_int(k))
