"""ctypes binding of libtl3d.so (include/tl3d.h).  No CPU fallback: a missing library or a missing
GPU raises -- the product path never routes through oracle/."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtl3d.so")

ABI_VERSION = 1
OK, E_INVALID, E_HIP, E_NOMEM, E_CAPACITY, E_STATE, E_NODEVICE = 0, -1, -2, -3, -4, -5, -6
CH_TSDF, CH_CENTROID = 1, 2
DEPTH_F32_M, DEPTH_U16_MM = 0, 1
F_SCALE_F64, F_NO_POSE = 1, 2
EXTRACT_CENTROID, EXTRACT_TSDF = 0, 1
ICP_LANES = 16

# every symbol include/tl3d.h declares (checked by tests/test_cabi_symbols.py against the header text)
SYMBOLS = [
    "tl3d_last_error", "tl3d_version", "tl3d_device_count", "tl3d_create", "tl3d_destroy", "tl3d_sync",
    "tl3d_upload_frame", "tl3d_download_depth", "tl3d_pinned_alloc", "tl3d_pinned_free", "tl3d_upload_frame_async",
    "tl3d_slot_wait", "tl3d_attach_grid", "tl3d_backproject", "tl3d_accumulate_centroid",
    "tl3d_accumulate_points", "tl3d_points_bounds", "tl3d_integrate", "tl3d_build_normals",
    "tl3d_download_normals", "tl3d_icp_p2plane", "tl3d_icp_enqueue", "tl3d_icp_collect", "tl3d_grid_reset", "tl3d_grid_device_ptr",
    "tl3d_grid_download", "tl3d_grid_upload", "tl3d_grid_add", "tl3d_extract", "tl3d_statistical_outlier",
    "tl3d_set_profile", "tl3d_get_stats", "tl3d_reset_stats", "tl3d_event_record", "tl3d_event_elapsed_ms",
]


class Tl3dError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libtl3d error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("min_depth", C.c_double), ("max_depth", C.c_double),
                ("n_slots", C.c_int32), ("channels", C.c_uint32),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("origin", C.c_double * 3), ("voxel_size", C.c_double), ("sdf_trunc", C.c_double),
                ("ext_tsdf", C.c_void_p), ("ext_centroid", C.c_void_p), ("stream", C.c_void_p)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("fitness", C.c_double), ("rmse", C.c_double),
                ("n_corr", C.c_int64), ("n_src", C.c_int64), ("iters_run", C.c_int32), ("status", C.c_int32)]


class IcpParams(C.Structure):
    _fields_ = [("iters", C.c_int32), ("stride", C.c_int32),
                ("max_dist", C.c_double), ("damping", C.c_double), ("eps", C.c_double), ("eig_rel", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("tsdf_launches", C.c_uint64), ("tsdf_records_read", C.c_uint64), ("tsdf_records_written", C.c_uint64),
                ("tsdf_bricks_visited", C.c_uint64), ("tsdf_bricks_free", C.c_uint64), ("centroid_launches", C.c_uint64), ("centroid_points", C.c_uint64),
                ("centroid_dropped", C.c_uint64), ("tsdf_kernel_ms", C.c_double), ("tsdf_kernel_timed", C.c_uint64)]


_lib = None


def _preload_shared_hip_runtime():
    """Keep ONE HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so
    (same SONAME as /opt/rocm's).  If libtl3d.so pulled in /opt/rocm's copy first, a later `import torch` would load a
    second runtime into the process, and the second one intermittently finds no GPU.  Loading torch's copy first (when
    torch is installed) makes both libtl3d.so and torch resolve to the same runtime whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return None
    return libdir


def load():
    """Load libtl3d.so; raises if it has not been built (python __graft_entry__.py / csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `bash {os.path.join(_HERE, 'csrc', 'build.sh')}` "
                          "(there is no CPU fallback for the HIP path)")
    # ICP lanes, the TSDF prep stream and the main stream want to run side by side; ROCm multiplexes streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4).  Only a default: an explicit setting wins, and it has no effect if
    # the HIP runtime was already initialised by the host application.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    _preload_shared_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    lib.tl3d_last_error.restype = C.c_char_p
    vp, i32, i64, dbl, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_uint32
    sig = {
        "tl3d_version": [],
        "tl3d_device_count": [C.POINTER(C.c_int)],
        "tl3d_create": [C.POINTER(Config), i32, C.POINTER(vp)],
        "tl3d_destroy": [vp],
        "tl3d_sync": [vp],
        "tl3d_upload_frame": [vp, i32, vp, i32, vp],
        "tl3d_download_depth": [vp, i32, vp],
        "tl3d_pinned_alloc": [C.c_size_t, C.POINTER(vp)],
        "tl3d_pinned_free": [vp],
        "tl3d_upload_frame_async": [vp, i32, vp, i32, vp],
        "tl3d_slot_wait": [vp, i32],
        "tl3d_attach_grid": [vp, C.POINTER(Config)],
        "tl3d_backproject": [vp, i32, vp, vp, dbl, u32, i32, dbl, dbl, vp, vp, i64, C.POINTER(i64)],
        "tl3d_accumulate_centroid": [vp, i32, vp, vp, dbl, u32, i32, dbl, dbl],
        "tl3d_accumulate_points": [vp, vp, vp, i64],
        "tl3d_points_bounds": [vp, vp, i64, vp, vp],
        "tl3d_integrate": [vp, i32, vp, vp, dbl],
        "tl3d_build_normals": [vp, i32, dbl, dbl],
        "tl3d_download_normals": [vp, i32, vp],
        "tl3d_icp_p2plane": [vp, i32, dbl, i32, vp, C.POINTER(IcpParams), C.POINTER(IcpResult)],
        "tl3d_icp_enqueue": [vp, i32, i32, dbl, i32, vp, C.POINTER(IcpParams)],
        "tl3d_icp_collect": [vp, i32, C.POINTER(IcpResult)],
        "tl3d_grid_reset": [vp],
        "tl3d_grid_device_ptr": [vp, u32, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "tl3d_grid_download": [vp, u32, vp, C.c_size_t],
        "tl3d_grid_upload": [vp, u32, vp, C.c_size_t],
        "tl3d_grid_add": [vp, u32, vp, C.c_size_t],
        "tl3d_extract": [vp, i32, i32, i32, dbl, vp, vp, i64, C.POINTER(i64)],
        "tl3d_statistical_outlier": [vp, vp, i64, i32, dbl, dbl, vp, C.POINTER(i64)],
        "tl3d_set_profile": [vp, i32, i32],
        "tl3d_get_stats": [vp, C.POINTER(Stats)],
        "tl3d_reset_stats": [vp],
        "tl3d_event_record": [vp, i32],
        "tl3d_event_elapsed_ms": [vp, C.POINTER(C.c_float)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    if lib.tl3d_version() != ABI_VERSION:
        raise ImportError(f"libtl3d ABI {lib.tl3d_version()} != binding {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        raise Tl3dError(rc, load().tl3d_last_error().decode(errors="replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(load().tl3d_device_count(C.byref(n)))
    return n.value


def ptr(a):
    """void* of a numpy array, a torch tensor (host or device), an int address, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous(), "tensor must be contiguous"
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"cannot take the address of {type(a)}")


def d9(r):
    return np.ascontiguousarray(np.asarray(r, dtype=np.float64).reshape(9))


def d3(t):
    return np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(3))
