"""ctypes binding of libtl3d.so (include/tl3d.h).  No CPU fallback: a missing library or a missing
GPU raises -- the product path never routes through oracle/."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TL3D_LIB") or os.path.join(_HERE, "libtl3d.so")     # TL3D_LIB: a diagnostic build

ABI_VERSION = 5
OK, E_INVALID, E_HIP, E_NOMEM, E_CAPACITY, E_STATE, E_NODEVICE = 0, -1, -2, -3, -4, -5, -6
CH_TSDF, CH_CENTROID, CH_FREE, CH_SUB = 1, 2, 4, 8
DEPTH_F32_M, DEPTH_U16_MM = 0, 1
F_SCALE_F64, F_NO_POSE = 1, 2
EXTRACT_CENTROID, EXTRACT_TSDF = 0, 1
ICP_LANES = 16
TSDF_MAX_WEIGHT = 65536

# every symbol include/tl3d.h declares (checked by tests/test_cabi_symbols.py against the header text)
SYMBOLS = [
    "tl3d_last_error", "tl3d_version", "tl3d_device_count", "tl3d_runtime_info", "tl3d_probe_hw_queues", "tl3d_grid_max_weight", "tl3d_create", "tl3d_destroy", "tl3d_sync", "tl3d_get_stream", "tl3d_release_cached_memory",
    "tl3d_upload_frame", "tl3d_download_depth", "tl3d_pinned_alloc", "tl3d_pinned_free", "tl3d_upload_frame_async",
    "tl3d_slot_wait", "tl3d_attach_grid", "tl3d_backproject", "tl3d_backproject_device", "tl3d_frame_bounds", "tl3d_frames_bounds", "tl3d_count_bricks", "tl3d_accumulate_centroid",
    "tl3d_accumulate_points", "tl3d_points_bounds", "tl3d_integrate", "tl3d_build_normals",
    "tl3d_download_normals", "tl3d_icp_p2plane", "tl3d_icp_enqueue", "tl3d_icp_collect", "tl3d_icp_batch_enqueue", "tl3d_icp_batch_collect", "tl3d_host_pack_bgr_rows", "tl3d_host_copy_rows", "tl3d_build_normals_many", "tl3d_fuse_frames", "tl3d_grid_reset", "tl3d_grid_device_ptr",
    "tl3d_grid_download", "tl3d_grid_upload", "tl3d_grid_add", "tl3d_grid_touched_bricks", "tl3d_grid_pack_bricks", "tl3d_grid_unpack_bricks", "tl3d_rccl_unique_id", "tl3d_rccl_init", "tl3d_allreduce_grid", "tl3d_extract", "tl3d_statistical_outlier",
    "tl3d_set_profile", "tl3d_set_normal_smoothing", "tl3d_set_tsdf_pairing", "tl3d_get_stats", "tl3d_reset_stats", "tl3d_event_record", "tl3d_event_elapsed_ms",
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
                ("ext_tsdf", C.c_void_p), ("ext_centroid", C.c_void_p), ("stream", C.c_void_p),
                ("pool_bricks_tsdf", C.c_int64), ("pool_bricks_centroid", C.c_int64), ("voxel_offset", C.c_int64 * 3)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("fitness", C.c_double), ("rmse", C.c_double),
                ("n_corr", C.c_int64), ("n_src", C.c_int64), ("iters_run", C.c_int32), ("status", C.c_int32),
                ("scale", C.c_double)]


class IcpParams(C.Structure):
    _fields_ = [("iters", C.c_int32), ("stride", C.c_int32),
                ("max_dist", C.c_double), ("damping", C.c_double), ("eps", C.c_double), ("eig_rel", C.c_double),
                ("estimate_scale", C.c_int32), ("reserved", C.c_int32)]


class IcpPair(C.Structure):
    _fields_ = [("slot_src", C.c_int32), ("slot_tgt", C.c_int32), ("scale_src", C.c_double), ("T_init", C.c_double * 16)]


ICP_MAX_LEVELS = 4


class Stats(C.Structure):
    _fields_ = [("tsdf_launches", C.c_uint64), ("tsdf_records_read", C.c_uint64), ("tsdf_records_written", C.c_uint64),
                ("tsdf_bricks_visited", C.c_uint64), ("tsdf_bricks_free", C.c_uint64), ("tsdf_bricks_free_counted", C.c_uint64), ("centroid_launches", C.c_uint64), ("centroid_points", C.c_uint64),
                ("centroid_dropped", C.c_uint64), ("tsdf_kernel_ms", C.c_double), ("tsdf_kernel_timed", C.c_uint64),
                ("tsdf_batch_bricks", C.c_uint64), ("bp_lookback_retries", C.c_uint64), ("icp_batch_timeouts", C.c_uint64),
                ("icp_batch_fallback_pairs", C.c_uint64), ("merge_bricks_sent", C.c_uint64), ("merge_bricks_total", C.c_uint64),
                ("pool_slots_tsdf", C.c_uint64), ("pool_slots_centroid", C.c_uint64), ("pool_refused", C.c_uint64),
                ("centroid_record_updates", C.c_uint64)]


_lib = None


def ensure_hw_queues(default: str = "16") -> str:
    """ICP lanes (16), the two TSDF prep streams and the main stream want to run side by side; ROCm multiplexes HIP streams
    onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable ONCE, when the HIP runtime initialises (the
    first HIP call of the process, whoever makes it -- `torch.cuda.set_device` counts).  Called at `import tl3d`, so an
    application that imports the package before its first GPU call gets the setting; an explicit value always wins.

    Why 16 and not more (measured, tools/probe_queues.py, profiles/r02_probe_queues.txt): streams multiplexed onto Q <= 22
    queues behave exactly like ceil(n / Q) rounds; from 24 live queues per process on, the GPU's hardware-queue slots are
    over-subscribed and the firmware scheduler time-slices them with a ~10.3 ms quantum -- every cross-stream hand-over can
    then stall for 10 ms (round 1's "intermittent cross-stream hang").  16 keeps the ICP lanes
    concurrent and leaves room for the queues torch and RCCL create in the same process.
    Returns the value in effect for a runtime initialised from now on (tl3d_probe_hw_queues measures the real one)."""
    return os.environ.setdefault("GPU_MAX_HW_QUEUES", default)


def _preload_shared_hip_runtime():
    """Keep ONE HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so
    (same SONAME as /opt/rocm's).  If libtl3d.so pulled in /opt/rocm's copy first, a later `import torch` would load a
    second runtime into the process, and the second one finds no GPU (`RuntimeError: No HIP GPUs are available`: two HSA
    runtimes cannot both open /dev/kfd's queues for one process).  Loading torch's copy first (when torch is installed) makes
    libtl3d.so's DT_NEEDED libamdhip64.so resolve to the copy already in the process, whatever the import order.
    Returns the directory loaded from, or None when torch is not installed (then /opt/rocm's runtime is the only one)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    loaded = None
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError as e:            # falling through would recreate the two-runtimes state: say so instead
                raise ImportError(f"cannot pre-load PyTorch's bundled HIP runtime {path} ({e}); libtl3d.so and torch must share "
                                  "one runtime per process") from e
            loaded = libdir
    return loaded


RUNTIME = {}


def _check_runtime(lib):
    """libtl3d.so is compiled by /opt/rocm's hipcc and runs on whichever libamdhip64.so the process resolved (PyTorch's
    bundled one when torch is installed -- an older minor release in this image).  The host stubs and code-object format
    are stable within a major release; a different MAJOR version is refused, a minor skew is recorded (RUNTIME) and a
    kernel that the runtime could not load would fail loudly at its first launch (every launch is error-checked)."""
    comp, run, drv = C.c_int(0), C.c_int(0), C.c_int(0)
    lib.tl3d_runtime_info(C.byref(comp), C.byref(run), C.byref(drv))
    RUNTIME.update(hip_compiled=comp.value, hip_runtime=run.value, hip_driver=drv.value,
                   GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES"))
    if run.value and comp.value // 10_000_000 != run.value // 10_000_000:
        raise ImportError(f"libtl3d.so was built for HIP {comp.value} but the process runs HIP runtime {run.value}: "
                          "rebuild it with the ROCm release of the runtime in use")


def load():
    """Load libtl3d.so; raises if it has not been built (python __graft_entry__.py / csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `bash {os.path.join(_HERE, 'csrc', 'build.sh')}` "
                          "(there is no CPU fallback for the HIP path)")
    ensure_hw_queues()
    _preload_shared_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    lib.tl3d_last_error.restype = C.c_char_p
    vp, i32, i64, dbl, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_uint32
    sig = {
        "tl3d_version": [],
        "tl3d_device_count": [C.POINTER(C.c_int)],
        "tl3d_runtime_info": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "tl3d_probe_hw_queues": [i32, i32, dbl, C.POINTER(dbl)],
        "tl3d_grid_max_weight": [vp, C.POINTER(i64)],
        "tl3d_create": [C.POINTER(Config), i32, C.POINTER(vp)],
        "tl3d_destroy": [vp],
        "tl3d_sync": [vp],
        "tl3d_get_stream": [vp, C.POINTER(vp)],
        "tl3d_release_cached_memory": [],
        "tl3d_upload_frame": [vp, i32, vp, i32, vp],
        "tl3d_download_depth": [vp, i32, vp],
        "tl3d_pinned_alloc": [C.c_size_t, C.POINTER(vp)],
        "tl3d_pinned_free": [vp],
        "tl3d_upload_frame_async": [vp, i32, vp, i32, vp],
        "tl3d_slot_wait": [vp, i32],
        "tl3d_attach_grid": [vp, C.POINTER(Config)],
        "tl3d_backproject": [vp, i32, vp, vp, dbl, u32, i32, dbl, dbl, vp, vp, i64, C.POINTER(i64)],
        "tl3d_backproject_device": [vp, i32, vp, vp, dbl, u32, i32, dbl, dbl, vp, vp, i64, vp],
        "tl3d_frame_bounds": [vp, i32, vp, vp, dbl, u32, i32, dbl, dbl, vp, vp, vp],
        "tl3d_frames_bounds": [vp, i32, vp, vp, vp, vp, u32, i32, dbl, dbl, vp, vp],
        "tl3d_count_bricks": [vp, C.POINTER(Config), i32, vp, vp, vp, vp, i32, dbl, dbl, C.POINTER(i64), C.POINTER(i64)],
        "tl3d_accumulate_centroid": [vp, i32, vp, vp, dbl, u32, i32, dbl, dbl],
        "tl3d_accumulate_points": [vp, vp, vp, i64],
        "tl3d_points_bounds": [vp, vp, i64, vp, vp],
        "tl3d_integrate": [vp, i32, vp, vp, dbl],
        "tl3d_build_normals": [vp, i32, dbl, dbl],
        "tl3d_download_normals": [vp, i32, vp],
        "tl3d_icp_p2plane": [vp, i32, dbl, i32, vp, C.POINTER(IcpParams), C.POINTER(IcpResult)],
        "tl3d_icp_enqueue": [vp, i32, i32, dbl, i32, vp, C.POINTER(IcpParams)],
        "tl3d_icp_collect": [vp, i32, C.POINTER(IcpResult)],
        "tl3d_icp_batch_enqueue": [vp, C.POINTER(IcpPair), i32, C.POINTER(IcpParams), i32],
        "tl3d_icp_batch_collect": [vp, C.POINTER(IcpResult), i32],
        "tl3d_build_normals_many": [vp, i32, vp, vp, dbl],
        "tl3d_fuse_frames": [vp, i32, vp, vp, vp, vp, u32, i32, dbl, dbl],
        "tl3d_host_pack_bgr_rows": [vp, vp, i32, i32],
        "tl3d_host_copy_rows": [vp, vp, i32, C.c_size_t],
        "tl3d_grid_reset": [vp],
        "tl3d_grid_device_ptr": [vp, u32, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "tl3d_grid_download": [vp, u32, vp, C.c_size_t],
        "tl3d_grid_upload": [vp, u32, vp, C.c_size_t],
        "tl3d_grid_add": [vp, u32, vp, C.c_size_t],
        "tl3d_rccl_unique_id": [vp],
        "tl3d_rccl_init": [vp, i32, i32, vp],
        "tl3d_allreduce_grid": [vp, u32],
        "tl3d_extract": [vp, i32, i32, i32, dbl, vp, vp, i64, C.POINTER(i64)],
        "tl3d_statistical_outlier": [vp, vp, i64, i32, dbl, dbl, vp, C.POINTER(i64)],
        "tl3d_set_profile": [vp, i32, i32],
        "tl3d_set_tsdf_pairing": [vp, i32],
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
    _check_runtime(lib)
    _lib = lib
    return lib


def probe_hw_queues(device: int = 0, n_streams: int = 48, spin_ms: float = 1.0) -> dict:
    """How many hardware queues HIP streams really get in this process (needs a GPU): n_streams kernels of spin_ms on
    n_streams fresh streams take ceil(n_streams / Q) * spin_ms."""
    ms = C.c_double(0.0)
    check(load().tl3d_probe_hw_queues(int(device), int(n_streams), float(spin_ms), C.byref(ms)))
    rounds = max(1, int(round(ms.value / spin_ms - 0.2)))          # ~0.2 ms of launch overhead ride on the last round
    return dict(streams=int(n_streams), spin_ms=float(spin_ms), elapsed_ms=round(ms.value, 3),
                effective_queues=-(-int(n_streams) // rounds), GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES"))


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
