"""CPU: the C-ABI library loads and exports every symbol include/tl3d.h declares; constants and struct layouts in the
binding agree with the header; the product path fails loudly without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from tl3d import _cabi as abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tl3d.h")


def _header():
    with open(HEADER) as f:
        return f.read()


def test_library_exports_every_declared_symbol():
    declared = sorted(set(re.findall(r"\b(tl3d_[a-z0-9_]+)\s*\(", _header())))
    assert len(declared) >= 28
    lib = abi.load()
    for name in declared:
        assert hasattr(lib, name), f"libtl3d.so does not export {name}"
    assert sorted(abi.SYMBOLS) == declared, "binding's symbol list and the header disagree"
    assert lib.tl3d_version() == abi.ABI_VERSION


def test_constants_match_header():
    h = _header()

    def define(name):
        m = re.search(rf"#define\s+{name}\s+\(?(-?\d+)u?\)?", h)
        assert m, name
        return int(m.group(1))
    for name, val in (("TL3D_ABI_VERSION", abi.ABI_VERSION), ("TL3D_OK", abi.OK), ("TL3D_E_INVALID", abi.E_INVALID),
                      ("TL3D_E_HIP", abi.E_HIP), ("TL3D_E_NOMEM", abi.E_NOMEM), ("TL3D_E_CAPACITY", abi.E_CAPACITY),
                      ("TL3D_E_STATE", abi.E_STATE), ("TL3D_E_NODEVICE", abi.E_NODEVICE), ("TL3D_CH_TSDF", abi.CH_TSDF),
                      ("TL3D_CH_CENTROID", abi.CH_CENTROID), ("TL3D_DEPTH_F32_M", abi.DEPTH_F32_M),
                      ("TL3D_DEPTH_U16_MM", abi.DEPTH_U16_MM), ("TL3D_F_SCALE_F64", abi.F_SCALE_F64),
                      ("TL3D_F_NO_POSE", abi.F_NO_POSE), ("TL3D_EXTRACT_CENTROID", abi.EXTRACT_CENTROID),
                      ("TL3D_EXTRACT_TSDF", abi.EXTRACT_TSDF)):
        assert define(name) == val, name


def _struct_fields(name):
    m = re.search(rf"typedef struct {name} \{{(.*?)\}} {name};", _header(), re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("void") else decl[len("void"):]
        for n in names.split(","):
            out.append(re.sub(r"[\*\[\]0-9 ]", "", n))
    return out


@pytest.mark.parametrize("cname,cls", [("tl3d_config", abi.Config), ("tl3d_icp_result", abi.IcpResult),
                                       ("tl3d_icp_params", abi.IcpParams), ("tl3d_stats", abi.Stats)])
def test_struct_field_order_matches_header(cname, cls):
    assert _struct_fields(cname) == [f for f, _ in cls._fields_]


def test_no_cpu_fallback_without_gpu():
    if abi.device_count() > 0:
        pytest.skip("a GPU is visible")
    import tl3d
    with pytest.raises(RuntimeError, match="no HIP device"):
        tl3d.FusionContext(64, 48, 50.0, 50.0, 32.0, 24.0)
    # and the raw ABI reports the same through its error channel instead of computing anything
    cfg = abi.Config()
    cfg.abi_version, cfg.width, cfg.height, cfg.fx, cfg.fy, cfg.n_slots = abi.ABI_VERSION, 8, 8, 1.0, 1.0, 1
    h = C.c_void_p()
    rc = abi.load().tl3d_create(C.byref(cfg), 0, C.byref(h))
    assert rc == abi.E_NODEVICE and b"no HIP device" in abi.load().tl3d_last_error()


def test_argument_validation_needs_no_gpu():
    lib = abi.load()
    h = C.c_void_p()
    cfg = abi.Config()
    cfg.abi_version = 99
    assert lib.tl3d_create(C.byref(cfg), 0, C.byref(h)) == abi.E_INVALID
    cfg.abi_version, cfg.width, cfg.height, cfg.fx, cfg.fy, cfg.n_slots = abi.ABI_VERSION, 8, 8, 1.0, 1.0, 1
    cfg.channels, cfg.nx, cfg.ny, cfg.nz, cfg.voxel_size, cfg.sdf_trunc = abi.CH_TSDF, 12, 8, 8, 0.01, 0.04
    assert lib.tl3d_create(C.byref(cfg), 0, C.byref(h)) == abi.E_INVALID          # 12 is not a multiple of the brick
    assert b"multiples of 8" in lib.tl3d_last_error()
    assert lib.tl3d_sync(None) == abi.E_INVALID and lib.tl3d_destroy(None) == abi.OK
