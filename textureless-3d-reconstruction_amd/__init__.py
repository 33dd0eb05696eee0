"""MI355X-native depth-fusion back end for kamalnath26/textureless-3d-reconstruction.

Hot path only: depth map -> back-projection -> point-to-plane ICP -> TSDF /
voxel-centroid fusion -> .ply, as hand-written HIP kernels for gfx950 behind the C-ABI in
include/tl3d.h.  Import as `tl3d` (tl3d.py at the repo root maps the hyphenated directory name onto
an importable package).
"""
from . import _cabi  # noqa: F401

_cabi.ensure_hw_queues()        # before the process's first HIP call (the runtime reads it once): see _cabi.ensure_hw_queues
from ._cabi import CH_CENTROID, CH_TSDF, EXTRACT_CENTROID, EXTRACT_TSDF, ICP_LANES, Tl3dError  # noqa: F401
from .fusion import FusionContext, GridSpec, release_cached_memory  # noqa: F401

__all__ = ["FusionContext", "GridSpec", "release_cached_memory", "Tl3dError", "CH_TSDF", "CH_CENTROID", "EXTRACT_CENTROID", "EXTRACT_TSDF"]
