"""Configuration objects with the reference's names and defaults.

ReconstructionConfig mirrors depth_to_reconstruction.py:45-73; CameraIntrinsics mirrors
depth_enhanced_reconstruction.py:57-80.  Fields the SfM front end used (match_ratio,
ransac_threshold) are kept so existing call sites keep constructing the object; the
additive fields at the end configure the device grid and the ICP pose source.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class ReconstructionConfig:
    # camera intrinsics (D2R:49-52)
    fx: float = 1719.0
    fy: float = 1719.0
    cx: float = 540.0
    cy: float = 960.0
    # depth processing (D2R:55-57)
    depth_scale: float = 1.0
    min_depth: float = 0.1
    max_depth: float = 50.0
    # feature matching (D2R:60-61) -- unused by the ICP pose source, kept for signature compatibility
    match_ratio: float = 0.75
    ransac_threshold: float = 3.0
    # point cloud (D2R:64-65)
    voxel_size: float = 0.005
    subsample_factor: int = 2

    # ---- additive: device fusion ------------------------------------------------------------------
    grid_dim: int = 1024                # fusion volume budget: at most grid_dim^3 voxels IN TOTAL, spread over the axes as
                                        # the scene needs (a corridor gets 400 x 480 x 2500, not a clipped cube)
    sdf_trunc_voxels: float = 4.0       # TSDF truncation in voxels
    icp_coarse: tuple = ((10, 4, 0.20),)   # levels before the final one: (iterations, pixel stride, gate in metres)
    icp_iters: int = 15
    icp_stride: int = 2
    icp_max_dist: float = 0.05
    icp_damping: float = 1e-6
    icp_eps: float = 1e-7               # a level stops once the largest component of its pose update is below this (rad / m)
    icp_eig_rel: float = 1e-4           # relative eigenvalue cutoff: unobservable DOFs keep the motion prior
    icp_smooth_radius: int = 1          # normals (and the registration's source depth) from the depth averaged over a (2 r + 1)^2 window; 0 = raw
    scale_update_weight: float = 0.3    # estimate_scale runs: avg = (1 - w) avg + w scale_i (D2R:650 uses 0.3); 1 = each view's own estimate
    tsdf_min_weight: int = 0            # > 0: gate the emitted centroids by the TSDF (outlier suppression)
    tsdf_max_abs: float = 1.0
    # statistical outlier removal after the voxel merge: D2R's merge_pointclouds runs it (20 neighbours, 2 sigma,
    # D2R:412-415); DER's merge_pointclouds has none (DER:615-645) -- the DER command line switches it off
    outlier_filter: bool = True
    outlier_nb_neighbors: int = 20
    outlier_std_ratio: float = 2.0
    device: int = 0

    @property
    def K(self) -> np.ndarray:
        return np.array([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]], dtype=np.float64)


@dataclass
class CameraIntrinsics:
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int

    def to_matrix(self) -> np.ndarray:
        return np.array([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]], dtype=np.float64)

    @classmethod
    def from_matrix(cls, K: np.ndarray, width: int, height: int) -> "CameraIntrinsics":
        return cls(fx=float(K[0, 0]), fy=float(K[1, 1]), cx=float(K[0, 2]), cy=float(K[1, 2]),
                   width=int(width), height=int(height))
