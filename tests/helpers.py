"""Shared builders for the parity tests: small scenes, grids, oracle/GPU context pairs."""
import numpy as np

import tl3d
from tl3d import synth
from oracle import c_oracle

SMALL = dict(width=160, height=120, fx=140.0, fy=140.0, cx=79.5, cy=59.5)


def small_scene_frames(n=3, deg=2.0, radius=1.0, noise=0.0, cam=SMALL, scene=None):
    scene = scene or synth.object_scene()
    poses = synth.orbit_poses(n, radius, deg)
    frames = [synth.render(scene, p, cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                           noise_sigma=noise, seed=i) for i, p in enumerate(poses)]
    return poses, frames


def make_pair(cam=SMALL, dims=(64, 64, 64), voxel=0.02, centre=(0.0, 0.0, 0.0), trunc=None, n_slots=4,
              min_depth=0.1, max_depth=50.0, channels=tl3d.CH_TSDF | tl3d.CH_CENTROID):
    """(GPU FusionContext, C oracle) over the same geometry."""
    origin = tuple(centre[i] - 0.5 * dims[i] * voxel for i in range(3))
    trunc = 4 * voxel if trunc is None else trunc
    spec = tl3d.GridSpec(dims, origin, voxel, trunc, channels)
    ctx = tl3d.FusionContext(cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                             min_depth=min_depth, max_depth=max_depth, n_slots=n_slots, grid=spec)
    orc = c_oracle.Oracle(cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                          min_depth=min_depth, max_depth=max_depth, dims=dims, origin=origin, voxel_size=voxel,
                          sdf_trunc=trunc)
    return ctx, orc


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)
