"""GPU: the BASELINE.json configurations as parity cases.  Config 1 lives in
test_gpu_pipeline.py (CLI plumbing); here: config 2 (1080x1920 turntable into 256^3), config 4 (textureless cylinder
orbit, ICP every frame, reduced length) and config 5 at FULL size (3840x2160 into 1024^3 @ 2 mm) bit for bit."""
import numpy as np
import pytest

import tl3d
from oracle import c_oracle
from oracle import ref_numpy as rn
from tl3d import synth
from tl3d.config import ReconstructionConfig
from tl3d.pipeline import DepthToReconstructionPipeline, align_grid_to_open3d

pytestmark = pytest.mark.gpu


def test_config5_full_size_grid_bit_exact():
    W, H = 3840, 2160
    cam = dict(width=W, height=H, fx=3000.0, fy=3000.0, cx=1920.0, cy=1080.0)
    dims, voxel = (1024, 1024, 1024), 0.002
    origin = (-1.024, -1.124, -1.024)
    scene = synth.object_scene(with_room=True)
    pose = synth.orbit_poses(4, 1.0, 90.0)[1]
    depth, _ = synth.render(scene, pose, want_color=False, **cam)
    spec = tl3d.GridSpec(dims, origin, voxel, 4 * voxel, tl3d.CH_TSDF)
    orc = c_oracle.Oracle(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], dims=dims, origin=origin, voxel_size=voxel,
                          sdf_trunc=4 * voxel)
    orc.centroid = None
    with tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=1, grid=spec) as ctx:
        ctx.upload(0, depth, None)
        ctx.set_profile(True, False)
        ctx.integrate(0, pose)
        st = ctx.stats()
        g = ctx.download_grid(tl3d.CH_TSDF)
    orc.tsdf_integrate(depth, pose[0], pose[1])
    assert g.shape == (1 << 30, 2)
    updated = int(orc.tsdf[:, 1].sum())
    assert updated > 50_000_000
    assert np.array_equal(g, orc.tsdf)                                       # 2^30 records, bit for bit
    assert st["tsdf_bricks_free"] > 10000 and st["tsdf_bricks_visited"] < 2_097_152
    # the highest brick of the grid is addressable: last record index = 2^30 - 1
    assert orc.vox_index(1023, 1023, 1023) == (1 << 30) - 1


def test_config2_turntable_into_256_grid():
    W, H = 1080, 1920
    cfg = ReconstructionConfig(voxel_size=0.005, subsample_factor=2, grid_dim=256)          # reference defaults 1719/540/960
    scene = synth.object_scene(with_room=False)                                            # object only: background invalid
    poses = synth.orbit_poses(10, 1.0, 7.2)
    frames = [synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy) for p in poses]
    assert 0.5 < (frames[0][0] == 0).mean() < 0.95                                          # most rays miss: depth 0, dropped
    clouds = [rn.backproject(dd, cc, cfg.fx, cfg.fy, cfg.cx, cfg.cy, pose=p, subsample=2) for (dd, cc), p in zip(frames, poses)]
    min_bound = np.min([c[0].min(0) for c in clouds], axis=0)
    # 1.28 m cube on the object, snapped (< 1 voxel) onto Open3D's voxel lattice so centroids are comparable (H1)
    grid = align_grid_to_open3d(tl3d.GridSpec.cube(256, 0.005, centre=(0.0, -0.05, 0.0)), min_bound)
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames([c for d, c in frames], [d for d, c in frames])
    pts, col, _ = pipe.reconstruct(grid=grid, poses=poses)
    assert len(pts) > 15000 and pipe.stats["points_dropped"] == 0
    # every fused point lies on the analytic sphere union (nearest sphere surface) within the voxel scale
    d = np.min([np.abs(np.linalg.norm(pts - np.asarray(c), axis=1) - r) for c, r in scene.spheres], axis=0)
    assert d.mean() < 1e-3 and np.percentile(d, 99) < 4e-3
    # and equals the restated reference CPU path (vstack -> voxel centroid -> SOR) up to the offset quantum
    ref_p, _ = rn.merge_open3d(clouds, 0.005, sor=True)
    ch = rn.chamfer_mean(pts, ref_p)
    assert ch < 1e-4, ch                                # north-star bar: 1 mm mean Chamfer


def test_config4_cylinder_orbit_icp_chain_reduced():
    W, H = 1280, 720
    cfg = ReconstructionConfig(fx=1000.0, fy=1000.0, cx=640.0, cy=360.0, voxel_size=0.01, subsample_factor=4, grid_dim=512,
                               icp_iters=12, icp_stride=2, icp_max_dist=0.05, max_depth=4.0)   # the ground plane runs to the horizon
    scene = synth.cylinder_scene(ground=True)
    n = 24
    poses = synth.orbit_poses(n, 1.5, 0.36, height=-0.2)
    frames = [synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy) for p in poses]
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames([c for d, c in frames], [d for d, c in frames])
    pts, col, est = pipe.reconstruct()
    assert len(est) == n and all(r["status"] != 2 for r in pipe.icp_log)
    # chain drift against the analytic orbit, expressed relative to camera 0
    r0, t0 = poses[0]
    rg = poses[-1][0] @ r0.T
    tg = poses[-1][1].reshape(3) - rg @ t0.reshape(3)
    err = np.linalg.norm(est[-1][0] - rg) + np.linalg.norm(est[-1][1].reshape(3) - tg)
    total_motion = np.linalg.norm(np.eye(3) - rg) + np.linalg.norm(tg)
    assert err < 0.1 * total_motion + 2e-3, (err, total_motion)
    assert len(pts) > 5000
