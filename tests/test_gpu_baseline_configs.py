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


def test_config3_corridor_dolly_icp_chain_in_a_budgeted_grid():
    """BASELINE config 3 (stand-in for exp_tunnel_set1_images_1_fps): textureless rectangular
    corridor 2 m x 2.4 m, 640x480, fx = fy = 512, camera advancing 10 cm per frame, 5 mm voxels.  Sliding along the
    axis is observable only through the far wall and two small spheres, so the chain leans on the eigenvalue cutoff and
    the constant-velocity prior.  The fusion grid is planned from the data under a voxel BUDGET: ~400 x 480 x 2000+
    voxels, nothing clipped.  Fused cloud vs the restated reference CPU path (back-project with the analytic poses ->
    vstack -> Open3D voxel centroid; DER-style: no outlier filter) and vs the analytic walls."""
    W, H = 640, 480
    cfg = ReconstructionConfig(fx=512.0, fy=512.0, cx=320.0, cy=240.0, voxel_size=0.005, subsample_factor=2, grid_dim=1024,
                               outlier_filter=False)
    scene = synth.corridor_scene()
    n = 40
    poses = synth.dolly_poses(n, (0.0, 0.0, 0.0), (0.0, 0.0, 0.1))
    assert np.allclose(poses[0][0], np.eye(3)) and np.allclose(poses[0][1], 0)            # camera 0 is the world frame
    frames = [synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy) for p in poses]
    assert all((d > 0).all() for d, _ in frames)                                           # closed corridor: every ray hits
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames([c for d, c in frames], [d for d, c in frames])
    pts, col, est = pipe.reconstruct()
    assert len(est) == n and all(r["status"] != 2 for r in pipe.icp_log)
    # the grid follows the corridor: longer than the old 512-per-axis cap, nothing dropped
    assert pipe.grid.dims[2] > 1536 and pipe.grid.dims[0] <= 416 and pipe.grid.dims[1] <= 496
    assert pipe.stats["points_dropped"] == 0 and pipe.stats["voxels"] == len(pts)          # and no outlier filter ran
    # chain drift over 3.9 m of travel
    err = max(np.linalg.norm(r - rg) + np.linalg.norm(t.reshape(3) - tg.reshape(3)) for (r, t), (rg, tg) in zip(est, poses))
    assert err < 5e-3, err
    # fused points lie on the analytic corridor (side walls, floor/ceiling, far wall, spheres)
    (lo, hi) = scene.room
    d_wall = np.min(np.stack([np.abs(pts[:, 0] - lo[0]), np.abs(pts[:, 0] - hi[0]), np.abs(pts[:, 1] - lo[1]),
                              np.abs(pts[:, 1] - hi[1]), np.abs(pts[:, 2] - hi[2])]), axis=0)
    d_sph = np.min([np.abs(np.linalg.norm(pts - np.asarray(c), axis=1) - r) for c, r in scene.spheres], axis=0)
    d = np.minimum(d_wall, d_sph)
    assert d.mean() < 1e-3 and np.percentile(d, 99) < 5e-3, (d.mean(), np.percentile(d, 99))
    # the reference CPU path on the same frames with the analytic poses
    clouds = [rn.backproject(dd, cc, cfg.fx, cfg.fy, cfg.cx, cfg.cy, pose=p, subsample=2) for (dd, cc), p in zip(frames, poses)]
    ref_p, _ = rn.merge_open3d(clouds, cfg.voxel_size, sor=False)
    assert abs(len(pts) - len(ref_p)) < 0.05 * len(ref_p)
    ch = rn.chamfer_mean(pts, ref_p)
    assert ch < 1e-3, ch                                 # north-star bar: 1 mm mean Chamfer


def test_icp_at_the_headline_resolution_matches_oracle():
    """1080x1920 (fx = fy = 1719, the reference defaults): normal map bit for bit, ICP at pixel strides 2 and 4 vs the C
    oracle -- same source count, same iteration count, pose within the north-star 1e-4 Frobenius (and in fact ~1e-9)."""
    W, H = 1080, 1920
    cam = dict(width=W, height=H, fx=1719.0, fy=1719.0, cx=540.0, cy=960.0)
    scene = synth.object_scene(with_room=True)
    poses = synth.orbit_poses(2, 1.0, 1.5)
    frames = [synth.render(scene, p, want_color=False, **cam) for p in poses]
    orc = c_oracle.Oracle(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    with tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=2, grid=None) as ctx:
        ctx.upload(0, frames[0][0], None)
        ctx.upload(1, frames[1][0], None)
        ctx.build_normals(1)
        onm = orc.normals(frames[1][0])
        assert np.array_equal(ctx.download_normals(1), onm)
        r_rel, t_rel = synth.relative_pose(poses[0], poses[1])
        T_true = np.eye(4); T_true[:3, :3] = r_rel; T_true[:3, 3] = t_rel.ravel()
        for stride, iters in ((4, 10), (2, 10), (1, 4)):
            res = ctx.icp(0, 1, iters=iters, stride=stride, max_dist=0.1)
            ores = orc.icp(frames[0][0], onm, iters=iters, stride=stride, max_dist=0.1)
            assert res["n_src"] == ores["n_src"] == -(-H // stride) * -(-W // stride)
            assert abs(res["n_corr"] - ores["n_corr"]) <= 2 and res["iters_run"] == ores["iters_run"]
            dT = np.linalg.norm(res["T"] - ores["T"])
            assert dT <= 1e-4 and dT <= 1e-8, (stride, dT)
            assert abs(res["rmse"] - ores["rmse"]) < 1e-9
            assert np.linalg.norm(res["T"] - T_true) < 2e-3, stride
            # the batched kernel (64 workgroups share this pair at strides 1 and 2, 32 at stride 4)
            bres = ctx.icp_batch([(0, 1)], [dict(iters=iters, stride=stride, max_dist=0.1)])[0]
            assert bres["n_src"] == ores["n_src"] and abs(bres["n_corr"] - ores["n_corr"]) <= 2 and bres["iters_run"] == ores["iters_run"]
            assert np.linalg.norm(bres["T"] - ores["T"]) <= 1e-8, (stride, "batched")


def test_config2_with_one_millimetre_depth_noise():
    """Config 2's second variant (depth with N(0, 1 mm) noise).  Parity is what is asserted: on noisy frames the
    device ICP equals the C oracle's (pose within the north-star 1e-4 Frobenius), and the fused cloud of the noisy frames at the
    true poses is within 1 mm mean Chamfer of the restated reference CPU path on the same frames.  The pose chain over the noisy
    frames uses normals (and source depth) averaged over 3 x 3 windows: it stays within 2 mm / 0.1 degrees over the 12 frames, and
    the cloud fused with THOSE poses is within the 1 mm bar too."""
    W, H = 540, 960                                                         # half resolution: the oracle-side merge stays quick
    cfg = ReconstructionConfig(fx=859.5, fy=859.5, cx=270.0, cy=480.0, voxel_size=0.005, subsample_factor=2, grid_dim=256)
    scene = synth.object_scene(with_room=False)
    poses = synth.orbit_poses(12, 1.0, 3.6)
    frames = [synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, noise_sigma=0.001, seed=100 + i) for i, p in enumerate(poses)]
    clean = synth.render(scene, poses[0], W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy)[0]
    valid = clean > 0
    assert 2e-4 < np.abs(frames[0][0][valid] - clean[valid]).mean() < 2e-3        # the noise is really there
    # (a) ICP parity on a noisy pair, per-iteration kernel and batched kernel
    orc = c_oracle.Oracle(W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy)
    with tl3d.FusionContext(W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, n_slots=2, grid=None) as ctx:
        ctx.upload(0, frames[0][0], None)
        ctx.upload(1, frames[1][0], None)
        ctx.build_normals(1)
        onm = orc.normals(frames[1][0])
        assert np.array_equal(ctx.download_normals(1), onm)
        res = ctx.icp(0, 1, iters=12, stride=2, max_dist=0.05)
        bres = ctx.icp_batch([(0, 1)], [dict(iters=12, stride=2, max_dist=0.05)])[0]
        ores = orc.icp(frames[0][0], onm, iters=12, stride=2, max_dist=0.05)
    assert np.linalg.norm(res["T"] - ores["T"]) <= 1e-4 and np.linalg.norm(bres["T"] - ores["T"]) <= 1e-4
    assert res["iters_run"] == ores["iters_run"] and abs(res["n_corr"] - ores["n_corr"]) <= 2
    # (b) fusion parity at the true poses (frame of camera 0)
    r0, t0 = poses[0]
    rel = [(r @ r0.T, t.reshape(3, 1) - (r @ r0.T) @ t0.reshape(3, 1)) for r, t in poses]
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames([c for d, c in frames], [d for d, c in frames])
    pts, col, _ = pipe.reconstruct(poses=rel)
    clouds = [rn.backproject(dd, cc, cfg.fx, cfg.fy, cfg.cx, cfg.cy, pose=p, subsample=2) for (dd, cc), p in zip(frames, rel)]
    ref_p, _ = rn.merge_open3d(clouds, 0.005, sor=True)
    ch = rn.chamfer_mean(pts, ref_p)
    assert len(pts) > 5000 and ch < 1e-3, ch                                     # north-star bar: 1 mm mean Chamfer
    # (c) the chain over the noisy frames: all frames kept, drift bounded
    pipe2 = DepthToReconstructionPipeline(cfg)
    pipe2.set_frames([c for d, c in frames], [d for d, c in frames])
    pts2, _, est = pipe2.reconstruct()
    assert len(est) == 12
    worst_t = max(float(np.linalg.norm(np.asarray(t).reshape(3) - np.asarray(te).reshape(3))) for (_, t), (_, te) in zip(rel, est))
    worst_r = max(float(np.degrees(np.arccos(np.clip((np.trace(r @ np.asarray(re).T) - 1) / 2, -1, 1)))) for (r, _), (re, _) in zip(rel, est))
    print(f"noisy chain over 40 degrees: worst translation error {worst_t * 1e3:.1f} mm, rotation {worst_r:.2f} deg")
    assert worst_t < 0.002 and worst_r < 0.1, (worst_t, worst_r)                 # window-averaged normals (config.icp_smooth_radius = 1)
    ch2 = rn.chamfer_mean(pts2, ref_p)
    assert ch2 < 1e-3, ch2                                                       # the cloud fused WITH the registration's poses: 1 mm bar
    # (d) the averaged depth / normal map of the device equal the oracle's bit for bit, and so does a registration on them
    with tl3d.FusionContext(W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, n_slots=2, grid=None) as ctx:
        ctx.set_normal_smoothing(1)
        for k in (0, 1):
            ctx.upload(k, frames[k][0], None)
            ctx.build_normals(k)
        sd0, _ = orc.normals_smooth(frames[0][0], radius=1)
        sd1, onm1 = orc.normals_smooth(frames[1][0], radius=1)
        assert np.array_equal(ctx.download_normals(1), onm1)
        res = ctx.icp(0, 1, iters=12, stride=2, max_dist=0.05)
        bres = ctx.icp_batch([(0, 1)], [dict(iters=12, stride=2, max_dist=0.05)])[0]
        ores = orc.icp(sd0, onm1, iters=12, stride=2, max_dist=0.05)
    assert np.linalg.norm(res["T"] - ores["T"]) <= 1e-4 and np.linalg.norm(bres["T"] - ores["T"]) <= 1e-4


def test_a_30_m_corridor_at_5_mm_fuses_in_a_sparse_volume():
    """What the reference's hash-map merge gives for free (any extent at any voxel size, D2R:404-410): a 2 m x 2.4 m x 30 m
    corridor at 5 mm is 400 x 488 x 6008 voxels -- 9.4 GB of TSDF and 37.5 GB of centroid records as a dense grid.  The sparse
    volume holds records only for the bricks near the walls: every point is kept (points_dropped == 0, no brick refused), in
    well under 8 GB, and the cloud lies on the walls."""
    from tl3d import synth
    from tl3d.config import ReconstructionConfig
    from tl3d.pipeline import DepthToReconstructionPipeline
    W, H = 640, 480
    cam = dict(fx=512.0, fy=512.0, cx=320.0, cy=240.0)
    scene = synth.Scene(room=((-1.0, -1.2, -0.5), (1.0, 1.2, 30.0)))
    n = 58
    poses = synth.dolly_poses(n, (0.0, 0.0, 0.0), (0.0, 0.0, 0.5))
    r0, t0 = poses[0]
    rel = [(r @ r0.T, t.reshape(3, 1) - (r @ r0.T) @ t0.reshape(3, 1)) for r, t in poses]
    frames = [synth.render(scene, p, W, H, **cam) for p in poses]
    cfg = ReconstructionConfig(**cam, voxel_size=0.005, subsample_factor=2, grid_dim=512, max_depth=4.0, outlier_filter=False)
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames([c for d, c in frames], [d for d, c in frames])
    pts, col, _ = pipe.reconstruct(poses=rel)
    g = pipe.grid
    assert g.sparse and g.nvox > 1.0e9 and g.dims[2] >= 5600, g          # (the floor comes into view 1.6 m ahead of the first camera)
    assert g.pool_centroid > 0 and g.device_bytes() < 14 * 2 ** 30, g.device_bytes() / 2 ** 30       # (centroid records: 37.5 GB dense, < 3 GB here; the TSDF channel keeps every free-space observation: most of its bricks hold records)
    st = pipe.stats
    assert st["points_dropped"] == 0 and st["pool_refused"] == 0
    assert st["bricks_centroid"] > 50000 and st["bricks_tsdf"] > st["bricks_centroid"]
    assert len(pts) > 2_000_000
    # on the walls / floor / ceiling of the corridor (camera-0 frame == world frame here up to cam0's pose)
    pw = (pts - t0.reshape(1, 3)) @ r0
    d = np.minimum(np.minimum(np.abs(np.abs(pw[:, 0]) - 1.0), np.abs(np.abs(pw[:, 1]) - 1.2)), np.abs(pw[:, 2] - 30.0))
    assert d.mean() < 1e-3 and np.percentile(d, 99) < 4e-3


def test_merge_pointclouds_of_a_large_extent_uses_a_sparse_grid():
    """merge_pointclouds (D2R:386-420 / DER:615-645) on clouds 61 m x 26 m x 3 m apart at 5 mm -- the reference's defaults reach 50 m
    (D2R:57, :64) and Open3D's hash map holds the occupied voxels whatever the extent.  The lattice has 3.9e10 voxels: more than one
    grid's brick table indexes (2^32), so it is fused block by block (tl3d_config.voxel_offset: blocks of ONE lattice, identical
    voxels), each block a sparse grid; the result equals the restated Open3D merge."""
    from tl3d.config import ReconstructionConfig
    from tl3d.dense import DensePointCloudGenerator
    from tl3d.config import CameraIntrinsics
    from oracle import ref_numpy as rn
    rng = np.random.default_rng(7)
    a = (rng.random((40000, 3)) * np.array([1.0, 1.0, 0.02])).astype(np.float32)
    b = a + np.array([61.0, 26.0, 3.0], np.float32)
    c = (rng.random((20000, 3)) * np.array([61.0, 0.01, 0.01]) + np.array([0.0, 13.0, 1.5])).astype(np.float32)     # a line across every block boundary in x
    ca = rng.integers(0, 255, (40000, 3)).astype(np.uint8)
    cc = rng.integers(0, 255, (20000, 3)).astype(np.uint8)
    gen = DensePointCloudGenerator(CameraIntrinsics(fx=500.0, fy=500.0, cx=320.0, cy=240.0, width=640, height=480))
    try:
        pts, colr = gen.merge_pointclouds([(a, ca), (b, ca), (c, cc)], voxel_size=0.005)
    finally:
        gen.close()
    ref_p, ref_c = rn.merge_open3d([(a, ca), (b, ca), (c, cc)], 0.005, sor=False)
    assert len(pts) == len(ref_p) and rn.chamfer_mean(pts, ref_p) < 2e-6
    # (the same number of voxels and every centroid within the accumulator quantum of one of the reference's: the same voxel set)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(ref_p).query(pts)
    assert d.max() < 0.005 / 1024 + 61.0 * 2 ** -22
