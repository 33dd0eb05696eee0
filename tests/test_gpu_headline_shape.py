"""GPU: the shape bench.py times, compared with the every-voxel oracle -- 1080x1920 frames (fx = fy = 1719), the closed-room orbit
0.70 degrees per frame, 40 frames = one full 32-frame update launch + one of 8, into 512^3 @ 5 mm, float32 and 16-bit frames;
and BASELINE configs 3 and 4 at (config 3: full; config 4: 256 frames) length with the assertions tools/run_config.py prints
(SURVEY.md section 8d Headline / Config 3 / Config 4; the reference steps are D2R:328-420 through the restated path)."""
import numpy as np
import pytest

import tl3d
from oracle import c_oracle
from oracle import ref_numpy as rn
from tl3d import synth
from tl3d.config import ReconstructionConfig
from tl3d.pipeline import DepthToReconstructionPipeline

pytestmark = pytest.mark.gpu


def _render_on_device(scene, poses, cam, want_color=False):
    """frames ray-cast with torch on the GPU (plumbing: 1080p frames take seconds each in numpy), returned as host arrays"""
    import torch
    dev = torch.device("cuda", 0)
    out = []
    for p in poses:
        d, c = synth.render(scene, p, cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], xp=torch, device=dev,
                            want_color=want_color)
        out.append((d.cpu().numpy(), c.cpu().numpy() if c is not None else None))
    return out


@pytest.mark.parametrize("as_u16", [False, True])
def test_headline_shape_32_plus_8_frames_into_512_cube_is_the_oracle_grid(as_u16):
    H = synth.HEADLINE
    cam = dict(width=H["width"], height=H["height"], fx=H["fx"], fy=H["fy"], cx=H["cx"], cy=H["cy"])
    n = 40
    scene = synth.object_scene(with_room=True)
    poses = synth.orbit_poses(n, H["radius"], H["deg_per_frame"], start_deg=37 * H["deg_per_frame"])
    frames = _render_on_device(scene, poses, cam)
    assert all((d > 0.1).all() for d, _ in frames)                             # closed room: every pixel valid, as in bench.py
    spec = tl3d.GridSpec.cube(H["grid"], H["voxel"], centre=(0.0, -0.1, 0.0), channels=tl3d.CH_TSDF)      # bench.py's grid
    orc = c_oracle.Oracle(cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], 0.1, 50.0, dims=spec.dims,
                          origin=spec.origin, voxel_size=spec.voxel_size, sdf_trunc=spec.sdf_trunc)
    orc.centroid = None
    with tl3d.FusionContext(cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], min_depth=0.1, max_depth=50.0,
                            n_slots=n, grid=spec) as ctx:
        host = []
        for i, (d, _) in enumerate(frames):
            if as_u16:                                                          # what a 16-bit PNG holds (D2R:85-90)
                mm = np.clip(np.round(d * 1000.0), 0, 65535).astype(np.uint16)
                ctx.upload(i, mm, None)
                host.append(mm.astype(np.float32) / np.float32(1000.0))
            else:
                ctx.upload(i, d, None)
                host.append(d)
        ctx.reset_stats()
        ctx.fuse_frames(list(range(n)), poses, [1.0] * n, centroid_subsample=0)  # one foreign call, as the bench's step
        st = ctx.stats()
        g = ctx.download_grid(tl3d.CH_TSDF)
    assert st["tsdf_launches"] == 2, st["tsdf_launches"]                        # 32 + 8
    for d, p in zip(host, poses):
        orc.tsdf_integrate(d, p[0], p[1])
    assert int((orc.tsdf[:, 1] > 0).sum()) > 15_000_000
    assert np.array_equal(g, orc.tsdf)


def _chain_drift(est, rel):
    worst_t = max(float(np.linalg.norm(np.asarray(t).reshape(3) - np.asarray(tg).reshape(3))) for (_, t), (_, tg) in zip(est, rel))
    worst_r = max(float(np.degrees(np.arccos(np.clip((np.trace(np.asarray(r) @ np.asarray(rg).T) - 1) / 2, -1, 1))))
                  for (r, _), (rg, _) in zip(est, rel))
    return worst_t, worst_r


def test_config3_full_length_85_frames_within_one_millimetre_of_the_reference_path():
    """BASELINE config 3 at its full 85 frames (8.4 m of travel): no frame dropped, chain within 2 mm / 0.05 degrees of the analytic
    dolly everywhere, fused cloud < 1 mm mean Chamfer from the restated reference CPU path on the same frames with the analytic
    poses (back-project -> vstack -> voxel centroid, D2R:328-410; DER-style, no outlier filter)."""
    W, H, n = 640, 480, 85
    cfg = ReconstructionConfig(fx=512.0, fy=512.0, cx=320.0, cy=240.0, voxel_size=0.005, subsample_factor=2, outlier_filter=False)
    cam = dict(width=W, height=H, fx=cfg.fx, fy=cfg.fy, cx=cfg.cx, cy=cfg.cy)
    scene = synth.corridor_scene()
    poses = synth.dolly_poses(n, (0.0, 0.0, 0.0), (0.0, 0.0, 0.1))
    frames = _render_on_device(scene, poses, cam, want_color=True)
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames([c for d, c in frames], [d for d, c in frames])
    pts, col, est = pipe.reconstruct()
    assert len(est) == n and all(r["status"] != 2 for r in pipe.icp_log)
    assert pipe.stats["points_dropped"] == 0 and pipe.stats["pool_refused"] == 0
    worst_t, worst_r = _chain_drift(est, poses)                                  # camera 0 is the world frame
    print(f"config 3, 85 frames: worst drift {worst_t * 1e3:.2f} mm / {worst_r:.4f} deg")
    assert worst_t < 2e-3 and worst_r < 0.05, (worst_t, worst_r)
    clouds = [rn.backproject(dd, cc, cfg.fx, cfg.fy, cfg.cx, cfg.cy, pose=p, subsample=2) for (dd, cc), p in zip(frames, poses)]
    ref_p, _ = rn.merge_open3d(clouds, cfg.voxel_size, sor=False)
    assert abs(len(pts) - len(ref_p)) < 0.05 * len(ref_p)
    ch = rn.chamfer_mean(pts, ref_p)
    print(f"config 3, 85 frames: mean Chamfer vs the restated reference path {ch * 1e3:.3f} mm ({len(pts)} / {len(ref_p)} points)")
    assert ch < 1e-3, ch                                                         # north-star bar


def test_config4_256_frames_icp_every_frame_within_one_millimetre_of_the_reference_path():
    """BASELINE config 4 (cylinder + ground orbit, 1280x720, 0.36 degrees per frame, ICP every frame) over 256 frames = 92 degrees
    of the orbit: chain drift bounded, and the cloud of the first 100 frames < 1 mm mean Chamfer from the restated reference path
    on those frames (analytic poses; with the outlier filter, D2R:412-415)."""
    W, H, n, ref_n = 1280, 720, 256, 100
    cfg = ReconstructionConfig(fx=1000.0, fy=1000.0, cx=640.0, cy=360.0, voxel_size=0.01, subsample_factor=4, max_depth=4.0)
    cam = dict(width=W, height=H, fx=cfg.fx, fy=cfg.fy, cx=cfg.cx, cy=cfg.cy)
    scene = synth.cylinder_scene(ground=True)
    poses = synth.orbit_poses(n, 1.5, 0.36, height=-0.2)
    r0, t0 = poses[0]
    rel = [(r @ r0.T, t.reshape(3, 1) - (r @ r0.T) @ t0.reshape(3, 1)) for r, t in poses]
    frames = _render_on_device(scene, poses, cam, want_color=True)
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames([c for d, c in frames], [d for d, c in frames])
    pts, col, est = pipe.reconstruct()
    assert len(est) == n and all(r["status"] != 2 for r in pipe.icp_log)
    worst_t, worst_r = _chain_drift(est, rel)
    print(f"config 4, 256 frames: worst drift {worst_t * 1e3:.2f} mm / {worst_r:.4f} deg over {0.36 * (n - 1):.0f} degrees of orbit")
    assert worst_t < 3e-3 and worst_r < 0.15, (worst_t, worst_r)
    assert len(pts) > 100_000
    # like for like: the pipeline on the first 100 frames (its own ICP poses) against the reference cloud of the same frames
    pipe2 = DepthToReconstructionPipeline(cfg)
    pipe2.set_frames([c for d, c in frames[:ref_n]], [d for d, c in frames[:ref_n]])
    pts2, _, est2 = pipe2.reconstruct()
    assert len(est2) == ref_n
    clouds = [rn.backproject(dd, cc, cfg.fx, cfg.fy, cfg.cx, cfg.cy, pose=p, subsample=cfg.subsample_factor, min_depth=cfg.min_depth,
                             max_depth=cfg.max_depth) for (dd, cc), p in zip(frames[:ref_n], rel[:ref_n])]
    ref_p, _ = rn.merge_open3d(clouds, cfg.voxel_size, sor=True)
    ch = rn.chamfer_mean(pts2, ref_p)
    print(f"config 4, first {ref_n} frames: mean Chamfer vs the restated reference path {ch * 1e3:.3f} mm")
    assert ch < 1e-3, ch
