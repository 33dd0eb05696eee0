#!/usr/bin/env python3
"""BASELINE.json configurations at FULL length through the product pipeline (one GPU), with the numbers the parity tests only
sample: pose-chain drift against the analytic trajectory, Chamfer of the fused cloud against the analytic scene and against the
restated reference CPU path, frames per second of the whole reconstruct() call.  One JSON object per config on stdout (and
appended to --out).

    python tools/run_config.py --config 2      # 50 frames 1080x1920 turntable of the sphere-union object, 5 mm voxels
    python tools/run_config.py --config 3      # 85 frames 640x480 corridor dolly, 10 cm per frame, 5 mm voxels
    python tools/run_config.py --config 4      # 1000 frames 1280x720 orbit of a textureless cylinder + ground, ICP every frame
    python tools/run_config.py --config 4b     # the same orbit around the PURE cylinder (rotation about its axis is unobservable)
Frames are ray-cast on the GPU with torch (plumbing) and handed to the pipeline as device tensors.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tl3d  # noqa: E402,F401
import torch  # noqa: E402
from oracle import ref_numpy as rn  # noqa: E402  (checker only: the reference CPU path the cloud is compared with)
from tl3d import synth  # noqa: E402
from tl3d.config import ReconstructionConfig  # noqa: E402
from tl3d.pipeline import DepthToReconstructionPipeline  # noqa: E402


def rel_to_first(poses):
    r0, t0 = poses[0]
    out = []
    for r, t in poses:
        rr = r @ r0.T
        out.append((rr, t.reshape(3, 1) - rr @ t0.reshape(3, 1)))
    return out


def scene_distance(scene, pts_world):
    """distance of points to the nearest analytic surface of the scene"""
    d = np.full(len(pts_world), np.inf)
    for c, r in scene.spheres:
        d = np.minimum(d, np.abs(np.linalg.norm(pts_world - np.asarray(c), axis=1) - r))
    for nrm, dd in scene.planes:
        d = np.minimum(d, np.abs(pts_world @ np.asarray(nrm) - dd))
    if scene.room is not None:
        lo, hi = scene.room
        for a in range(3):
            d = np.minimum(d, np.minimum(np.abs(pts_world[:, a] - lo[a]), np.abs(pts_world[:, a] - hi[a])))
    for c, r, hgt in scene.cylinders:
        rad = np.hypot(pts_world[:, 0] - c[0], pts_world[:, 2] - c[2])
        side = np.abs(rad - r)
        caps = np.minimum(np.abs(pts_world[:, 1] - (c[1] - 0.5 * hgt)), np.abs(pts_world[:, 1] - (c[1] + 0.5 * hgt)))
        d = np.minimum(d, np.where(rad <= r, np.minimum(side, caps), side))
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=["2", "3", "4", "4b"])
    ap.add_argument("--frames", type=int, default=0, help="override the sequence length")
    ap.add_argument("--ref-frames", type=int, default=0, help="frames of the restated reference CPU path to compare with (0 = config default)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--smooth-radius", type=int, default=-1, help="icp_smooth_radius (default: the configuration's)")
    ap.add_argument("--depth-noise-mm", type=float, default=0.0, help="Gaussian depth noise added to every frame")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.config == "2":
        W, H, n = 1080, 1920, args.frames or 50
        cfg = ReconstructionConfig(voxel_size=0.005, subsample_factor=2)                       # reference defaults 1719 / 540 / 960
        scene, poses = synth.object_scene(with_room=False), synth.orbit_poses(n, 1.0, 7.2)
        ref_n, name = args.ref_frames or n, "config 2: sphere-union turntable, 50 x 1080x1920, 5 mm"
    elif args.config == "3":
        W, H, n = 640, 480, args.frames or 85
        cfg = ReconstructionConfig(fx=512.0, fy=512.0, cx=320.0, cy=240.0, voxel_size=0.005, subsample_factor=2, outlier_filter=False)
        scene, poses = synth.corridor_scene(), synth.dolly_poses(n, (0.0, 0.0, 0.0), (0.0, 0.0, 0.1))
        ref_n, name = args.ref_frames or n, "config 3: corridor dolly, 85 x 640x480, 10 cm per frame, 5 mm"
    else:
        W, H, n = 1280, 720, args.frames or 1000
        cfg = ReconstructionConfig(fx=1000.0, fy=1000.0, cx=640.0, cy=360.0, voxel_size=0.01, subsample_factor=4, max_depth=4.0)
        scene = synth.cylinder_scene(ground=args.config == "4")
        poses = synth.orbit_poses(n, 1.5, 0.36, height=-0.2)
        ref_n = args.ref_frames or 100
        name = ("config 4: cylinder + ground orbit" if args.config == "4" else "config 4b: PURE cylinder orbit (degenerate)") + \
            f", {n} x 1280x720, 0.36 deg per frame, ICP every frame, 10 mm"
    if args.smooth_radius >= 0:
        cfg.icp_smooth_radius = args.smooth_radius
        name += f", icp_smooth_radius {args.smooth_radius}"
    if args.depth_noise_mm > 0:
        name += f", {args.depth_noise_mm} mm depth noise"
    rel = rel_to_first(poses)
    gen = torch.Generator(device=dev).manual_seed(7)
    t0 = time.perf_counter()
    images, depths, host = [], [], []
    for i, p in enumerate(poses):
        d, c = synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, xp=torch, device=dev)
        if args.depth_noise_mm > 0:
            d = torch.where(d > 0, d + 1e-3 * args.depth_noise_mm * torch.randn(d.shape, device=dev, generator=gen, dtype=d.dtype), d)
        depths.append(d.contiguous())
        images.append(c.contiguous())
        if i < ref_n:
            host.append((d.cpu().numpy(), c.cpu().numpy()))
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t0
    valid_frac = float(np.mean([(h[0] > cfg.min_depth).mean() for h in host[:8]]))
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames(images, depths)
    import contextlib
    import io
    log = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(log):
        pts, col, est = pipe.reconstruct()
    t_first = time.perf_counter() - t0
    first_timings = pipe.timings
    # the same call again in the same process: the first one also pays for what a process pays once (code objects loaded on first
    # launch, the allocator's first large blocks, registration buffers) -- both are reported
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames(images, depths)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(log):
        pts, col, est = pipe.reconstruct()
    t_rec = time.perf_counter() - t0
    second_timings = pipe.timings
    # ... and a third time (the second call still pays for what the first context's destruction left to do: its buffers return to
    # the library's caches while the second context is being built)
    pipe = DepthToReconstructionPipeline(cfg)
    pipe.set_frames(images, depths)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(log):
        pts, col, est = pipe.reconstruct()
    t_third = time.perf_counter() - t0
    occupancy = [l.strip() for l in log.getvalue().splitlines() if "Occupancy" in l][-1:]
    # (the second call is not always the faster one: with 1000 resident 720p frames it re-allocates 25 GB the first context has
    # just freed, and the driver hands such memory back at ~26 GB/s -- config 4's second registration stage takes 0.7 s for that
    # reason alone; both calls are in the record, the headline figure is the better of the two)
    out = dict(config=name, frames=n, width=W, height=H, valid_pixel_fraction=round(valid_frac, 3), render_s=round(t_render, 1),
               reconstruct_s=round(min(t_rec, t_first, t_third), 3), frames_per_s_whole_pipeline=round(n / min(t_rec, t_first, t_third), 1),
               stage_s=pipe.timings if t_third <= min(t_rec, t_first) else (second_timings if t_rec <= t_first else first_timings),
               first_call_in_process=dict(reconstruct_s=round(t_first, 3), frames_per_s=round(n / t_first, 1), stage_s=first_timings),
               second_call_in_process=dict(reconstruct_s=round(t_rec, 3), frames_per_s=round(n / t_rec, 1), stage_s=second_timings),
               third_call_in_process=dict(reconstruct_s=round(t_third, 3), frames_per_s=round(n / t_third, 1), stage_s=pipe.timings),
               layout=occupancy[0] if occupancy else "dense without asking (small grid)")
    if pts is None:
        out["error"] = "reconstruction failed"
        print(json.dumps(out))
        return 1
    kept = pipe.frame_index
    out.update(cameras=len(est), dropped_frames=n - len(est), points=int(len(pts)), grid=list(pipe.grid.dims), stats=pipe.stats,
               icp_iterations_mean=round(float(np.mean([e["iters_run"] for e in pipe.icp_log])), 2),
               icp_rmse_mm_mean=round(1e3 * float(np.mean([e["rmse"] for e in pipe.icp_log])), 4))
    # drift of the chained poses against the analytic trajectory (both relative to camera 0)
    drift = {}
    for k in sorted(set([len(kept) // 10 * j for j in range(1, 10)] + [len(kept) - 1])):
        if 0 < k < len(kept):
            (r, t), (rg, tg) = est[k], rel[kept[k]]
            ang = float(np.degrees(np.arccos(np.clip((np.trace(r @ rg.T) - 1) / 2, -1, 1))))
            drift[int(kept[k])] = dict(rot_deg=round(ang, 4), trans_mm=round(1e3 * float(np.linalg.norm(t.reshape(3) - tg.reshape(3))), 3))
    out["drift_vs_analytic"] = drift
    # the fused cloud against the analytic scene (world = camera-0 frame)
    r0, t0v = poses[0]
    pw = (pts - t0v.reshape(1, 3)) @ r0
    d = scene_distance(scene, pw)
    out["distance_to_analytic_surface_mm"] = dict(mean=round(1e3 * float(d.mean()), 3), p50=round(1e3 * float(np.percentile(d, 50)), 3),
                                                   p99=round(1e3 * float(np.percentile(d, 99)), 3))
    # against the restated reference CPU path on the first ref_n frames with the ANALYTIC poses (D2R:328-420)
    tr = time.perf_counter()
    clouds = [rn.backproject(dd, cc, cfg.fx, cfg.fy, cfg.cx, cfg.cy, pose=p, subsample=cfg.subsample_factor, min_depth=cfg.min_depth,
                             max_depth=cfg.max_depth) for (dd, cc), p in zip(host, rel[:ref_n])]
    ref_p, _ = rn.merge_open3d(clouds, cfg.voxel_size, sor=cfg.outlier_filter)
    t_ref = time.perf_counter() - tr
    if ref_n == n:
        out["chamfer_vs_reference_cpu_path_mm"] = round(1e3 * rn.chamfer_mean(pts, ref_p), 4)
    else:
        # like for like: the product pipeline again on just those frames (its own ICP poses), symmetric Chamfer against the
        # reference cloud of the same frames -- two voxelisations of different frame sets would sit on different lattices
        pipe2 = DepthToReconstructionPipeline(cfg)
        pipe2.set_frames(images[:ref_n], depths[:ref_n])
        with contextlib.redirect_stdout(log):
            pts2, _, _ = pipe2.reconstruct()
        out[f"chamfer_vs_reference_cpu_path_first_{ref_n}_frames_mm"] = round(1e3 * rn.chamfer_mean(pts2, ref_p), 4)
    out["reference_cpu_path"] = dict(frames=ref_n, points=int(len(ref_p)), seconds=round(t_ref, 1), frames_per_s=round(ref_n / t_ref, 2))
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
