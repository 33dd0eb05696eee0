#!/usr/bin/env python3
"""Whole-pipeline repeatability: reconstruct() of the same frames several times in one process must give the same points, colours,
poses and registration log, bit for bit (integer accumulators, fixed-order sums; nothing may depend on which workgroup ran when).
    python tools/soak_pipeline.py [--reps 6]"""
import argparse
import contextlib
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tl3d  # noqa: E402,F401
import torch  # noqa: E402
from tl3d import synth  # noqa: E402
from tl3d.config import ReconstructionConfig  # noqa: E402
from tl3d.pipeline import DepthToReconstructionPipeline  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=6)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cases = [
    ("turntable 40 x 1080x1920", 1080, 1920, 40, ReconstructionConfig(voxel_size=0.005, subsample_factor=2), synth.object_scene(with_room=False), synth.orbit_poses(40, 1.0, 7.2)),
    ("orbit 200 x 1280x720", 1280, 720, 200, ReconstructionConfig(fx=1000.0, fy=1000.0, cx=640.0, cy=360.0, voxel_size=0.01, subsample_factor=4, max_depth=4.0),
     synth.cylinder_scene(ground=True), synth.orbit_poses(200, 1.5, 0.36, height=-0.2)),
]
bad = 0
for name, W, H, n, cfg, scene, poses in cases:
    images, depths = [], []
    for p in poses:
        d, c = synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, xp=torch, device=dev)
        depths.append(d.contiguous())
        images.append(c.contiguous())
    torch.cuda.synchronize()
    ref = None
    for rep in range(args.reps):
        pipe = DepthToReconstructionPipeline(cfg)
        pipe.set_frames(images, depths)
        with contextlib.redirect_stdout(io.StringIO()):
            pts, col, est = pipe.reconstruct()
        log = np.array([[e["rmse"], e["fitness"], e["n_corr"], e["iters_run"], e["status"]] for e in pipe.icp_log])
        now = (pts, col, np.stack([np.hstack([r, np.asarray(t).reshape(3, 1)]) for r, t in est]), log)
        if ref is None:
            ref = now
        same = [np.array_equal(a, b) for a, b in zip(now, ref)]
        if not all(same):
            bad += 1
        print(f"{name}: run {rep}: {len(pts)} points, {len(est)} cameras; equal to run 0: points {same[0]} colours {same[1]} poses {same[2]} registration log {same[3]}", flush=True)
print("REPEATABLE" if bad == 0 else f"{bad} runs differ")
sys.exit(1 if bad else 0)
