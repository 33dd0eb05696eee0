#!/usr/bin/env python3
"""End-to-end rates INCLUDING file decode and PCIe upload (never the headline `value`): write a synthetic 1080x1920
sequence to disk as the reference's depth producer would (depth_processor.py:905-921), then time
files -> [decode -> upload] -> fuse with known poses, for the in-memory loader and the streaming prefetcher."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tl3d  # noqa: E402
from tl3d import fileio, synth  # noqa: E402
from tl3d.config import ReconstructionConfig  # noqa: E402
from tl3d.pipeline import DepthToReconstructionPipeline  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
W, H = 1080, 1920
cfg = ReconstructionConfig(grid_dim=512, subsample_factor=2)
scene = synth.object_scene(True)
poses = synth.orbit_poses(N, 1.0, 360.0 / N)
r0, t0 = poses[0]
rel = [(r @ r0.T, t.reshape(3, 1) - (r @ r0.T) @ t0.reshape(3, 1)) for r, t in poses]
root = tempfile.mkdtemp(prefix="tl3d_e2e_")
try:
    from PIL import Image
    for kind in ("npy", "png"):
        rgb, dep = os.path.join(root, kind, "rgb"), os.path.join(root, kind, "depth")
        os.makedirs(rgb); os.makedirs(dep)
        for i, p in enumerate(poses):
            d, c = synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy)
            Image.fromarray(c[..., ::-1]).save(os.path.join(rgb, f"frame_{i:04d}.jpg"), quality=90)
            if kind == "npy":
                np.save(os.path.join(dep, f"frame_{i:04d}_depth.npy"), d)
            else:
                Image.fromarray(np.clip(d * 1000, 0, 65535).astype(np.uint16)).save(os.path.join(dep, f"frame_{i:04d}_depth.png"))
    grid = tl3d.GridSpec.cube(512, 0.005, centre=(0.0, -0.1, 0.0))
    import contextlib, io
    for kind in ("npy", "png"):
        rgb, dep = os.path.join(root, kind, "rgb"), os.path.join(root, kind, "depth")
        for mode in ("in-memory", "streaming"):
            pipe = DepthToReconstructionPipeline(cfg)
            t0_ = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                n = (pipe.load_data if mode == "in-memory" else pipe.load_data_streaming)(rgb, dep)
                t1 = time.perf_counter()
                pts, col, _ = pipe.reconstruct(grid=grid, poses=rel)
            t2 = time.perf_counter()
            dec = getattr(pipe, "decode_stats", None)
            print(f"{kind:4s} {mode:10s}: load {t1 - t0_:6.2f} s, reconstruct {t2 - t1:6.2f} s, total {N / (t2 - t0_):7.1f} frames/s, "
                  f"{len(pts)} points; stages {pipe.timings}" + (f"; decode {dec}" if dec else ""), flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
