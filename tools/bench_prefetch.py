#!/usr/bin/env python3
"""Row f2 alone: files -> decode threads -> pinned ring -> tl3d_upload_frame_async, no fusion.  Sweeps (workers, ring size) of
fileio.FramePrefetcher over a synthetic 1080x1920 sequence (JPEG colour + .npy or 16-bit PNG depth, D2R:439-477 / DP:905-921).
    python tools/bench_prefetch.py [frames=384]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tl3d  # noqa: E402
from tl3d import fileio, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
DISTINCT = 48
W, H = 1080, 1920
fx = fy = 1719.0
scene = synth.object_scene(True)
poses = synth.orbit_poses(DISTINCT, 1.0, 360.0 / DISTINCT)
root = tempfile.mkdtemp(prefix="tl3d_pf_")
try:
    from PIL import Image
    for i, p in enumerate(poses):
        d, c = synth.render(scene, p, W, H, fx, fy, 540.0, 960.0)
        Image.fromarray(c[..., ::-1]).save(os.path.join(root, f"c{i}.jpg"), quality=90)
        np.save(os.path.join(root, f"d{i}.npy"), d)
        Image.fromarray(np.clip(d * 1000, 0, 65535).astype(np.uint16)).save(os.path.join(root, f"d{i}.png"))
    ctx = tl3d.FusionContext(W, H, fx, fy, 540.0, 960.0, n_slots=N, grid=None)
    cpus = fileio.usable_cpus()
    print("usable cpus", cpus, flush=True)
    for kind in ("npy", "png"):
        rgb = [os.path.join(root, f"c{i % DISTINCT}.jpg") for i in range(N)]
        dep = [os.path.join(root, f"d{i % DISTINCT}.{kind}") for i in range(N)]
        for workers, ring in ((0, 0), (0, 0), (14, 16), (16, 36), (12, 28)):
            t0 = time.perf_counter()
            pre = fileio.FramePrefetcher(ctx, rgb, dep, n_staging=ring, workers=workers)
            t1 = time.perf_counter()
            for _ in pre:
                pass
            ctx.sync()
            t2 = time.perf_counter()
            pre.close()
            t3 = time.perf_counter()
            print(f"{kind} workers {workers:2d} ring {ring:2d}: {N / (t2 - t1):7.1f} frames/s in the loop, {N / (t3 - t0):7.1f} with ring set-up {1e3 * (t1 - t0):5.0f} ms and "
                  f"tear-down {1e3 * (t3 - t2):5.0f} ms; decode {1e3 * pre.decode_s / N:5.2f} ms per frame and thread, mean busy threads {pre.decode_s / (t2 - t1):4.1f}", flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
