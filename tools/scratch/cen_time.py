"""centroid accumulation alone: us per frame at stride 1 / 2 (1080x1920 bench scene, 512^3 @ 5 mm)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import tl3d, torch
from tl3d import synth
from tl3d.fusion import FusionContext, GridSpec
dev = torch.device("cuda", 0)
W, H, n = 1080, 1920, 16
fx = fy = 1719.0; cx, cy = 540.0, 960.0
scene, poses = synth.object_scene(with_room=True), synth.orbit_poses(n, 1.0, 7.2)
grid = GridSpec((512, 512, 512), (-1.28, -1.48, -1.28), 0.005, 0.02, tl3d.CH_CENTROID)
ctx = FusionContext(W, H, fx, fy, cx, cy, 0.1, 50.0, n_slots=n, grid=grid)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, W, H, fx, fy, cx, cy, xp=torch, device=dev)
    torch.cuda.synchronize()
    ctx.upload(i, d.contiguous(), c.contiguous())
ctx.sync()
for sub in (1, 2):
    for rep in range(3):
        ctx.sync(); t = time.perf_counter()
        for k in range(64):
            ctx.accumulate_centroid(k % n, poses[k % n], subsample=sub)
        ctx.sync()
        dt = (time.perf_counter() - t) / 64
    st = ctx.stats()
    print(f"stride {sub}: {1e6*dt:.1f} us per frame; points {st['centroid_points']} dropped {st['centroid_dropped']}")
g = ctx.download_grid(tl3d.CH_CENTROID)
import hashlib
print("grid sha1", hashlib.sha1(g.tobytes()).hexdigest(), "points in grid", int((g[:, 1] >> np.uint64(32)).sum()), "occupied", int(((g[:, 1] >> np.uint64(32)) > 0).sum()))
ctx.reset()
ctx.accumulate_centroid(0, poses[0], subsample=1)
g = ctx.download_grid(tl3d.CH_CENTROID)
print("one frame: sha1", hashlib.sha1(g.tobytes()).hexdigest(), "points", int((g[:, 1] >> np.uint64(32)).sum()), "stats", ctx.stats())
ctx.close()
