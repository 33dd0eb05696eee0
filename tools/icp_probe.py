"""Run a few ICP calls at 1080p (for rocprofv3 kernel traces)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tl3d
from tl3d import synth
W, H = 1080, 1920
cam = dict(width=W, height=H, fx=1719.0, fy=1719.0, cx=540.0, cy=960.0)
scene = synth.object_scene(True)
poses = synth.orbit_poses(2, 1.0, 0.7)
ctx = tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=2)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, **cam, want_color=False)
    ctx.upload(i, d, None)
ctx.build_normals(1)
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for k in range(8):
    r = ctx.icp(0, 1, iters=10, stride=stride, max_dist=0.05, eps=0.0)
print(r["fitness"], r["rmse"], r["iters_run"])
