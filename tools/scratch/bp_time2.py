import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tl3d
from tl3d import synth
W, H = 1080, 1920
cam = dict(width=W, height=H, fx=1719.0, fy=1719.0, cx=540.0, cy=960.0)
dev = torch.device("cuda", 0)
scene = synth.object_scene(True)
poses = synth.orbit_poses(4, 1.0, 0.7)
ctx = tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=4, grid=None)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], xp=torch, device=dev)
    torch.cuda.synchronize()
    ctx.upload(i, d.contiguous(), c.contiguous()); ctx.sync()
cap = W * H
xyz = torch.empty((cap, 3), dtype=torch.float32, device=dev); rgb = torch.empty((cap, 3), dtype=torch.uint8, device=dev)
n = torch.zeros(1, dtype=torch.int64, device=dev)
for sub in (1, 2):
    for rep in range(3):
        ctx.sync(); t = time.perf_counter()
        for k in range(64):
            ctx.backproject_device(k % 4, xyz, rgb, n, pose=poses[k % 4], subsample=sub)
        ctx.sync(); dt = (time.perf_counter() - t) / 64
    print(f"stride {sub}: {1e6*dt:.1f} us per call, n {int(n.item())}")
