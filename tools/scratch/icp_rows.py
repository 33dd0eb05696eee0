"""bench's ICP rows alone (1080p, 32 resident frames)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, tl3d
from tl3d import synth
W, H, n = 1080, 1920, int(os.environ.get("NF", "32"))
fx = fy = 1719.0; cx, cy = 540.0, 960.0
dev = torch.device("cuda", 0)
scene, poses = synth.object_scene(with_room=True), synth.orbit_poses(n, 1.0, 360.0 / 512)
ctx = tl3d.FusionContext(W, H, fx, fy, cx, cy, n_slots=n, grid=None)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, W, H, fx, fy, cx, cy, xp=torch, device=dev)
    torch.cuda.synchronize()
    ctx.upload(i, d.contiguous(), c.contiguous()); ctx.build_normals(i)
ctx.sync()
T_rel = []
for k in range(n):
    r, t = synth.relative_pose(poses[(k - 1) % n], poses[k]); T0 = np.eye(4); T0[:3, :3], T0[:3, 3] = r, t.ravel(); T_rel.append(T0)
pairs = [((k - 1) % n, k) for k in range(n)]
def timed(fn, reps):
    fn(); ctx.sync(); t = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t) / reps
fixed = [dict(iters=10, stride=4, max_dist=0.05, eps=0.0)]
two = [dict(iters=10, stride=4, max_dist=0.2, eps=1e-7), dict(iters=15, stride=2, max_dist=0.05, eps=1e-7)]
t = timed(lambda: ctx.icp_batch(pairs, fixed, T_init=T_rel), 6); print(f"batch 10 x s4: {n/t:.0f} pairs/s, {1e6*t/n/11:.2f} us per pair-iteration")
t = timed(lambda: ctx.icp_batch(pairs, two), 6); print(f"two-level from identity: {n/t:.0f} pairs/s")
t = timed(lambda: ctx.icp_batch(pairs[1:2], fixed, T_init=T_rel[1:2]), 16); print(f"one pair: {1e6*t/11:.2f} us per iteration")
res = ctx.icp_batch(pairs, two)
print("checksum", float(sum(np.abs(r["T"]).sum() for r in res)), [r["iters_run"] for r in res[:8]])
