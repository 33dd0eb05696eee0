"""Where the registration stage of reconstruct() spends its time (config 2 / config 4 shapes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import tl3d
import torch
from tl3d import synth, _cabi as abi
from tl3d.config import ReconstructionConfig
from tl3d.fusion import FusionContext

which = sys.argv[1] if len(sys.argv) > 1 else "2"
dev = torch.device("cuda", 0)
if which == "2":
    W, H, n = 1080, 1920, 50
    cfg = ReconstructionConfig(voxel_size=0.005, subsample_factor=2)
    scene, poses = synth.object_scene(with_room=False), synth.orbit_poses(n, 1.0, 7.2)
else:
    W, H, n = 1280, 720, 200
    cfg = ReconstructionConfig(fx=1000.0, fy=1000.0, cx=640.0, cy=360.0, voxel_size=0.01, subsample_factor=4, max_depth=4.0)
    scene, poses = synth.cylinder_scene(ground=True), synth.orbit_poses(n, 1.5, 0.36, height=-0.2)
depths, images = [], []
for p in poses:
    d, c = synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, xp=torch, device=dev)
    depths.append(d.contiguous()); images.append(c.contiguous())
torch.cuda.synchronize()
clk = time.perf_counter
t = clk()
ctx = FusionContext(W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.min_depth, cfg.max_depth, n_slots=n, grid=None)
print("create", round(1e3 * (clk() - t), 2), "ms")
t = clk()
for i in range(n):
    ctx.upload(i, depths[i], images[i])
ctx.sync(); print("upload", round(1e3 * (clk() - t), 2), "ms")
t = clk()
for i in range(n):
    ctx.build_normals(i, scale=1.0)
ctx.sync(); print("normals", round(1e3 * (clk() - t), 2), "ms  per frame us", round(1e6 * (clk() - t) / n, 1))
levels = [tuple(l) for l in cfg.icp_coarse] + [(cfg.icp_iters, cfg.icp_stride, cfg.icp_max_dist)]
lanes = abi.ICP_LANES
for rep in range(2):
    i = 1
    T_all = clk()
    while i < n:
        batch = list(range(i, min(n, i + lanes)))
        T0s = [np.eye(4)] * len(batch)
        for li, lv in enumerate(levels):
            t = clk()
            for k, cur in enumerate(batch):
                ctx.icp_enqueue(k, cur - 1, cur, T_init=T0s[k], scale_src=1.0, iters=lv[0], stride=lv[1], max_dist=lv[2], damping=cfg.icp_damping, eig_rel=cfg.icp_eig_rel, eps=cfg.icp_eps)
            t1 = clk()
            res = [ctx.icp_collect(k) for k in range(len(batch))]
            t2 = clk()
            T0s = [r["T"] for r in res]
            if i < 1 + 3 * lanes:
                print(f"rep {rep} batch@{i} level {li}: enqueue {1e3*(t1-t):.2f} ms collect {1e3*(t2-t1):.2f} ms iters {[r['iters_run'] for r in res]}")
        i = batch[-1] + 1
    print(f"rep {rep}: all pairs {1e3*(clk()-T_all):.2f} ms = {1e6*(clk()-T_all)/(n-1):.1f} us/pair")
ctx.close()

# ---- batched: all pairs, all levels, one launch
ctx = FusionContext(W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.min_depth, cfg.max_depth, n_slots=n, grid=None)
for i in range(n):
    ctx.upload(i, depths[i], images[i])
    ctx.build_normals(i, scale=1.0)
ctx.sync()
lv = [dict(iters=l[0], stride=l[1], max_dist=l[2], damping=cfg.icp_damping, eig_rel=cfg.icp_eig_rel, eps=cfg.icp_eps) for l in levels]
pairs = [(i - 1, i) for i in range(1, n)]
for rep in range(3):
    t = clk()
    ctx.icp_batch_enqueue(pairs, lv)
    t1 = clk()
    res = ctx.icp_batch_collect()
    t2 = clk()
    print(f"batch rep {rep}: enqueue {1e3*(t1-t):.2f} ms collect {1e3*(t2-t1):.2f} ms = {1e6*(t2-t)/(n-1):.1f} us/pair  iters {[r['iters_run'] for r in res[:16]]} status {[r['status'] for r in res[:16]]}")
for rep in range(2):
    t = clk()
    r1 = ctx.icp_batch(pairs[:1], lv)
    print(f"single pair through the batch kernel: {1e6*(clk()-t):.1f} us, iters {r1[0]['iters_run']}")
ctx.close()
