import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import tl3d, torch
from tl3d import synth
from tl3d.config import ReconstructionConfig
from tl3d.fusion import FusionContext
dev = torch.device("cuda", 0)
W, H, n = 1280, 720, 200
cfg = ReconstructionConfig(fx=1000.0, fy=1000.0, cx=640.0, cy=360.0, voxel_size=0.01, subsample_factor=4, max_depth=4.0)
scene, poses = synth.cylinder_scene(ground=True), synth.orbit_poses(n, 1.5, 0.36, height=-0.2)
ctx = FusionContext(W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.min_depth, cfg.max_depth, n_slots=n, grid=None)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, xp=torch, device=dev)
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    ctx.upload(i, d.contiguous(), c.contiguous()); ctx.build_normals(i)
ctx.sync()
levels = [tuple(l) for l in cfg.icp_coarse] + [(cfg.icp_iters, cfg.icp_stride, cfg.icp_max_dist)]
lv = [dict(iters=l[0], stride=l[1], max_dist=l[2], damping=cfg.icp_damping, eig_rel=cfg.icp_eig_rel, eps=cfg.icp_eps) for l in levels]
clk = time.perf_counter
for P in (1, 2, 4, 5, 8, 16, 32, 64, 128, 199):
    pairs = [(i - 1, i) for i in range(1, P + 1)]
    for rep in range(2):
        t = clk()
        try:
            res = ctx.icp_batch(pairs, lv)
            print(f"P={P}: {1e3*(clk()-t):.3f} ms = {1e6*(clk()-t)/P:.1f} us/pair", flush=True)
        except Exception as e:
            print(f"P={P}: FAILED after {1e3*(clk()-t):.1f} ms: {e}", flush=True)
            sys.exit(1)
ctx.close()
