import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import tl3d, torch
from tl3d import synth
from tl3d.config import ReconstructionConfig
from tl3d.fusion import FusionContext
dev = torch.device("cuda", 0)
W, H, n = 1280, 720, int(os.environ.get('NP', '1')) + 1
cfg = ReconstructionConfig(fx=1000.0, fy=1000.0, cx=640.0, cy=360.0, voxel_size=0.01, subsample_factor=4, max_depth=4.0)
scene, poses = synth.cylinder_scene(ground=True), synth.orbit_poses(200, 1.5, 0.36, height=-0.2)[:n]
ctx = FusionContext(W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.min_depth, cfg.max_depth, n_slots=n, grid=None)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, xp=torch, device=dev)
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    ctx.upload(i, d.contiguous(), c.contiguous()); ctx.build_normals(i)
ctx.sync()
levels = [tuple(l) for l in cfg.icp_coarse] + [(cfg.icp_iters, cfg.icp_stride, cfg.icp_max_dist)]
lv = [dict(iters=l[0], stride=l[1], max_dist=l[2], damping=cfg.icp_damping, eig_rel=cfg.icp_eig_rel, eps=cfg.icp_eps) for l in levels]
for rep in range(3):
    t = time.perf_counter()
    res = ctx.icp_batch([(i, i + 1) for i in range(n - 1)], lv)
    print("ms", 1e3 * (time.perf_counter() - t), res[0]["iters_run"])
ctx.close()
rows = [list(map(int, l.split())) for l in open(os.environ["TL3D_ICP_TRACE"])]
M = max(r[0] for r in rows) + 1
print("members", M, "(100 MHz ticks -> us = /100)")
# slots: 0 pass start, 1 T loaded, 2 accumulate done, 3 sample loop done, 4 wave reduction done, 5 arrived, 6 barrier passed, 7 last flag
for g in range(16):
    rr = [r for r in rows if r[1] == g]
    if not rr: break
    a = np.array([r[3:] for r in rr], float) / 100.0
    last = np.array([r[2] for r in rr]) == 1
    l = a[last][0]
    o = a[~last] if (~last).any() else a
    print(f"pass {g}: start {a[:,0].min():.1f}..{a[:,0].max():.1f} | T {np.mean(a[:,1]-a[:,0]):.2f} | samples {np.mean(a[:,3]-a[:,1]):.2f} (max {np.max(a[:,3]-a[:,1]):.2f}) | wave-reduce {np.mean(a[:,4]-a[:,3]):.2f} | block-reduce {np.mean(a[:,2]-a[:,4]):.2f} | arrive {np.mean(a[:,5]-a[:,2]):.2f} | "
          f"last arrived {l[5]:.1f}, finish+publish {l[6]-l[5]:.2f} | others pass the barrier at {np.mean(o[:,6]):.1f} (max {o[:,6].max():.1f})")
