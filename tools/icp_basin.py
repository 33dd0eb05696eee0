"""Convergence basin of frame-to-frame ICP without a motion prior: orbit steps of increasing size, single level vs a
coarse-to-fine schedule (wide gate / coarse sampling first).  Prints the pose error ||T_est - T_true||_F per case."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tl3d  # noqa: E402
from tl3d import synth  # noqa: E402

W, H = 640, 480
cam = dict(width=W, height=H, fx=525.0, fy=525.0, cx=320.0, cy=240.0)
SCENES = {"object+room": (synth.object_scene(True), 1.0), "cylinder+ground": (synth.cylinder_scene(True), 1.5)}
SCHEDULES = {
    "single 20x s2 d0.05": [(20, 2, 0.05)],
    "single 20x s2 d0.15": [(20, 2, 0.15)],
    "c2f 8x s8 d0.30 | 8x s4 d0.12 | 10x s2 d0.05": [(8, 8, 0.30), (8, 4, 0.12), (10, 2, 0.05)],
    "c2f 10x s4 d0.20 | 10x s2 d0.05": [(10, 4, 0.20), (10, 2, 0.05)],
}


def run(scene, radius, deg):
    poses = synth.orbit_poses(2, radius, deg)
    ctx = tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=2, max_depth=6.0)
    for i, p in enumerate(poses):
        d, _ = synth.render(scene, p, **cam, want_color=False)
        ctx.upload(i, d, None)
    ctx.build_normals(1)
    r_rel, t_rel = synth.relative_pose(poses[0], poses[1])
    T_true = np.eye(4)
    T_true[:3, :3], T_true[:3, 3] = r_rel, t_rel.ravel()
    out = []
    for sched in SCHEDULES.values():
        T = np.eye(4)
        for iters, stride, md in sched:
            T = ctx.icp(0, 1, T_init=T, iters=iters, stride=stride, max_dist=md)["T"]
        out.append(f"{np.linalg.norm(T - T_true):9.2e}")
    ctx.close()
    return float(np.linalg.norm(t_rel)), out


for sname, (scene, radius) in SCENES.items():
    print(sname)
    for deg in (4.0, 8.0, 12.0, 16.0, 20.0, 30.0):
        tn, out = run(scene, radius, deg)
        print(f"  step {deg:5.1f} deg (|t| = {tn:.3f} m): " + "  ".join(out))
print("columns: " + " || ".join(SCHEDULES))
