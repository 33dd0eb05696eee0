#!/usr/bin/env python3
"""Back-projection to a device point list (rows a4/a5), 1080x1920 frames resident in HBM: us per frame and GB/s on
(4 + 3) B per sample read + 15 B per point written, strides 1, 2 (the reference's default, D2R:65) and 4.
    python tools/bench_bp.py          (experiments flavour: TL3D_BP_TILE=512|1024|2048 pins the tile size)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tl3d  # noqa: E402
from tl3d import synth  # noqa: E402

W, H = 1080, 1920
N = 16
scene = synth.object_scene(True)
poses = synth.orbit_poses(N, 1.0, 0.7)
dev = torch.device("cuda", 0)
ctx = tl3d.FusionContext(W, H, 1719.0, 1719.0, 540.0, 960.0, n_slots=N, grid=None)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, W, H, 1719.0, 1719.0, 540.0, 960.0, xp=torch, device=dev)
    d, c = d.contiguous(), c.contiguous()
    torch.cuda.synchronize()
    ctx.upload(i, d, c)
    ctx.sync()
xyz = torch.empty((H * W, 3), dtype=torch.float32, device=dev)
rgb = torch.empty((H * W, 3), dtype=torch.uint8, device=dev)
n_d = torch.zeros(1, dtype=torch.int64, device=dev)
out = []
for sub in (1, 2, 4):
    def go(k):
        ctx.backproject_device(k % N, xyz, rgb, n_d, pose=poses[k % N], subsample=sub)
    go(0)
    ctx.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for k in range(64):
            go(k)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / 64)
    npts = int(n_d.item())
    by = (4 + 3) * ((W + sub - 1) // sub) * ((H + sub - 1) // sub) + 15 * npts
    out.append(f"s{sub}: {1e6 * best:6.2f} us {by / best / 1e9:7.1f} GB/s")
print(os.environ.get("TL3D_BP_TILE", "default tile"), " | ".join(out), flush=True)
