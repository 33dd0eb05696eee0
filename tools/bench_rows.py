#!/usr/bin/env python3
"""Per-row timings of the hot path on one GPU, 1080x1920 frames resident in HBM.
Not the headline bench: this prints a small table (ms per call, derived GB/s on algorithmic bytes) for DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tl3d  # noqa: E402
from tl3d import synth  # noqa: E402

W, H = 1080, 1920
cam = dict(width=W, height=H, fx=1719.0, fy=1719.0, cx=540.0, cy=960.0)
N = 16
scene = synth.object_scene(True)
poses = synth.orbit_poses(N, 1.0, 0.7)
dev = torch.device("cuda", 0)
spec = tl3d.GridSpec.cube(512, 0.005, centre=(0.0, -0.1, 0.0))
ctx = tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=N, grid=spec)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], xp=torch, device=dev)
    d, c = d.contiguous(), c.contiguous()
    torch.cuda.synchronize()
    ctx.upload(i, d, c)
    ctx.sync()
rows = {}


def timeit(name, fn, reps, bytes_per_call=None):
    fn(0)
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(reps):
        fn(k)
    ctx.sync()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    rows[name] = dict(ms=round(ms, 4), per_s=round(1e3 / ms, 1))
    if bytes_per_call:
        rows[name]["GBps"] = round(bytes_per_call / ms / 1e6, 1)
    print(f"{name:42s} {ms:9.4f} ms   {1e3 / ms:10.1f} /s" + (f"   {bytes_per_call / ms / 1e6:8.1f} GB/s" if bytes_per_call else ""), flush=True)


px = W * H
timeit("tsdf integrate (F=1)", lambda k: ctx.integrate(k % N, poses[k % N]), 128)
timeit("centroid accumulate s=1", lambda k: ctx.accumulate_centroid(k % N, poses[k % N], subsample=1), 64, px * 7)
timeit("centroid accumulate s=2 (D2R default)", lambda k: ctx.accumulate_centroid(k % N, poses[k % N], subsample=2), 64, px * 7 / 4)
timeit("centroid accumulate s=4 (DER default)", lambda k: ctx.accumulate_centroid(k % N, poses[k % N], subsample=4), 64, px * 7 / 16)
timeit("build normals", lambda k: ctx.build_normals(k % N), 64, px * 20)
for i in range(N):
    ctx.build_normals(i)


def icp(k, iters, stride):
    a, b = k % (N - 1), k % (N - 1) + 1
    r_rel, t_rel = synth.relative_pose(poses[a], poses[b])
    T0 = np.eye(4)
    T0[:3, :3], T0[:3, 3] = r_rel, t_rel.ravel()
    return ctx.icp(a, b, T_init=T0, iters=iters, stride=stride, max_dist=0.05, eps=0.0)


for iters, stride in ((10, 4), (10, 2), (10, 1)):
    timeit(f"ICP {iters} iters stride {stride} (incl. read-back)", lambda k: icp(k, iters, stride), 32,
           iters * (px / stride ** 2) * 20)
def icp_lanes(k, iters, stride, lanes=tl3d.ICP_LANES):
    for l in range(lanes):
        a = (k * lanes + l) % (N - 1)
        r_rel, t_rel = synth.relative_pose(poses[a], poses[a + 1])
        T0 = np.eye(4)
        T0[:3, :3], T0[:3, 3] = r_rel, t_rel.ravel()
        ctx.icp_enqueue(l, a, a + 1, T_init=T0, iters=iters, stride=stride, max_dist=0.05, eps=0.0)
    return [ctx.icp_collect(l) for l in range(lanes)]


for iters, stride in ((10, 4), (10, 2)):
    name = f"ICP x{tl3d.ICP_LANES} lanes, {iters} iters stride {stride} ({tl3d.ICP_LANES} pairs)"
    timeit(name, lambda k: icp_lanes(k, iters, stride), 16, tl3d.ICP_LANES * iters * (px / stride ** 2) * 20)
    rows[name]["pairs_per_s"] = round(tl3d.ICP_LANES * rows[name]["per_s"], 1)
    print(f"   -> {rows[name]['pairs_per_s']} pairs/s")
res = icp(0, 10, 2)
print("   ICP result: fitness %.3f rmse %.2e iters %d" % (res["fitness"], res["rmse"], res["iters_run"]))
cap = -(-H // 1) * -(-W // 1)
xyz_d = torch.empty((cap, 3), dtype=torch.float32, device=dev)
rgb_d = torch.empty((cap, 3), dtype=torch.uint8, device=dev)
import ctypes as C
from tl3d import _cabi as abi
n_out = C.c_int64()


def bp(k, sub):
    r, t = poses[k % N]
    abi.check(ctx._lib.tl3d_backproject(ctx._h, k % N, abi.ptr(abi.d9(r)), abi.ptr(abi.d3(t)), 1.0, 0, sub, 0.1, 50.0,
                                        abi.ptr(xyz_d), abi.ptr(rgb_d), cap, C.byref(n_out)))


timeit("backproject s=1 -> device point list (blocking: count read back)", lambda k: bp(k, 1), 32, px * (7 + 15))
timeit("backproject s=2 -> device point list (blocking: count read back)", lambda k: bp(k, 2), 32, px * (7 + 15) / 4)
n_dev = torch.zeros(1, dtype=torch.int64, device=dev)
timeit("backproject s=1, device-only (one kernel, no read-back)", lambda k: ctx.backproject_device(k % N, xyz_d, rgb_d, n_dev, pose=poses[k % N], subsample=1), 64, px * (7 + 15))
timeit("backproject s=2, device-only (one kernel, no read-back)", lambda k: ctx.backproject_device(k % N, xyz_d, rgb_d, n_dev, pose=poses[k % N], subsample=2), 64, px * (7 + 15) / 4)
timeit("backproject s=4, device-only (one kernel, no read-back)", lambda k: ctx.backproject_device(k % N, xyz_d, rgb_d, n_dev, pose=poses[k % N], subsample=4), 64, px * (7 + 15) / 16)
t0 = time.perf_counter()
xyz, rgb = ctx.extract(tl3d.EXTRACT_CENTROID)
t_ext = time.perf_counter() - t0
print(f"extract centroid (512^3 -> host)            {1e3 * t_ext:9.2f} ms   {len(xyz)} points")
t0 = time.perf_counter()
x2, _ = ctx.extract(tl3d.EXTRACT_TSDF, min_weight=2)
print(f"extract tsdf zero crossings (-> host)        {1e3 * (time.perf_counter() - t0):9.2f} ms   {len(x2)} points")
t0 = time.perf_counter()
keep = ctx.statistical_outlier(xyz, 20, 2.0, cell_size=0.01)
t_sor = time.perf_counter() - t0
print(f"statistical outlier filter (k=20)            {1e3 * t_sor:9.2f} ms   kept {keep.sum()} of {len(xyz)}")
rows["extract_centroid_ms"] = round(1e3 * t_ext, 2)
rows["sor_ms"] = round(1e3 * t_sor, 2)
rows["points"] = int(len(xyz))
print(json.dumps(rows))
