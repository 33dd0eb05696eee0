#!/usr/bin/env python3
"""Batched registration alone (row a10), 1080x1920 frames of the headline orbit resident in HBM: every pair (k-1, k) in ONE
launch of the persistent kernel.  Prints us per pair-iteration of a single-level stride-4 launch (what bench.py's
rows.icp_batch_us_per_pair_iteration is), of the pipeline's two-level schedule, and of stride 2 and 1, for raw-depth normals
(radius 0) and window-averaged ones (radius 1: the pipeline's default).

    python tools/bench_icp.py [--frames 256] [--radius 0,1] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tl3d  # noqa: E402
from tl3d import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--radius", default="0,1")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--json", default=None)
ap.add_argument("--cases", default="", help="comma-separated substrings: only the cases whose name holds one of them")
args = ap.parse_args()
W, H = 1080, 1920
cam = dict(fx=1719.0, fy=1719.0, cx=540.0, cy=960.0)
N = args.frames
scene = synth.object_scene(True)
poses = synth.orbit_poses(512, 1.0, 0.7)[:N]
dev = torch.device("cuda", 0)
ctx = tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=N, grid=None)
for i, p in enumerate(poses):
    d, c = synth.render(scene, p, W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], xp=torch, device=dev)
    d, c = d.contiguous(), c.contiguous()
    torch.cuda.synchronize()
    ctx.upload(i, d, c)
    ctx.sync()
T_rel = []
for k in range(N):
    r_rel, t_rel = synth.relative_pose(poses[(k - 1) % N], poses[k])
    T0 = np.eye(4)
    T0[:3, :3], T0[:3, 3] = r_rel, t_rel.ravel()
    T_rel.append(T0)
pairs = [(k - 1, k) for k in range(1, N)]
Tp = T_rel[1:]
out = {}
for radius in [int(x) for x in args.radius.split(",")]:
    ctx.set_normal_smoothing(radius)
    for i in range(N):
        ctx.build_normals(i)
    ctx.sync()
    for name, levels, init, passes in (
            ("stride 4 x %d" % args.iters, [dict(iters=args.iters, stride=4, max_dist=0.05, eps=0.0)], Tp, args.iters + 1),
            ("stride 2 x %d" % args.iters, [dict(iters=args.iters, stride=2, max_dist=0.05, eps=0.0)], Tp, args.iters + 1),
            ("stride 1 x 4", [dict(iters=4, stride=1, max_dist=0.05, eps=0.0)], Tp, 5),
            ("two-level from identity", [dict(iters=10, stride=4, max_dist=0.2, eps=1e-7), dict(iters=15, stride=2, max_dist=0.05, eps=1e-7)], None, None),
            ("Sim(3) stride 4 x %d" % args.iters, [dict(iters=args.iters, stride=4, max_dist=0.05, eps=0.0, estimate_scale=True)], Tp, args.iters + 1)):
        if args.cases and not any(c in name for c in args.cases.split(",")):
            continue
        res = ctx.icp_batch(pairs, levels, T_init=init)
        ctx.sync()
        ts = []
        ref_T = np.stack([r["T"] for r in res])
        ref_stats = np.array([[r["rmse"], r["fitness"], r["n_corr"], r["n_src"], r["iters_run"], r["status"]] for r in res])
        mismatches = 0
        stat_mismatches = 0
        worst = None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            res = ctx.icp_batch(pairs, levels, T_init=init)
            ts.append(time.perf_counter() - t0)
            mismatches += int(not np.array_equal(np.stack([r["T"] for r in res]), ref_T))      # every launch must give the same bits
            st_now = np.array([[r["rmse"], r["fitness"], r["n_corr"], r["n_src"], r["iters_run"], r["status"]] for r in res])
            if not np.array_equal(st_now, ref_stats):
                stat_mismatches += 1
                bad = np.nonzero(np.any(st_now != ref_stats, axis=1))[0]
                worst = dict(pairs=bad[:8].tolist(), now=st_now[bad[0]].tolist(), first=ref_stats[bad[0]].tolist())
        t = sorted(ts)[len(ts) // 2]
        key = f"radius {radius}, {name}"
        row = dict(pairs_per_s=round(len(pairs) / t, 1), ms=round(1e3 * t, 3))
        if passes:
            row["us_per_pair_iteration"] = round(1e6 * t / len(pairs) / passes, 3)
        row["rmse_mean"] = float(np.mean([r["rmse"] for r in res]))
        row["iters_mean"] = float(np.mean([r["iters_run"] for r in res]))
        row["T_digest"] = float(np.sum([np.abs(r["T"]).sum() for r in res]))
        row["launches_differing_from_the_first"] = mismatches
        row["launches_with_other_statistics"] = stat_mismatches
        if worst:
            row["example"] = worst
        out[key] = row
        print(f"{key:44s} {row}", flush=True)
out["stats"] = {k: v for k, v in ctx.stats().items() if k.startswith("icp")}
print(json.dumps(out))
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
