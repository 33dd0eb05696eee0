"""Analytic synthetic depth sequences.

No depth maps, poses or .ply outputs ship with the reference, and its depth network
(Depth-Anything, depth_enhanced_reconstruction.py:114-118) is fetched by name, so every
configuration runs on ray-cast scenes with known poses.  The ray-caster is written
against an array namespace `xp` (numpy, or torch for on-GPU generation in bench.py --
plumbing, not the measured path).

Pose convention everywhere: world->camera  X_c = R X_w + t  (depth_to_reconstruction.py:373-376).
Depth is the camera-frame z of the first hit, 0 where the ray misses (invalid, dropped by the
reference's `depth > min_depth` test).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np


@dataclass
class Scene:
    spheres: List[Tuple[Tuple[float, float, float], float]] = field(default_factory=list)   # (centre, radius)
    planes: List[Tuple[Tuple[float, float, float], float]] = field(default_factory=list)    # n.x = d, hit from n side
    room: Optional[Tuple[Tuple[float, float, float], Tuple[float, float, float]]] = None    # (min, max) seen from inside
    cylinders: List[Tuple[Tuple[float, float, float], float, float]] = field(default_factory=list)  # (centre, r, h) axis = y


def look_at(eye, target, up=(0.0, -1.0, 0.0)):
    """world->camera (R, t) with +z forward, +x right, +y down (image v grows downwards)."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    upv = np.asarray(up, np.float64)
    r = np.cross(-upv, f)
    if np.linalg.norm(r) < 1e-9:
        r = np.cross(np.array([0.0, 0.0, 1.0]), f)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f], axis=0)
    t = -R @ eye
    return R, t.reshape(3, 1)


def orbit_poses(n, radius, deg_per_frame, target=(0.0, 0.0, 0.0), height=0.0, start_deg=0.0):
    poses = []
    for i in range(n):
        a = math.radians(start_deg + i * deg_per_frame)
        eye = (target[0] + radius * math.sin(a), target[1] + height, target[2] - radius * math.cos(a))
        poses.append(look_at(eye, target))
    return poses


def dolly_poses(n, start, step, yaw_deg_per_frame=0.0):
    """Camera translating by `step` per frame, looking along +z (tunnel / corridor, config 3; config 1)."""
    poses = []
    for i in range(n):
        eye = np.asarray(start, np.float64) + i * np.asarray(step, np.float64)
        a = math.radians(i * yaw_deg_per_frame)
        tgt = eye + np.array([math.sin(a), 0.0, math.cos(a)])
        poses.append(look_at(eye, tgt))
    return poses


def relative_pose(pose_prev, pose_curr):
    """(R_rel, t_rel) with X_curr = R_rel X_prev + t_rel, the quantity chained at D2R:619-620."""
    rp, tp = pose_prev
    rc, tc = pose_curr
    r_rel = rc @ rp.T
    return r_rel, tc.reshape(3, 1) - r_rel @ tp.reshape(3, 1)


def _where(xp, c, a, b):
    return xp.where(c, a, b)


def render(scene: Scene, pose, width, height, fx, fy, cx, cy, xp=np, device=None, noise_sigma=0.0, seed=0,
           want_color=True):
    """Ray-cast `scene` from `pose`.  Returns (depth f32[H,W], bgr u8[H,W,3])."""
    R, t = pose
    R = np.asarray(R, np.float64)
    t = np.asarray(t, np.float64).reshape(3)
    C = -R.T @ t
    is_torch = xp is not np
    if is_torch:
        kw = dict(dtype=xp.float64, device=device)
        u = xp.arange(width, **kw)[None, :]
        v = xp.arange(height, **kw)[:, None]
        big = xp.tensor(1e30, **kw)
    else:
        u = np.arange(width, dtype=np.float64)[None, :]
        v = np.arange(height, dtype=np.float64)[:, None]
        big = 1e30
    dxc = (u - cx) / fx + 0.0 * v
    dyc = (v - cy) / fy + 0.0 * u
    # world direction D = R^T (dxc, dyc, 1); camera z of C + s D is exactly s
    D = [R[0, i] * dxc + R[1, i] * dyc + R[2, i] for i in range(3)]
    best = dxc * 0.0 + big
    eps = 1e-6

    for (c, r) in scene.spheres:
        oc = [C[i] - c[i] for i in range(3)]
        a = D[0] * D[0] + D[1] * D[1] + D[2] * D[2]
        b = 2.0 * (D[0] * oc[0] + D[1] * oc[1] + D[2] * oc[2])
        cc = oc[0] ** 2 + oc[1] ** 2 + oc[2] ** 2 - r * r
        disc = b * b - 4.0 * a * cc
        ok = disc > 0
        sq = xp.sqrt(_where(xp, ok, disc, disc * 0.0))
        s1 = (-b - sq) / (2.0 * a)
        s2 = (-b + sq) / (2.0 * a)
        s = _where(xp, s1 > eps, s1, s2)
        hit = ok & (s > eps)
        best = _where(xp, hit & (s < best), s, best)

    for (nrm, d) in scene.planes:
        den = D[0] * nrm[0] + D[1] * nrm[1] + D[2] * nrm[2]
        num = d - (C[0] * nrm[0] + C[1] * nrm[1] + C[2] * nrm[2])
        ok = abs(den) > 1e-12
        s = num / _where(xp, ok, den, den * 0.0 + 1.0)
        hit = ok & (s > eps)
        best = _where(xp, hit & (s < best), s, best)

    if scene.room is not None:
        lo, hi = scene.room
        s_exit = dxc * 0.0 + big
        for i in range(3):
            di = D[i]
            nz = abs(di) > 1e-12
            dsafe = _where(xp, nz, di, di * 0.0 + 1.0)
            sa = (lo[i] - C[i]) / dsafe
            sb = (hi[i] - C[i]) / dsafe
            smax = _where(xp, sa > sb, sa, sb)
            s_exit = _where(xp, nz & (smax < s_exit), smax, s_exit)
        hit = s_exit > eps
        best = _where(xp, hit & (s_exit < best), s_exit, best)

    for (c, r, h) in scene.cylinders:
        ocx, ocz = C[0] - c[0], C[2] - c[2]
        a = D[0] * D[0] + D[2] * D[2]
        b = 2.0 * (D[0] * ocx + D[2] * ocz)
        cc = ocx * ocx + ocz * ocz - r * r
        disc = b * b - 4.0 * a * cc
        ok = (disc > 0) & (a > 1e-12)
        asafe = _where(xp, a > 1e-12, a, a * 0.0 + 1.0)
        sq = xp.sqrt(_where(xp, ok, disc, disc * 0.0))
        s1 = (-b - sq) / (2.0 * asafe)
        y1 = C[1] + s1 * D[1]
        hit = ok & (s1 > eps) & (abs(y1 - c[1]) <= 0.5 * h)
        best = _where(xp, hit & (s1 < best), s1, best)
        for ycap in (c[1] - 0.5 * h, c[1] + 0.5 * h):
            nzd = abs(D[1]) > 1e-12
            s = (ycap - C[1]) / _where(xp, nzd, D[1], D[1] * 0.0 + 1.0)
            px, pz = C[0] + s * D[0] - c[0], C[2] + s * D[2] - c[2]
            hit = nzd & (s > eps) & (px * px + pz * pz <= r * r)
            best = _where(xp, hit & (s < best), s, best)

    miss = best >= big * 0.5
    depth64 = _where(xp, miss, best * 0.0, best)
    if noise_sigma > 0:
        if is_torch:
            g = xp.Generator(device=device)
            g.manual_seed(seed)
            nz_ = xp.randn(depth64.shape, generator=g, dtype=xp.float64, device=device) * noise_sigma
        else:
            nz_ = np.random.default_rng(seed).standard_normal(depth64.shape) * noise_sigma
        depth64 = _where(xp, miss, depth64, depth64 + nz_)
    if is_torch:
        depth = depth64.to(xp.float32)
    else:
        depth = depth64.astype(np.float32)
    if not want_color:
        return depth, None
    # procedural colour from the world hit point (smooth, so voxel-mean colours are well defined)
    P = [C[i] + depth64 * D[i] for i in range(3)]
    chans = []
    for k, (fq, ph) in enumerate(((9.0, 0.0), (7.0, 1.0), (5.0, 2.0))):
        val = 128.0 + 100.0 * xp.sin(fq * P[k] + ph) + 20.0 * xp.cos(3.0 * P[(k + 1) % 3])
        val = _where(xp, miss, val * 0.0, val)
        chans.append(val)
    if is_torch:
        bgr = xp.stack([chans[2], chans[1], chans[0]], dim=-1).clamp(0, 255).to(xp.uint8)
    else:
        bgr = np.clip(np.stack([chans[2], chans[1], chans[0]], axis=-1), 0, 255).astype(np.uint8)
    return depth, bgr


# ----------------------------------------------------------------------------------------------
# canned configurations
# ----------------------------------------------------------------------------------------------
def object_scene(with_room=True):
    """'buddha stand-in': union of spheres on a turntable, inside a closed room so every ray has depth."""
    sc = Scene(spheres=[((0.0, 0.05, 0.0), 0.22), ((0.0, -0.22, 0.0), 0.14), ((0.16, 0.08, 0.05), 0.10),
                        ((-0.15, 0.12, -0.04), 0.11), ((0.0, -0.38, 0.02), 0.08), ((0.05, 0.25, -0.12), 0.09)])
    if with_room:
        sc.room = ((-1.2, -1.2, -1.2), (1.2, 0.35, 1.2))
    return sc


def corridor_scene():
    """config 3: rectangular corridor 2 m x 2.4 m, closed at the far end."""
    return Scene(room=((-1.0, -1.2, -0.5), (1.0, 1.2, 12.0)),
                 spheres=[((0.6, 0.8, 3.0), 0.3), ((-0.5, 0.9, 6.0), 0.35)])


def cylinder_scene(ground=True):
    """config 4: textureless cylinder r=0.3 h=1 (+ ground plane to break the symmetry)."""
    sc = Scene(cylinders=[((0.0, 0.0, 0.0), 0.3, 1.0)])
    if ground:
        sc.planes = [((0.0, -1.0, 0.0), -0.5)]       # y = 0.5 seen from above (y down)
        sc.spheres = [((0.55, 0.38, 0.1), 0.12), ((-0.2, 0.40, 0.6), 0.10)]
    return sc


def plane_sphere_scene():
    """config 1: tilted plane + sphere at 1-2 m."""
    n = np.array([0.15, 0.1, -1.0])
    n /= np.linalg.norm(n)
    return Scene(planes=[(tuple(n), float(n @ np.array([0.0, 0.0, 1.8])))], spheres=[((0.05, 0.0, 1.3), 0.25)])


HEADLINE = dict(width=1080, height=1920, fx=1719.0, fy=1719.0, cx=540.0, cy=960.0,
                grid=512, voxel=0.005, radius=1.0, deg_per_frame=360.0 / 512)
