"""A SECOND, independently structured formulation of the rows the reference cannot pin -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

The reference holds no TSDF and no ICP: oracle/tl3d_oracle.c DEFINES them for this repo and the HIP
kernels are checked against it bit for bit.  Kernel and C oracle were written by one hand to "the same f32 sequence", so a
misreading shared by both would pass every such test.  This file restates the same contracts a second time
in a different shape -- whole-array numpy expressions over every voxel / every sample instead of scalar loops, its own emulation
of the fused multiply-add, its own record-order arithmetic -- WITHOUT reference to tl3d_oracle.c, and
tests/test_oracle_second_formulation.py compares the two on the CPU: TSDF grids bit for bit, normal maps bit for bit, the ICP
normal equations (27 + 3 sums, and the 8 sums of the Sim(3) column) to 1e-12 relative.  It cannot pin parity at the reference
(nothing reference-held exists for these rows); it guards against a misconception shared by kernel and oracle.

Contracts restated here:
  TSDF   voxel centre w = fma(i + 0.5, voxel, origin) per axis (f32); camera point by nested fma
         x = fma(r0, wx, fma(r1, wy, fma(r2, wz, t0))) ...; inv = 1 / z (IEEE f32); u = fma(fx x, inv, cx); in image iff
         -0.5 <= u < W - 0.5 (same for v) and z > 0; pixel = floor(u + 0.5); d = depth[pixel] * scale; valid iff
         min_depth < d < max_depth (f32 compares); sdf = d - z; update iff sdf >= -trunc;
         tsdf = min(1, sdf * (1 / trunc)); q = rint(tsdf * 32767); record += (q, 1).
         Record order: brick-major (bricks x fastest), inside a brick eight 4x4x4 sub-bricks (x>>2 | (y>>2)<<1 | (z>>2)<<2) of
         64 records, x fastest inside a sub-brick.
  normals  vertex p(u, v) = (((u - cx) / fx) d, ((v - cy) / fy) d, d) (f32); tangents from the pixels `step` to either side, all
         five valid and within depth_jump of the centre; n = a x b with one fma per component, normalised by 1 / sqrt(len2),
         flipped towards the camera; nmap = (n, d).
  smoothed depth  harmonic mean over the centre and the symmetric pixel PAIRS of the (2 r + 1)^2 window that are both valid and
         within depth_jump of the centre, f32 sum in row-major order of the pair's first pixel.
  ICP    source sample (every stride-th pixel) -> p by the vertex rule with scale_src; q = R p + t by nested fma (f32 pose);
         projective association: pixel = floor(fma(fx qx, 1 / qz, cx) + 0.5) (same window rule as the TSDF); target depth and
         normal from nmap; target vertex by the vertex rule at that pixel; gate dist2 = fma(dx, dx, fma(dy, dy, dz dz)) <= max_dist^2;
         residual = fma(dx, nx, fma(dy, ny, dz nz)); J = (q x n, n) with one fma per cross component; fp64 sums of fp64 products;
         Sim(3) column J_alpha = fma(nx, qx - tx, fma(ny, qy - ty, nz (qz - tz))).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64


def fma32(a, b, c):
    """round_f32(a * b + c) with ONE rounding, for float32 arrays (or scalars), via float64.

    a * b is exact in float64 (24 + 24 bits).  (a b) + c in float64 rounds once to 53 bits and the cast to float32 a second time:
    the two-step result differs from the fused one only when the float64 sum lands exactly on the midpoint of two neighbouring
    float32 values while the exact sum does not (the float64 rounding is monotone and midpoints are float64 numbers, so it cannot
    step over one).  The error of the float64 addition (two-sum) says on which side the exact value lies."""
    a64, b64, c64 = np.asarray(a, F32).astype(F64), np.asarray(b, F32).astype(F64), np.asarray(c, F32).astype(F64)
    p = a64 * b64
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)                   # exact error of the float64 addition
    r = s.astype(F32)
    r64 = r.astype(F64)
    with np.errstate(over="ignore", invalid="ignore"):
        lo = np.where(r64 <= s, r, np.nextafter(r, F32(-np.inf)))
        hi = np.where(r64 >= s, r, np.nextafter(r, F32(np.inf)))
        mid = 0.5 * (lo.astype(F64) + hi.astype(F64))
        tie = (lo != hi) & (mid == s) & (err != 0.0) & np.isfinite(s)
    return np.where(tie, np.where(err > 0.0, hi, lo), r).astype(F32)


def record_index(i, j, k, nbx, nby):
    """record of voxel (i, j, k): brick (i>>3, j>>3, k>>3), bricks x fastest; sub-brick, then x fastest inside it"""
    i, j, k = np.asarray(i, np.int64), np.asarray(j, np.int64), np.asarray(k, np.int64)
    brick = ((k >> 3) * nby + (j >> 3)) * nbx + (i >> 3)
    sub = ((i >> 2) & 1) | (((j >> 2) & 1) << 1) | (((k >> 2) & 1) << 2)
    inner = (i & 3) | ((j & 3) << 2) | ((k & 3) << 4)
    return brick * 512 + sub * 64 + inner


class Geometry:
    """intrinsics / limits / grid as the float32 values the arithmetic uses (the configuration holds doubles)"""

    def __init__(self, width, height, fx, fy, cx, cy, min_depth=0.1, max_depth=50.0, dims=(0, 0, 0), origin=(0.0, 0.0, 0.0),
                 voxel_size=0.005, sdf_trunc=0.02):
        self.W, self.H = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy = F32(fx), F32(fy), F32(cx), F32(cy)
        self.mind, self.maxd = F32(min_depth), F32(max_depth)
        self.dims = tuple(int(d) for d in dims)
        self.origin = tuple(F32(o) for o in origin)
        self.voxel = F32(voxel_size)
        self.trunc = F32(sdf_trunc)
        self.inv_trunc = F32(1.0) / self.trunc if self.trunc > 0 else F32(0)


def _pose32(R, t):
    return np.asarray(R, F64).reshape(9).astype(F32), np.asarray(t, F64).reshape(3).astype(F32)


def _transform(r, t, x, y, z):
    """(r x + t) by the nested-fma chain of the contract: fma(r0, x, fma(r1, y, fma(r2, z, t0)))"""
    out = []
    for a in range(3):
        out.append(fma32(r[3 * a], x, fma32(r[3 * a + 1], y, fma32(r[3 * a + 2], z, t[a]))))
    return out


def _project(g: Geometry, x, y, z):
    """(in-window flag, u, v) of camera points; pixel = floor(uf + 0.5)"""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = F32(1.0) / z
        uf = fma32(g.fx * x, inv, g.cx)
        vf = fma32(g.fy * y, inv, g.cy)
        ok = (z > F32(0)) & (uf >= F32(-0.5)) & (uf < F32(g.W) - F32(0.5)) & (vf >= F32(-0.5)) & (vf < F32(g.H) - F32(0.5))
        u = np.floor(uf + F32(0.5))
        v = np.floor(vf + F32(0.5))
    u = np.where(ok, u, 0).astype(np.int64)
    v = np.where(ok, v, 0).astype(np.int64)
    return ok, np.minimum(u, g.W - 1), np.minimum(v, g.H - 1)


def tsdf_integrate(g: Geometry, grid, depth, R, t, scale=1.0):
    """grid: int32 [nvox, 2] in record order, updated in place; depth float32 [H, W]"""
    nx, ny, nz = g.dims
    r, tt = _pose32(R, t)
    depth = np.ascontiguousarray(depth, F32)
    sc = F32(scale)
    half = F32(0.5)
    wx = fma32(np.arange(nx, dtype=F32) + half, g.voxel, g.origin[0])
    wy = fma32(np.arange(ny, dtype=F32) + half, g.voxel, g.origin[1])
    wz = fma32(np.arange(nz, dtype=F32) + half, g.voxel, g.origin[2])
    K, J, I = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    X, Y, Z = wx[I], wy[J], wz[K]
    xc, yc, zc = _transform(r, tt, X, Y, Z)
    ok, u, v = _project(g, xc, yc, zc)
    d = depth[v, u] * sc
    ok &= (d > g.mind) & (d < g.maxd)
    with np.errstate(invalid="ignore", over="ignore"):
        sdf = d - zc
        ok &= sdf >= -g.trunc
        tsdf = np.minimum(F32(1.0), sdf * g.inv_trunc)
        q = np.rint(tsdf * F32(32767.0))
    rec = record_index(I[ok], J[ok], K[ok], nx // 8, ny // 8)
    grid[rec, 0] += q[ok].astype(np.int32)
    grid[rec, 1] += 1
    return int(ok.sum())


def _vertex_maps(g: Geometry, depth, scale):
    """(valid, px, py, d) of every pixel: p = (((u - cx) / fx) d, ((v - cy) / fy) d, d)"""
    d = np.ascontiguousarray(depth, F32) * F32(scale)
    with np.errstate(invalid="ignore"):
        valid = (d > g.mind) & (d < g.maxd)
    xf = (np.arange(g.W, dtype=F32) - g.cx) / g.fx
    yf = (np.arange(g.H, dtype=F32) - g.cy) / g.fy
    with np.errstate(invalid="ignore", over="ignore"):
        return valid, xf[None, :] * d, yf[:, None] * d, d


def normals(g: Geometry, depth, scale=1.0, depth_jump=0.05, step=1):
    """nmap float32 [H, W, 4] = (n, depth) or zeros"""
    H, W = g.H, g.W
    valid, px, py, d = _vertex_maps(g, depth, scale)
    jump = F32(depth_jump)
    out = np.zeros((H, W, 4), F32)
    if H <= 2 * step or W <= 2 * step:
        return out
    c = (slice(step, H - step), slice(step, W - step))
    le = (slice(step, H - step), slice(0, W - 2 * step))
    ri = (slice(step, H - step), slice(2 * step, W))
    up = (slice(0, H - 2 * step), slice(step, W - step))
    dn = (slice(2 * step, H), slice(step, W - step))
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        ok = valid[c] & valid[le] & valid[ri] & valid[up] & valid[dn]
        for s in (le, ri, up, dn):
            ok &= np.abs(d[s] - d[c]) <= jump
        ax, ay, az = px[ri] - px[le], py[ri] - py[le], d[ri] - d[le]
        bx, by, bz = px[dn] - px[up], py[dn] - py[up], d[dn] - d[up]
        nx = fma32(ay, bz, -(az * by))
        ny = fma32(az, bx, -(ax * bz))
        nz = fma32(ax, by, -(ay * bx))
        len2 = fma32(nx, nx, fma32(ny, ny, nz * nz))
        ok &= len2 > F32(1e-30)
        inv = F32(1.0) / np.sqrt(len2)
        nx, ny, nz = nx * inv, ny * inv, nz * inv
        dot = fma32(nx, px[c], fma32(ny, py[c], nz * d[c]))
        flip = dot > F32(0)
    nx, ny, nz = np.where(flip, -nx, nx), np.where(flip, -ny, ny), np.where(flip, -nz, nz)
    res = np.stack([nx, ny, nz, d[c]], axis=-1)
    out[c] = np.where(ok[..., None], res, F32(0))
    return out


def smooth_depth(g: Geometry, depth, scale=1.0, depth_jump=0.05, radius=1):
    """harmonic window mean, in the units of `depth`; 0 where the centre pixel is invalid"""
    H, W = g.H, g.W
    raw = np.ascontiguousarray(depth, F32)
    sc, jump = F32(scale), F32(depth_jump)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        d = raw * sc
        valid = (d > g.mind) & (d < g.maxd)
        inv_raw = F32(1.0) / raw
        total = np.where(valid, inv_raw, F32(0)).astype(F32)
        count = np.where(valid, 1, 0).astype(np.int64)
        for dv in range(0, radius + 1):
            for du in range(1 if dv == 0 else -radius, radius + 1):
                # pixel pairs (u + du, v + dv), (u - du, v - dv) of every centre (u, v) for which both lie in the image
                v0, v1 = max(0, dv, -dv), min(H, H - dv, H + dv)
                u0, u1 = max(0, du, -du), min(W, W - du, W + du)
                if v0 >= v1 or u0 >= u1:
                    continue
                cs = (slice(v0, v1), slice(u0, u1))
                sa = (slice(v0 + dv, v1 + dv), slice(u0 + du, u1 + du))
                sb = (slice(v0 - dv, v1 - dv), slice(u0 - du, u1 - du))
                ok = valid[cs] & valid[sa] & valid[sb] & (np.abs(d[sa] - d[cs]) <= jump) & (np.abs(d[sb] - d[cs]) <= jump)
                t = total[cs]
                t = np.where(ok, (t + inv_raw[sa]).astype(F32), t)           # two separate f32 additions, in this order
                t = np.where(ok, (t + inv_raw[sb]).astype(F32), t)
                total[cs] = t
                count[cs] += 2 * ok
        out = np.where(valid, count.astype(F32) / total, F32(0)).astype(F32)
    return out


def icp_sums(g: Geometry, depth_src, nmap_tgt, T, stride=4, max_dist=0.05, scale_src=1.0, with_scale_column=False):
    """The normal equations of ONE point-to-plane pass at pose T (4x4, source camera -> target camera).

    Returns (A [6,6] = sum J J^T, b [6] = sum J r, sum r^2, correspondences, valid source samples) and, with the scale column,
    additionally (sum J_a J_alpha [6], sum J_alpha^2, sum J_alpha r)."""
    T = np.asarray(T, F64).reshape(4, 4)
    r = T[:3, :3].reshape(9).astype(F32)
    t = T[:3, 3].astype(F32)
    us = np.arange(0, g.W, stride)
    vs = np.arange(0, g.H, stride)
    draw = np.ascontiguousarray(depth_src, F32)[np.ix_(vs, us)]
    d = draw * F32(scale_src)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        src_ok = (d > g.mind) & (d < g.maxd)
        p0 = ((us.astype(F32) - g.cx) / g.fx)[None, :] * d
        p1 = ((vs.astype(F32) - g.cy) / g.fy)[:, None] * d
        qx, qy, qz = _transform(r, t, p0, p1, d)
        ok, ut, vt = _project(g, qx, qy, qz)
        ok &= src_ok
        nm = np.ascontiguousarray(nmap_tgt, F32)[vt, ut]
        dt = nm[..., 3]
        ok &= dt > F32(0)
        tx = ((ut.astype(F32) - g.cx) / g.fx) * dt
        ty = ((vt.astype(F32) - g.cy) / g.fy) * dt
        dx, dy, dz = qx - tx, qy - ty, qz - dt
        dist2 = fma32(dx, dx, fma32(dy, dy, dz * dz))
        ok &= dist2 <= F32(max_dist * max_dist)
        nx, ny, nz = nm[..., 0], nm[..., 1], nm[..., 2]
        res = fma32(dx, nx, fma32(dy, ny, dz * nz))
        J = [fma32(qy, nz, -(qz * ny)), fma32(qz, nx, -(qx * nz)), fma32(qx, ny, -(qy * nx)), nx, ny, nz]
        ja = fma32(nx, qx - t[0], fma32(ny, qy - t[1], nz * (qz - t[2])))
    Jm = np.stack([j[ok].astype(F64) for j in J], axis=0)                  # [6, n]
    rr = res[ok].astype(F64)
    A = Jm @ Jm.T
    b = Jm @ rr
    out = (A, b, float(rr @ rr), int(ok.sum()), int(src_ok.sum()))
    if with_scale_column:
        jam = ja[ok].astype(F64)
        out = out + (Jm @ jam, float(jam @ jam), float(jam @ rr))
    return out
