#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (run in the build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.ply

The reference is pure Python.  `cv2` is imported unconditionally at
depth_to_reconstruction.py:25 / depth_enhanced_reconstruction.py:15 but the dense
back end never calls it, so an empty stub module named `cv2` is enough to import
both files.  Nothing of the reference is copied: the files
written here hold inputs digests and OUTPUT arrays only.

Symbols exercised (file:line in /root/reference):
  DenseReconstructor._get_projection_factors   depth_to_reconstruction.py:287-295
  DenseReconstructor.depth_to_pointcloud       depth_to_reconstruction.py:328-384
  DenseReconstructor.estimate_scale            depth_to_reconstruction.py:297-326
  DenseReconstructor.merge_pointclouds         depth_to_reconstruction.py:386-420 (no-Open3D branch)
  DepthToReconstructionPipeline.save_reconstruction  depth_to_reconstruction.py:673-703 (ASCII branch)
  DensePointCloudGenerator.depth_to_pointcloud depth_enhanced_reconstruction.py:554-613
  DensePointCloudGenerator.merge_pointclouds   depth_enhanced_reconstruction.py:615-645
  DepthScaleEstimator.estimate_scale           depth_enhanced_reconstruction.py:659-697
  DepthEnhancedReconstruction._save_pointcloud depth_enhanced_reconstruction.py:1283-1311
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs as gi  # noqa: E402

REF = os.environ.get("TL3D_REFERENCE", "/root/reference")


def import_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import depth_to_reconstruction as d2r
        import depth_enhanced_reconstruction as der
    assert not d2r.O3D_AVAILABLE and not der.O3D_AVAILABLE
    return d2r, der


# ---------------------------------------------------------------------------------------
# back-projection cases.  store: 'full' keeps every output point, 'sample' keeps N, a strided
# sample, the first/last 16 points and fp64 column sums (large frames would not fit a fixture).
# ---------------------------------------------------------------------------------------
BP_CASES = [
    # name            seed  h     w     K     api    pose   tshape scale               sub limits       store
    ("s_none_s1",     11,   60,   80,  "K_S", "d2r", None,  None,  ("py", 1.0),        1,  (0.1, 50.0),  "full"),
    ("s_pose_s1",     12,   60,   80,  "K_S", "d2r", 3,     (3, 1), ("py", 1.0),       1,  (0.1, 50.0),  "full"),
    ("s_pose_s2",     13,   60,   80,  "K_S", "d2r", 4,     (3, 1), ("py", 1.25),      2,  (0.1, 50.0),  "full"),
    ("s_pose_s4",     14,   60,   80,  "K_S", "d2r", 5,     (3,),   ("py", 0.75),      4,  (0.1, 50.0),  "full"),
    ("s_np64scale",   15,   60,   80,  "K_S", "d2r", 6,     (3, 1), ("np64", 1.25),    1,  (0.1, 50.0),  "full"),
    ("s_np64scale_b", 15,   60,   80,  "K_S", "d2r", 6,     (3, 1), ("py", 1.25),      1,  (0.1, 50.0),  "full"),
    ("s_odd_s3",      16,   61,   83,  "K_S", "d2r", 7,     (3, 1), ("py", 1.0),       3,  (0.1, 50.0),  "full"),
    ("s_der_s1",      17,   60,   80,  "K_S", "der", 8,     (3, 1), ("py", 1.0),       1,  (0.1, 100.0), "full"),
    ("s_der_s4",      18,   60,   80,  "K_S", "der", 9,     (3, 1), ("premul", 1.5),   4,  (0.1, 100.0), "full"),
    ("s_der_mismatch", 19,  48,   64,  "K_S", "der_mismatch", 9, (3, 1), ("py", 1.0),  2,  (0.1, 100.0), "full"),
    ("A_s2",          0,    480,  640, "K_A", "d2r", "roty", (3, 1), ("py", 1.25),     2,  (0.1, 50.0),  "sample"),
    ("A_s1_none",     1,    480,  640, "K_A", "d2r", None,  None,   ("py", 1.0),       1,  (0.1, 50.0),  "sample"),
    ("A_s4_none",     2,    480,  640, "K_A", "d2r", None,  None,   ("py", 1.0),       4,  (0.1, 50.0),  "sample"),
    ("A_odd_s1",      3,    481,  641, "K_A", "d2r", 10,    (3, 1), ("py", 1.0),       1,  (0.1, 50.0),  "sample"),
    ("A_odd_s2",      3,    481,  641, "K_A", "d2r", 10,    (3, 1), ("py", 1.0),       2,  (0.1, 50.0),  "sample"),
    ("A_odd_s3",      3,    481,  641, "K_A", "d2r", 10,    (3, 1), ("py", 1.0),       3,  (0.1, 50.0),  "sample"),
    ("A_odd_s4",      3,    481,  641, "K_A", "d2r", 10,    (3, 1), ("py", 1.0),       4,  (0.1, 50.0),  "sample"),
    ("B_d2r_s2",      4,    1920, 1080, "K_B", "d2r", 11,   (3, 1), ("py", 1.0),       2,  (0.1, 50.0),  "sample"),
    ("B_d2r_s1",      5,    1920, 1080, "K_B", "d2r", 12,   (3, 1), ("np64", 0.9),     1,  (0.1, 50.0),  "sample"),
    ("B_der_s4",      6,    1920, 1080, "K_B", "der", 13,   (3, 1), ("py", 1.0),       4,  (0.1, 100.0), "sample"),
]


def case_inputs(case):
    """Rebuild (depth, color, pose, scale) for a BP case -- used by tests too."""
    name, seed, h, w, kname, api, pose, tshape, scale, sub, limits, store = case
    if name == "A_s2":
        # depth in [1,2), rotation about y by 0.1 rad, t=(0.1,0.02,-0.3)
        depth, color = gi.frame(seed, h, w, 1.0, 2.0)
        r, t = gi.rot_y(0.1), np.array([[0.1], [0.02], [-0.3]])
    else:
        lo, hi = (0.5, 3.5) if h >= 480 else (0.3, 2.5)
        depth, color = gi.frame(seed, h, w, lo, hi)
        r, t = (None, None) if pose is None else gi.pose(pose)
    if pose is None:
        p = None
    else:
        p = (r, t.reshape(tshape))
    kind, val = scale
    if kind == "np64":
        sc = np.float64(val)
    else:
        sc = float(val)
    return depth, color, p, (kind, sc)


def run_bp(d2r, der, case):
    name, seed, h, w, kname, api, pose, tshape, scale, sub, limits, store = case
    K = getattr(gi, kname)
    depth, color, p, (kind, sc) = case_inputs(case)
    if api == "d2r":
        cfg = d2r.ReconstructionConfig(fx=K["fx"], fy=K["fy"], cx=K["cx"], cy=K["cy"],
                                       min_depth=limits[0], max_depth=limits[1])
        dense = d2r.DenseReconstructor(cfg)
        pts, col = dense.depth_to_pointcloud(depth, color, pose=p, scale=sc, subsample=sub)
    else:
        # "der_mismatch": intrinsics W,H differ from the frame -> factors recomputed (DER:575-578)
        iw, ih = (w, h) if api == "der" else (w + 7, h + 5)
        intr = der.CameraIntrinsics(fx=K["fx"], fy=K["fy"], cx=K["cx"], cy=K["cy"], width=iw, height=ih)
        gen = der.DensePointCloudGenerator(intr)
        d_in = depth * sc if kind == "premul" else depth      # caller pre-multiplies (DER:1135)
        pts, col = gen.depth_to_pointcloud(d_in, color, pose=p, min_depth=limits[0],
                                           max_depth=limits[1], subsample=sub)
    assert pts.dtype == np.float32 and col.dtype == np.uint8
    out = {"n": np.int64(len(pts)), "in_digest": gi.digest(depth, color)}
    if store == "full":
        out["points"] = pts
        out["colors"] = col
    else:
        stride = 997
        out["stride"] = np.int64(stride)
        out["points_strided"] = pts[::stride].copy()
        out["colors_strided"] = col[::stride].copy()
        out["points_head"] = pts[:16].copy()
        out["points_tail"] = pts[-16:].copy()
        out["colors_head"] = col[:16].copy()
        out["colors_tail"] = col[-16:].copy()
        out["points_sum"] = pts.astype(np.float64).sum(axis=0)
        out["colors_sum"] = col.astype(np.int64).sum(axis=0)
    return {f"{name}/{k}": v for k, v in out.items()}


def run_thresholds(d2r, der):
    """1x8 threshold row of SURVEY.md section 8a: strict compares in the dtype of depth*scale."""
    row = np.array([[0.1, 0.10000001, 50, 49.999996, np.inf, -1, np.nan, 1]], dtype=np.float32)
    col = np.arange(24, dtype=np.uint8).reshape(1, 8, 3)
    out = {}
    cfg = d2r.ReconstructionConfig(fx=10.0, fy=10.0, cx=4.0, cy=0.0)
    pts, c = d2r.DenseReconstructor(cfg).depth_to_pointcloud(row, col, pose=None, scale=1.0, subsample=1)
    out["thr/d2r_points"], out["thr/d2r_colors"] = pts, c
    pts, c = d2r.DenseReconstructor(cfg).depth_to_pointcloud(row, col, pose=None, scale=np.float64(1.0), subsample=1)
    out["thr/d2r64_points"], out["thr/d2r64_colors"] = pts, c
    row2 = np.array([[100, 99.99999, 0.1, 0.10000001, 75, 0, 2, 3]], dtype=np.float32)
    intr = der.CameraIntrinsics(fx=10.0, fy=10.0, cx=4.0, cy=0.0, width=8, height=1)
    pts, c = der.DensePointCloudGenerator(intr).depth_to_pointcloud(row2, col)
    out["thr/der_points"], out["thr/der_colors"] = pts, c
    out["thr/row"], out["thr/row2"], out["thr/col"] = row, row2, col
    return out


def run_factors(d2r):
    cfg = d2r.ReconstructionConfig(**gi.K_S)
    xf, yf = d2r.DenseReconstructor(cfg)._get_projection_factors(6, 9)
    return {"factors/xf": xf, "factors/yf": yf}


def scale_cases():
    rng = np.random.default_rng(77)
    depth = (0.5 + rng.random((40, 50))).astype(np.float32)
    depth[3, 4] = 0.0
    cases = {}
    # regular: 30 points, true scale 2.5 with noise, fractional pixel coords (int() truncation)
    pts2d = rng.uniform([0, 0], [49.99, 39.99], size=(30, 2))
    z = np.array([depth[int(p[1]), int(p[0])] for p in pts2d], dtype=np.float64) * 2.5
    z *= 1.0 + 0.05 * rng.standard_normal(30)
    pts3d = np.stack([rng.standard_normal(30), rng.standard_normal(30), z], axis=1)
    cases["regular"] = (pts3d, pts2d, depth)
    # the survey's truncation example + out-of-bounds + zero-depth pixel + negative Z
    pts2d_b = np.array([[10.9, 20.2], [4.7, 3.2], [-0.5, 3.0], [50.0, 10.0], [12.0, 39.9], [7.5, 7.5], [8.1, 9.9]])
    pts3d_b = np.array([[0, 0, 2.0], [0, 0, 1.0], [0, 0, 1.0], [0, 0, 1.0], [0, 0, 3.0], [0, 0, -1.0], [0, 0, 1.7]])
    cases["edges"] = (pts3d_b, pts2d_b, depth)
    # < 3 usable samples -> 1.0
    cases["too_few"] = (pts3d_b[:2], pts2d_b[:2], depth)
    # 4 points: D2R uses them, DER returns 1.0 (< 5 input points, DER:673)
    cases["four"] = (pts3d[:4], pts2d[:4], depth)
    # sanity clamp (D2R only, D2R:315): scales 1e4 and 1e-4 rejected
    pts3d_c = pts3d[:8].copy()
    pts3d_c[0, 2] = 1e4
    pts3d_c[1, 2] = 1e-4
    cases["clamp"] = (pts3d_c, pts2d[:8], depth)
    # even count -> median averages the two central values
    cases["even"] = (pts3d[:6], pts2d[:6], depth)
    return cases


def run_scale(d2r, der):
    out = {}
    dense = d2r.DenseReconstructor(d2r.ReconstructionConfig())
    for name, (p3, p2, dm) in scale_cases().items():
        with contextlib.redirect_stdout(io.StringIO()):
            s1 = dense.estimate_scale(p3, p2, dm)
            s2 = der.DepthScaleEstimator.estimate_scale(p3, p2, dm, np.eye(3))
        out[f"scale/{name}_d2r"] = np.float64(s1)
        out[f"scale/{name}_der"] = np.float64(s2)
    return out


def run_merge_and_ply(d2r, der):
    out = {}
    dense = d2r.DenseReconstructor(d2r.ReconstructionConfig())
    e = dense.merge_pointclouds([(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8))])
    out["merge/empty_shape0"] = np.array(e[0].shape)
    out["merge/empty_dtype"] = np.array(str(e[0].dtype))
    p1 = np.array([[0.1, 0.2, 0.3], [1, 2, 3]], np.float32)
    c1 = np.array([[1, 2, 3], [4, 5, 6]], np.uint8)
    p2 = np.array([[-1.5, 0.25, 7.125]], np.float32)
    c2 = np.array([[255, 0, 128]], np.uint8)
    mp, mc = dense.merge_pointclouds([(p1, c1), (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)), (p2, c2)])
    out["merge/points"], out["merge/colors"] = mp, mc
    intr = der.CameraIntrinsics(1.0, 1.0, 0.0, 0.0, 4, 4)
    mp2, mc2 = der.DensePointCloudGenerator(intr).merge_pointclouds([(p1, c1), (p2, c2)])
    assert np.array_equal(mp, mp2) and np.array_equal(mc, mc2)

    # ASCII PLY writers; save_reconstruction/_save_pointcloud never touch self
    ply_pts = np.array([[0.1, -2.5, 3.0], [1e-7, 123456.789, -0.333333343], [1.5, 2.25, 1e10]], np.float32)
    ply_col = np.array([[255, 0, 17], [1, 2, 3], [128, 64, 32]], np.uint8)
    out["ply/points"], out["ply/colors"] = ply_pts, ply_col
    for tag, fn in (("d2r", lambda path: d2r.DepthToReconstructionPipeline.save_reconstruction(None, ply_pts, ply_col, path)),
                    ("der", lambda path: der.DepthEnhancedReconstruction._save_pointcloud(None, ply_pts, ply_col, path))):
        path = os.path.join(HERE, f"ascii_{tag}.ply")
        with contextlib.redirect_stdout(io.StringIO()) as s:
            fn(path)
        out[f"ply/{tag}_stdout"] = np.array(s.getvalue().replace(HERE, "<DIR>"))
        with open(path) as f:
            out[f"ply/{tag}_text"] = np.array(f.read())
    with contextlib.redirect_stdout(io.StringIO()) as s:
        d2r.DepthToReconstructionPipeline.save_reconstruction(None, np.array([]), np.array([]), os.path.join(HERE, "never.ply"))
    out["ply/empty_stdout"] = np.array(s.getvalue())
    assert not os.path.exists(os.path.join(HERE, "never.ply"))
    return out


def main():
    d2r, der = import_reference()
    bp = {}
    for case in BP_CASES:
        bp.update(run_bp(d2r, der, case))
        print("bp", case[0], int(bp[f"{case[0]}/n"]))
    np.savez_compressed(os.path.join(HERE, "backproject.npz"), **bp)
    misc = {}
    misc.update(run_thresholds(d2r, der))
    misc.update(run_factors(d2r))
    misc.update(run_scale(d2r, der))
    misc.update(run_merge_and_ply(d2r, der))
    misc["defaults/d2r"] = np.array([getattr(d2r.ReconstructionConfig(), k) for k in
                                     ("fx", "fy", "cx", "cy", "min_depth", "max_depth", "voxel_size", "subsample_factor")],
                                    dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **misc)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
