"""GPU parity for rows a3-a5: the HIP back-projection vs vectors captured from the reference itself."""
import numpy as np
import pytest

import inputs as gi
import make_golden as mg
import tl3d
from helpers import ulp_diff
from oracle import ref_numpy as rn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden_dir):
    import os
    return np.load(os.path.join(golden_dir, "backproject.npz"))


def _run_case(case):
    name, seed, h, w, kname, api, pose, tshape, scale, sub, limits, store = case
    K = getattr(gi, kname)
    depth, color, p, (kind, sc) = mg.case_inputs(case)
    if kind == "premul":
        depth_in, s, f64 = depth * sc, 1.0, False          # DER: caller pre-multiplies in f32 (DER:1135)
    else:
        depth_in, s, f64 = depth, float(sc), kind == "np64"
    with tl3d.FusionContext(w, h, K["fx"], K["fy"], K["cx"], K["cy"], min_depth=limits[0], max_depth=limits[1],
                            n_slots=1) as ctx:
        ctx.upload(0, depth_in, color)
        pts, col = ctx.backproject(0, pose=p, scale=s, subsample=sub, scale_f64=f64)
        pts, col = pts.copy(), col.copy()
    return depth, color, pts, col


@pytest.mark.parametrize("case", mg.BP_CASES, ids=[c[0] for c in mg.BP_CASES])
def test_backproject_matches_reference_golden(case, gold):
    name, store = case[0], case[-1]
    depth, color, pts, col = _run_case(case)
    assert gi.digest(depth, color) == str(gold[f"{name}/in_digest"])
    n = int(gold[f"{name}/n"])
    assert len(pts) == n                                    # identical validity mask
    if store == "full":
        gp, gc = gold[f"{name}/points"], gold[f"{name}/colors"]
        assert np.array_equal(col, gc)                      # BGR->RGB, bit exact
        ud = ulp_diff(pts, gp)
        # fp64 intermediates on both sides; only the association of the 3-term sums may differ
        assert ud.max() <= 1 and (ud > 0).mean() < 1e-3, (ud.max(), (ud > 0).mean())
    else:
        st = int(gold[f"{name}/stride"])
        assert np.array_equal(col[::st], gold[f"{name}/colors_strided"])
        assert np.array_equal(col[:16], gold[f"{name}/colors_head"]) and np.array_equal(col[-16:], gold[f"{name}/colors_tail"])
        assert np.array_equal(col.astype(np.int64).sum(0), gold[f"{name}/colors_sum"])
        for a, b in ((pts[::st], gold[f"{name}/points_strided"]), (pts[:16], gold[f"{name}/points_head"]),
                     (pts[-16:], gold[f"{name}/points_tail"])):
            assert ulp_diff(a, b).max() <= 1
        assert np.allclose(pts.astype(np.float64).sum(0), gold[f"{name}/points_sum"], rtol=0, atol=1e-4)


def test_threshold_row_and_u16(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    row, col = g["thr/row"], g["thr/col"]
    with tl3d.FusionContext(8, 1, 10.0, 10.0, 4.0, 0.0, 0.1, 50.0, n_slots=1) as ctx:
        ctx.upload(0, row, col)
        p, c = ctx.backproject(0)
        assert np.array_equal(p, g["thr/d2r_points"]) and np.array_equal(c, g["thr/d2r_colors"])
        p, c = ctx.backproject(0, scale=1.0, scale_f64=True)
        assert np.array_equal(p, g["thr/d2r64_points"]) and np.array_equal(c, g["thr/d2r64_colors"])
        ctx.upload(0, g["thr/row2"], col)
        p, c = ctx.backproject(0, min_depth=0.1, max_depth=100.0)
        assert np.array_equal(p, g["thr/der_points"]) and np.array_equal(c, g["thr/der_colors"])
        # 16-bit millimetre PNG payload: f32(u16)/1000 as depth_to_reconstruction.py:90
        mm = np.array([[0, 99, 100, 101, 1500, 49999, 50000, 65535]], np.uint16)
        ctx.upload(0, mm, col)
        assert np.array_equal(ctx.download_depth(0), mm.astype(np.float32) / 1000.0)
        p, c = ctx.backproject(0)
        rp, rc = rn.backproject(mm.astype(np.float32) / 1000.0, col, 10.0, 10.0, 4.0, 0.0)
        assert np.array_equal(p, rp) and np.array_equal(c, rc)


def test_empty_and_capacity():
    with tl3d.FusionContext(16, 8, 10.0, 10.0, 8.0, 4.0, n_slots=1) as ctx:
        ctx.upload(0, np.zeros((8, 16), np.float32), None)
        p, c = ctx.backproject(0)
        assert p.shape == (0, 3) and c.shape == (0, 3)
        with pytest.raises(tl3d.Tl3dError):
            ctx.backproject(1)                          # bad slot
        with pytest.raises(tl3d.Tl3dError):
            ctx.integrate(0, (np.eye(3), np.zeros(3)))  # no grid channel


def test_randomised_shapes_strides_and_limits_match_pinned_oracle():
    """Ragged sizes (1xN, Nx1, odd, prime), every stride 1..5, D2R/DER limits, pose / no pose, f32 / fp64 scale: the HIP
    path against oracle/ref_numpy.py, which is itself bit-checked against the reference's own outputs."""
    rng = np.random.default_rng(2024)
    shapes = [(1, 1), (1, 37), (41, 1), (2, 3), (17, 19), (61, 83), (97, 64), (128, 257)]
    for case in range(40):
        h, w = shapes[case % len(shapes)]
        depth = (0.05 + 3.0 * rng.random((h, w))).astype(np.float32)
        bad = rng.random((h, w))
        depth[bad < 0.05] = 0.0
        depth[(bad > 0.05) & (bad < 0.08)] = np.nan
        depth[(bad > 0.08) & (bad < 0.10)] = np.inf
        depth[(bad > 0.10) & (bad < 0.12)] = 120.0
        color = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        K = dict(fx=float(rng.uniform(20, 900)), fy=float(rng.uniform(20, 900)), cx=float(rng.uniform(0, w)), cy=float(rng.uniform(0, h)))
        sub = int(rng.integers(1, 6))
        lim = (0.1, 50.0) if case % 2 else (0.1, 100.0)
        pose = None if case % 5 == 0 else gi.pose(100 + case)
        scale = [1.0, 0.37, np.float64(1.7), np.float64(1.0)][case % 4]
        with tl3d.FusionContext(w, h, K["fx"], K["fy"], K["cx"], K["cy"], lim[0], lim[1], n_slots=1) as ctx:
            ctx.upload(0, depth, color)
            p, c = ctx.backproject(0, pose=pose, scale=float(scale), subsample=sub, scale_f64=isinstance(scale, np.float64))
            p, c = p.copy(), c.copy()
        rp, rc = rn.backproject(depth, color, K["fx"], K["fy"], K["cx"], K["cy"], pose=pose, scale=scale, subsample=sub,
                                min_depth=lim[0], max_depth=lim[1])
        assert len(p) == len(rp), (case, h, w, sub)
        assert np.array_equal(c, rc)
        if len(p):
            assert ulp_diff(p, rp).max() <= 1, (case, ulp_diff(p, rp).max())


def test_full_hd_all_valid_stride_1_repeated_keeps_order_and_values():
    """1080 x 1920, every pixel valid, stride 1: 1013 tiles of 2048 samples, more than the chip holds at once (dynamic tile
    order, look-back over many windows, every staging pass full).  The case in which a development build of round 2 copied a
    wave's part of the LDS staging slab out before it was written; repeated, because that depended on timing."""
    rng = np.random.default_rng(77)
    h, w = 1920, 1080
    depth = (0.5 + 2.0 * rng.random((h, w))).astype(np.float32)
    color = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    pose = gi.pose(5)
    rp, rc = rn.backproject(depth, color, 1719.0, 1719.0, 540.0, 960.0, pose=pose, scale=1.0, subsample=1, min_depth=0.1, max_depth=50.0)
    assert len(rp) == h * w
    with tl3d.FusionContext(w, h, 1719.0, 1719.0, 540.0, 960.0, 0.1, 50.0, n_slots=1) as ctx:
        ctx.upload(0, depth, color)
        for rep in range(6):
            p, c = ctx.backproject(0, pose=pose, scale=1.0, subsample=1)
            assert len(p) == h * w, rep
            assert np.array_equal(c, rc), rep
            assert ulp_diff(p, rp).max() <= 1, rep
        assert ctx.stats()["bp_lookback_retries"] == 0


def test_u16_millimetre_conversion_is_exact_for_every_value():
    """uint16 mm -> float32 m on device == numpy's `.astype(float32) / 1000.0` (D2R:90) for all 65 536 inputs."""
    import tl3d
    mm = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    with tl3d.FusionContext(256, 256, 200.0, 200.0, 128.0, 128.0, n_slots=1) as ctx:
        ctx.upload(0, mm, None)
        got = ctx.download_depth(0)
    assert np.array_equal(got, mm.astype(np.float32) / np.float32(1000.0))


def test_device_only_backprojection_offsets_alignment_and_capacity():
    """tl3d_backproject_device: one kernel, points / colours / count stay on the device.  Output ranges start at arbitrary
    point offsets inside a tile and the caller's base pointers need not be 16-byte aligned (the kernel stages through LDS and
    writes 16-byte chunks where the destination allows): every combination must equal the blocking host-output call, which
    itself is checked against the reference goldens above.  cap smaller than the count: nothing beyond cap is written."""
    import torch
    rng = np.random.default_rng(11)
    h, w = 203, 517                                    # several tiles of 2048 samples, ragged last tile
    depth = (0.5 + 3.0 * rng.random((h, w))).astype(np.float32)
    depth[rng.random((h, w)) < 0.37] = 0.0             # holes: tiles start at arbitrary output offsets
    depth[:, 100:180] = 0.0                            # and whole runs without survivors
    color = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q *= np.sign(np.linalg.det(q))
    pose = (q, rng.normal(size=3))
    dev = torch.device("cuda", 0)
    with tl3d.FusionContext(w, h, 300.0, 310.0, 250.5, 99.25, n_slots=2) as ctx:
        ctx.upload(0, depth, color)
        ctx.upload(1, np.zeros((h, w), np.float32), color)
        for sub in (1, 2, 3):
            ref_p, ref_c = ctx.backproject(0, pose=pose, scale=1.1, subsample=sub)
            ref_p, ref_c = ref_p.copy(), ref_c.copy()
            n = len(ref_p)
            assert n > 1000
            op, oc = rn.backproject(depth, color, 300.0, 310.0, 250.5, 99.25, pose=pose, scale=1.1, subsample=sub)
            assert len(op) == n and np.array_equal(oc, ref_c) and ulp_diff(ref_p, op).max() <= 1
            cap = -(-h // sub) * -(-w // sub)
            for off_f, off_b in ((0, 0), (1, 1), (2, 7), (3, 13)):          # base pointers at odd dword / byte phases
                bx = torch.full((3 * cap + 8,), -7.0, dtype=torch.float32, device=dev)
                bc = torch.full((3 * cap + 32,), 201, dtype=torch.uint8, device=dev)
                nn = torch.zeros(1, dtype=torch.int64, device=dev)
                xv, cv = bx[off_f:off_f + 3 * cap].view(cap, 3), bc[off_b:off_b + 3 * cap].view(cap, 3)
                ctx.backproject_device(0, xv, cv, nn, pose=pose, scale=1.1, subsample=sub)
                ctx.sync()
                assert int(nn.item()) == n
                assert np.array_equal(xv[:n].cpu().numpy().view(np.int32), ref_p.view(np.int32)) and np.array_equal(cv[:n].cpu().numpy(), ref_c)
                # nothing outside [0, n) was touched, guard words included
                assert (bx[:off_f] == -7.0).all() and (bx[off_f + 3 * n:] == -7.0).all()
                assert (bc[:off_b] == 201).all() and (bc[off_b + 3 * n:] == 201).all()
            # capacity smaller than the count: true count reported, writes stop at cap
            small = n // 2 + 1
            bx = torch.full((3 * small + 64,), -7.0, dtype=torch.float32, device=dev)
            bc = torch.full((3 * small + 64,), 201, dtype=torch.uint8, device=dev)
            nn = torch.zeros(1, dtype=torch.int64, device=dev)
            ctx.backproject_device(0, bx[:3 * small].view(small, 3), bc[:3 * small].view(small, 3), nn, pose=pose, scale=1.1, subsample=sub, cap=small)
            ctx.sync()
            assert int(nn.item()) == n
            assert np.array_equal(bx[:3 * small].cpu().numpy().view(np.int32), ref_p[:small].reshape(-1).view(np.int32))
            assert (bx[3 * small:] == -7.0).all() and (bc[3 * small:] == 201).all()
            with pytest.raises(tl3d.Tl3dError) as e:                      # the blocking call reports it as an error
                x_small, c_small = np.empty((small, 3), np.float32), np.empty((small, 3), np.uint8)
                from tl3d import _cabi as abi
                import ctypes as C
                got = C.c_int64(0)
                r9, t3 = abi.d9(pose[0]), abi.d3(pose[1])
                abi.check(ctx._lib.tl3d_backproject(ctx._h, 0, abi.ptr(r9), abi.ptr(t3), 1.1, 0, sub, 0.1, 50.0, abi.ptr(x_small), abi.ptr(c_small),
                                                    small, C.byref(got)))
            assert e.value.code == abi.E_CAPACITY and got.value == n
        # a frame without survivors: count 0, nothing written
        bx = torch.full((64,), -7.0, dtype=torch.float32, device=dev)
        bc = torch.full((64,), 201, dtype=torch.uint8, device=dev)
        nn = torch.full((1,), 5, dtype=torch.int64, device=dev)
        ctx.backproject_device(1, bx[:63].view(21, 3), bc[:63].view(21, 3), nn, pose=pose)
        ctx.sync()
        assert int(nn.item()) == 0 and (bx == -7.0).all() and (bc == 201).all()


def test_frame_bounds_equal_the_extent_of_the_point_list():
    """tl3d_frame_bounds (the scene-bounding pass of the pipeline): min / max of the float32 points, without the points."""
    rng = np.random.default_rng(5)
    h, w = 97, 131
    depth = (0.4 + 2.0 * rng.random((h, w))).astype(np.float32)
    depth[rng.random((h, w)) < 0.3] = 0.0
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q *= np.sign(np.linalg.det(q))
    pose = (q, rng.normal(size=3))
    with tl3d.FusionContext(w, h, 120.0, 118.0, 64.5, 47.0, n_slots=2) as ctx:
        ctx.upload(0, depth, None)
        ctx.upload(1, np.zeros((h, w), np.float32), None)
        for sub in (1, 2, 3):
            for p, sc in ((pose, 1.0), (None, 0.7)):
                pts, _ = ctx.backproject(0, pose=p, scale=sc, subsample=sub)
                lo, hi = ctx.frame_bounds(0, pose=p, scale=sc, subsample=sub)
                assert np.array_equal(lo, pts.min(0).astype(np.float64)) and np.array_equal(hi, pts.max(0).astype(np.float64))
        lo, hi = ctx.frame_bounds(1, pose=pose)
        assert np.all(np.isinf(lo)) and np.all(lo > 0) and np.all(np.isinf(hi)) and np.all(hi < 0)


def test_frames_bounds_is_the_union_of_the_frames_extents():
    rng = np.random.default_rng(9)
    h, w, n = 60, 80, 37                                   # more frames than one read-back round holds
    with tl3d.FusionContext(w, h, 70.0, 72.0, 40.0, 30.0, n_slots=n) as ctx:
        poses, lo, hi = [], np.full(3, np.inf), np.full(3, -np.inf)
        scales = 0.5 + rng.random(n)
        for i in range(n):
            d = (0.3 + 3.0 * rng.random((h, w))).astype(np.float32)
            d[rng.random((h, w)) < 0.2] = 0.0
            ctx.upload(i, d, None)
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            q *= np.sign(np.linalg.det(q))
            poses.append((q, rng.normal(size=3)))
            a, b = ctx.frame_bounds(i, pose=poses[-1], scale=scales[i], subsample=2)
            lo, hi = np.minimum(lo, a), np.maximum(hi, b)
        got = ctx.frames_bounds(list(range(n)), poses, scales, subsample=2)
    assert np.array_equal(got[0], lo) and np.array_equal(got[1], hi)
