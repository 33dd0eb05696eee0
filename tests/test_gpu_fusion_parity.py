"""GPU parity proper: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.
Integer grids are compared bit for bit."""
import numpy as np
import pytest

import tl3d
from helpers import SMALL, make_pair, small_scene_frames, ulp_diff

pytestmark = pytest.mark.gpu


def test_tsdf_grid_bit_exact():
    poses, frames = small_scene_frames(n=4, deg=6.0)
    ctx, orc = make_pair(dims=(96, 64, 80), voxel=0.03, centre=(0.0, -0.2, 0.1))
    with ctx:
        for i, ((depth, bgr), pose) in enumerate(zip(frames, poses)):
            ctx.upload(i % 4, depth, bgr)
            ctx.integrate(i % 4, pose)
            orc.tsdf_integrate(depth, pose[0], pose[1])
        g = ctx.download_grid(tl3d.CH_TSDF)
    assert g.shape == orc.tsdf.shape
    assert orc.tsdf[:, 1].sum() > 10000, "scene must touch the grid"
    assert np.array_equal(g, orc.tsdf)


def test_tsdf_scale_and_limits():
    poses, frames = small_scene_frames(n=2, deg=3.0)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.04, min_depth=0.3, max_depth=1.6)
    with ctx:
        for i, ((depth, bgr), pose) in enumerate(zip(frames, poses)):
            d = (depth / 1.25).astype(np.float32)
            ctx.upload(i, d, None)
            ctx.integrate(i, pose, scale=1.25)
            orc.tsdf_integrate(d, pose[0], pose[1], scale=1.25)
        g = ctx.download_grid(tl3d.CH_TSDF)
    assert orc.tsdf[:, 1].sum() > 1000
    assert np.array_equal(g, orc.tsdf)


def test_tsdf_counting_mode_matches_and_counts():
    poses, frames = small_scene_frames(n=2, deg=3.0)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.04)
    with ctx:
        ctx.set_profile(count_records=True, time_kernels=True)
        for i, ((depth, bgr), pose) in enumerate(zip(frames, poses)):
            ctx.upload(i, depth, None)
            ctx.integrate(i, pose)
            orc.tsdf_integrate(depth, pose[0], pose[1])
        g = ctx.download_grid(tl3d.CH_TSDF)
        st = ctx.stats()
    assert np.array_equal(g, orc.tsdf)
    updates = int(orc.tsdf[:, 1].sum())
    assert st["tsdf_launches"] == 2 and st["tsdf_kernel_timed"] == 2 and st["tsdf_kernel_ms"] > 0
    # every voxel update is one 8-byte record read + written by the update kernel, or 1/512 of a counted free-space brick
    assert st["tsdf_records_written"] + 512 * st["tsdf_bricks_free_counted"] == updates
    assert st["tsdf_records_read"] == st["tsdf_records_written"]
    assert 0 < st["tsdf_bricks_visited"] <= 2 * 512


def test_centroid_grid_bit_exact_and_extract():
    poses, frames = small_scene_frames(n=3, deg=5.0)
    ctx, orc = make_pair(dims=(128, 96, 128), voxel=0.02, centre=(0.0, -0.3, 0.0))
    with ctx:
        for i, ((depth, bgr), pose) in enumerate(zip(frames, poses)):
            ctx.upload(i, depth, bgr)
            sub = 1 + (i % 2)
            ctx.accumulate_centroid(i, pose, subsample=sub)
            orc.centroid_accumulate(depth, bgr, pose[0], pose[1], subsample=sub)
        g = ctx.download_grid(tl3d.CH_CENTROID)
        st = ctx.stats()
        xyz, rgb = ctx.extract(tl3d.EXTRACT_CENTROID)
    assert np.array_equal(g, orc.centroid)
    assert st["centroid_points"] == orc.n_acc.value and st["centroid_dropped"] == orc.n_drop.value
    assert orc.n_acc.value > 10000
    oxyz, orgb = orc.extract(0)
    assert len(xyz) == len(oxyz) > 1000
    assert np.array_equal(xyz, oxyz) and np.array_equal(rgb, orgb)


def test_extract_tsdf_mode_and_gate():
    poses, frames = small_scene_frames(n=6, deg=8.0)
    ctx, orc = make_pair(dims=(96, 96, 96), voxel=0.025, centre=(0.0, -0.2, 0.0))
    with ctx:
        for i, ((depth, bgr), pose) in enumerate(zip(frames, poses)):
            ctx.upload(i % 4, depth, bgr)
            ctx.integrate(i % 4, pose)
            ctx.accumulate_centroid(i % 4, pose)
            orc.tsdf_integrate(depth, pose[0], pose[1])
            orc.centroid_accumulate(depth, bgr, pose[0], pose[1])
        xyz, rgb = ctx.extract(tl3d.EXTRACT_TSDF, min_weight=2)
        gx, gc = ctx.extract(tl3d.EXTRACT_CENTROID, min_count=2, min_weight=2, max_abs_tsdf=0.9)
    oxyz, orgb = orc.extract(1, min_weight=2)
    assert len(oxyz) > 500
    assert np.array_equal(xyz, oxyz) and np.array_equal(rgb, orgb)
    ogx, ogc = orc.extract(0, min_count=2, min_weight=2, max_abs_tsdf=0.9)
    assert len(ogx) > 100
    assert np.array_equal(gx, ogx) and np.array_equal(gc, ogc)


def test_grid_add_upload_roundtrip():
    poses, frames = small_scene_frames(n=2, deg=5.0)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.04)
    with ctx:
        ctx.upload(0, *frames[0])
        ctx.integrate(0, poses[0])
        ctx.accumulate_centroid(0, poses[0])
        t0, c0 = ctx.download_grid(tl3d.CH_TSDF), ctx.download_grid(tl3d.CH_CENTROID)
        ctx.reset()
        ctx.upload(1, *frames[1])
        ctx.integrate(1, poses[1])
        ctx.accumulate_centroid(1, poses[1])
        ctx.add_grid(tl3d.CH_TSDF, t0)           # merge of two partial grids == both frames in one grid
        ctx.add_grid(tl3d.CH_CENTROID, c0)
        t01, c01 = ctx.download_grid(tl3d.CH_TSDF), ctx.download_grid(tl3d.CH_CENTROID)
    for (d, b), p in zip(frames, poses):
        orc.tsdf_integrate(d, p[0], p[1])
        orc.centroid_accumulate(d, b, p[0], p[1])
    assert np.array_equal(t01, orc.tsdf) and np.array_equal(c01, orc.centroid)


def test_normals_and_icp_match_oracle():
    poses, frames = small_scene_frames(n=2, deg=1.5)
    ctx, orc = make_pair(channels=0, dims=(8, 8, 8))
    with ctx:
        ctx.upload(0, *frames[0])
        ctx.upload(1, *frames[1])
        ctx.build_normals(1, depth_jump=0.05)
        nm = ctx.download_normals(1)
        onm = orc.normals(frames[1][0], depth_jump=0.05)
        assert (onm[..., 3] > 0).mean() > 0.5
        assert np.array_equal(nm, onm)
        res = ctx.icp(0, 1, iters=15, stride=2, max_dist=0.1)
        ores = orc.icp(frames[0][0], onm, iters=15, stride=2, max_dist=0.1)
    assert res["n_src"] == ores["n_src"] and abs(res["n_corr"] - ores["n_corr"]) <= 2
    assert np.linalg.norm(res["T"] - ores["T"]) <= 1e-4        # north-star tolerance: pose within 1e-4 Frobenius
    assert np.linalg.norm(res["T"] - ores["T"]) <= 1e-9        # what identical association actually gives
    assert abs(res["rmse"] - ores["rmse"]) < 1e-9 and res["iters_run"] == ores["iters_run"]
    # and both recover the analytic relative pose
    from tl3d import synth
    r_rel, t_rel = synth.relative_pose(poses[0], poses[1])
    T_true = np.eye(4); T_true[:3, :3] = r_rel; T_true[:3, 3] = t_rel.ravel()
    assert np.linalg.norm(res["T"] - T_true) < 5e-3


def test_tsdf_culling_is_conservative_for_arbitrary_views():
    """Brick culling (frustum + depth tiles) must never drop a voxel the per-voxel rule updates: random poses with the
    camera inside / outside / grazing the grid, invalid-depth holes, depth discontinuities, tiny and huge depths."""
    from tl3d import synth
    rng = np.random.default_rng(5)
    cam = dict(width=200, height=152, fx=150.0, fy=160.0, cx=101.3, cy=70.7)
    ctx, orc = make_pair(cam=cam, dims=(64, 48, 56), voxel=0.05, centre=(0.1, -0.1, 0.3), trunc=0.12, n_slots=1)
    scene = synth.object_scene(with_room=True)
    with ctx:
        for trial in range(24):
            eye = rng.uniform([-1.6, -1.2, -1.6], [1.6, 0.3, 1.6])
            tgt = rng.uniform(-0.5, 0.5, size=3)
            pose = synth.look_at(eye, tgt)
            depth, _ = synth.render(scene, pose, want_color=False, **cam)
            depth = depth.copy()
            if trial % 3 == 0:       # holes and out-of-range stripes
                depth[rng.random(depth.shape) < 0.2] = 0.0
                depth[:, 50:60] = 80.0
            if trial % 4 == 1:       # foreground occluder: sharp discontinuity
                depth[40:90, 60:140] = 0.25
            if trial % 5 == 2:
                depth[:] = np.nan
                depth[70:80, 90:100] = 0.9
            ctx.upload(0, depth, None)
            ctx.integrate(0, pose)
            orc.tsdf_integrate(depth, pose[0], pose[1])
        g = ctx.download_grid(tl3d.CH_TSDF)
    assert orc.tsdf[:, 1].sum() > 100000
    assert np.array_equal(g, orc.tsdf)


def test_grid_tensor_is_a_zero_copy_view():
    """torch.distributed all-reduces the library's grid memory in place (bench.py, tl3d.distributed)."""
    import torch
    poses, frames = small_scene_frames(n=1)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.04)
    with ctx:
        ctx.upload(0, *frames[0])
        ctx.integrate(0, poses[0])
        ctx.accumulate_centroid(0, poses[0])
        ctx.sync()
        t = ctx.grid_tensor(tl3d.CH_TSDF)
        c = ctx.grid_tensor(tl3d.CH_CENTROID)
        assert t.dtype == torch.int32 and t.numel() == 2 * 64 ** 3 and c.dtype == torch.int64 and c.numel() == 4 * 64 ** 3
        g = ctx.download_grid(tl3d.CH_TSDF)
        assert np.array_equal(t.cpu().numpy().reshape(-1, 2), g)
        t.mul_(2)                                   # what a 2-rank all-reduce of identical grids would leave
        c.mul_(2)
        torch.cuda.synchronize()
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), 2 * g)
        xyz, _ = ctx.extract(tl3d.EXTRACT_CENTROID)
    orc.centroid_accumulate(frames[0][0], frames[0][1], poses[0][0], poses[0][1])
    oxyz, _ = orc.extract(0)
    assert np.array_equal(xyz, oxyz)                # doubled sums and doubled counts: same centroids


def test_tsdf_lane_mappings_agree_for_rolled_and_top_down_cameras():
    """The MIXED path picks its lane->voxel mapping from the pose (which grid axis is vertical in the image):
    exercise all three choices -- upright, rolled 90 degrees (x vertical), looking straight down (z vertical)."""
    from tl3d import synth
    cam = dict(width=200, height=152, fx=150.0, fy=160.0, cx=101.3, cy=70.7)
    ctx, orc = make_pair(cam=cam, dims=(64, 56, 64), voxel=0.04, centre=(0.0, -0.1, 0.0), n_slots=1)
    scene = synth.object_scene(with_room=True)
    rz = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    poses = []
    for eye in ((0.9, 0.0, -0.4), (-0.3, -0.2, 0.95)):
        r, t = synth.look_at(eye, (0.0, 0.0, 0.0))
        poses += [(r, t), (rz @ r, rz @ t), (rz @ rz @ r, rz @ rz @ t)]           # upright, rolled 90, rolled 180
    poses.append(synth.look_at((0.05, -1.0, 0.02), (0.0, 0.3, 0.0), up=(0.0, 0.0, 1.0)))   # top-down
    kinds = set()
    with ctx:
        for pose in poses:
            a = np.abs(pose[0][1])
            kinds.add(int(np.argmax(a)))
            depth, _ = synth.render(scene, pose, want_color=False, **cam)
            ctx.upload(0, depth, None)
            ctx.integrate(0, pose)
            orc.tsdf_integrate(depth, pose[0], pose[1])
        g = ctx.download_grid(tl3d.CH_TSDF)
    assert kinds == {0, 1, 2}, kinds
    assert orc.tsdf[:, 1].sum() > 100000 and np.array_equal(g, orc.tsdf)


@pytest.mark.parametrize("width,height,radius", [(161, 119, 0), (162, 120, 1), (163, 121, 1)])
def test_normal_maps_in_phase_major_rows_give_the_oracle_for_any_width(width, height, radius):
    """The library keeps normal maps and window-averaged depth in phase-major rows (pixel u at (u & 3) * ceil(W / 4) + (u >> 2)):
    widths that are not multiples of 4, plain and smoothed normals, host and device destinations of tl3d_download_normals, and
    both registration kernels reading through that layout (per-iteration and batched; strides 1, 2, 4) against the oracle."""
    import torch
    from oracle import c_oracle
    from tl3d import synth
    cam = dict(width=width, height=height, fx=150.0, fy=150.0, cx=0.5 * width, cy=0.5 * height)
    poses, frames = small_scene_frames(n=3, deg=1.5, cam=cam)
    ctx = tl3d.FusionContext(width, height, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=3, grid=None)
    orc = c_oracle.Oracle(width, height, cam["fx"], cam["fy"], cam["cx"], cam["cy"], 0.1, 50.0, dims=(8, 8, 8), origin=(0.0, 0.0, 0.0),
                          voxel_size=0.02, sdf_trunc=0.08)
    with ctx:
        for i, f in enumerate(frames):
            ctx.upload(i, *f)
        ctx.set_normal_smoothing(radius)
        for i in range(3):
            ctx.build_normals(i, depth_jump=0.05)
        onm = [orc.normals_smooth(f[0], radius, depth_jump=0.05)[1] if radius else orc.normals(f[0], depth_jump=0.05) for f in frames]
        dev = torch.empty((height, width, 4), dtype=torch.float32, device="cuda")
        for i in range(3):
            assert np.array_equal(ctx.download_normals(i), onm[i])
            ctx.download_normals(i, out=dev)
            assert np.array_equal(dev.cpu().numpy(), onm[i])
        for stride in (1, 2, 4):
            lv = [dict(iters=6, stride=stride, max_dist=0.1, eps=0.0)]
            one = ctx.icp(0, 1, iters=6, stride=stride, max_dist=0.1, eps=0.0)
            batch = ctx.icp_batch([(0, 1), (1, 2)], lv)
            src = orc.normals_smooth(frames[0][0], radius, depth_jump=0.05)[0] if radius else frames[0][0]
            ores = orc.icp(src, onm[1], iters=6, stride=stride, max_dist=0.1, eps=0.0)
            for res in (one, batch[0]):
                assert res["n_src"] == ores["n_src"] and abs(res["n_corr"] - ores["n_corr"]) <= 2
                assert np.linalg.norm(res["T"] - ores["T"]) <= 1e-8


def test_icp_lanes_match_blocking_calls():
    """Asynchronous ICP lanes (independent pairs in flight on separate streams) give the blocking call's result."""
    poses, frames = small_scene_frames(n=6, deg=1.5)
    ctx, orc = make_pair(channels=0, dims=(8, 8, 8), n_slots=6)
    with ctx:
        for i, f in enumerate(frames):
            ctx.upload(i, *f)
            ctx.build_normals(i)
        ref = [ctx.icp(i, i + 1, iters=8, stride=2, max_dist=0.1) for i in range(5)]
        for i in range(5):
            ctx.icp_enqueue(i, i, i + 1, iters=8, stride=2, max_dist=0.1)
        with pytest.raises(tl3d.Tl3dError):
            ctx.icp_enqueue(0, 0, 1)                 # lane 0 still holds an uncollected run
        got = [ctx.icp_collect(i) for i in range(5)]
        with pytest.raises(tl3d.Tl3dError):
            ctx.icp_collect(0)                       # nothing left to collect
    for a, b in zip(ref, got):
        assert np.array_equal(a["T"], b["T"]) and a["n_corr"] == b["n_corr"] and a["iters_run"] == b["iters_run"]


def test_fusion_is_deterministic_and_order_free():
    """Integer accumulators: the same frames give the same grids bit for bit on every run and in any order
    (a float-atomics design would not), which is also what makes the multi-GPU merge exact."""
    poses, frames = small_scene_frames(n=5, deg=7.0)
    grids = []
    for order in ([0, 1, 2, 3, 4], [0, 1, 2, 3, 4], [4, 2, 0, 3, 1]):
        ctx, _ = make_pair(dims=(96, 96, 96), voxel=0.025, centre=(0.0, -0.2, 0.0), n_slots=5)
        with ctx:
            for i in order:
                ctx.upload(i, *frames[i])
                ctx.integrate(i, poses[i])
                ctx.accumulate_centroid(i, poses[i], subsample=1 + (i & 1))
            grids.append((ctx.download_grid(tl3d.CH_TSDF), ctx.download_grid(tl3d.CH_CENTROID)))
    for t, c in grids[1:]:
        assert np.array_equal(t, grids[0][0]) and np.array_equal(c, grids[0][1])


def test_checkpoint_resume_equals_uninterrupted_run():
    """Grid state is plain integers: download -> new context -> upload -> keep fusing == never stopping (and == oracle)."""
    poses, frames = small_scene_frames(n=4, deg=4.0)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.04)
    with ctx:
        for i in range(2):
            ctx.upload(i, *frames[i])
            ctx.integrate(i, poses[i])
            ctx.accumulate_centroid(i, poses[i])
        ck_t, ck_c = ctx.download_grid(tl3d.CH_TSDF), ctx.download_grid(tl3d.CH_CENTROID)
    ctx2, _ = make_pair(dims=(64, 64, 64), voxel=0.04)
    with ctx2:
        ctx2.upload_grid(tl3d.CH_TSDF, ck_t)
        ctx2.upload_grid(tl3d.CH_CENTROID, ck_c)
        for i in range(2, 4):
            ctx2.upload(i, *frames[i])
            ctx2.integrate(i, poses[i])
            ctx2.accumulate_centroid(i, poses[i])
        t, c = ctx2.download_grid(tl3d.CH_TSDF), ctx2.download_grid(tl3d.CH_CENTROID)
    for (d, b), p in zip(frames, poses):
        orc.tsdf_integrate(d, p[0], p[1])
        orc.centroid_accumulate(d, b, p[0], p[1])
    assert np.array_equal(t, orc.tsdf) and np.array_equal(c, orc.centroid)


def test_point_list_accumulation_and_bounds_match_oracle():
    """tl3d_accumulate_points / tl3d_points_bounds: the merge entry points on caller-supplied clouds (D2R:386-420)."""
    rng = np.random.default_rng(11)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.02, channels=tl3d.CH_CENTROID)
    xyz = rng.uniform(-0.7, 0.7, (50000, 3)).astype(np.float32)        # some points fall outside the 1.28 m cube
    xyz[:2000] = xyz[2000:4000]                                        # exact duplicates -> same voxel, same offsets
    rgb = rng.integers(0, 256, (50000, 3), dtype=np.uint8)
    with ctx:
        ctx.accumulate_points(xyz, rgb)
        ctx.accumulate_points(xyz[:0], rgb[:0])                        # empty list is a no-op
        g = ctx.download_grid(tl3d.CH_CENTROID)
        st = ctx.stats()
        mn, mx = ctx.points_bounds(xyz)
    orc.centroid_points(xyz, rgb)
    assert np.array_equal(g, orc.centroid)
    assert st["centroid_points"] == orc.n_acc.value and st["centroid_dropped"] == orc.n_drop.value > 0
    assert np.array_equal(mn, xyz.min(0).astype(np.float64)) and np.array_equal(mx, xyz.max(0).astype(np.float64))


def test_deferred_update_batches_never_change_results():
    """tl3d_integrate defers its update launches to batch boundaries; every consumer issues them first.  Odd frame counts,
    slot re-use inside a batch (forces an early flush), interleaved centroid calls and a zero-copy torch view of the
    grid all see exactly the oracle's grid."""
    import torch
    poses, frames = small_scene_frames(n=11, deg=3.0)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.04, n_slots=3)
    with ctx:
        for i, ((d, b), p) in enumerate(zip(frames, poses)):
            ctx.upload(i % 3, d, b)                     # slot re-used every 3 frames while updates may be pending
            ctx.integrate(i % 3, p)
            ctx.accumulate_centroid(i % 3, p)
        view = ctx.grid_tensor(tl3d.CH_TSDF)            # issues the outstanding updates, then exposes the memory
        torch.cuda.synchronize()
        via_view = view.cpu().numpy().reshape(-1, 2)
        t, c = ctx.download_grid(tl3d.CH_TSDF), ctx.download_grid(tl3d.CH_CENTROID)
        assert ctx.stats()["tsdf_launches"] == 11
    for (d, b), p in zip(frames, poses):
        orc.tsdf_integrate(d, p[0], p[1])
        orc.centroid_accumulate(d, b, p[0], p[1])
    assert np.array_equal(t, orc.tsdf) and np.array_equal(c, orc.centroid) and np.array_equal(via_view, orc.tsdf)


def test_extract_size_query_is_reused_only_while_the_grid_is_untouched():
    """tl3d_extract is called twice (size, then buffers); the second call skips the counting pass unless something touched
    the grid or the shared scratch in between."""
    import ctypes as C
    from tl3d import _cabi as abi
    poses, frames = small_scene_frames(n=2, deg=6.0)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.04)
    with ctx:
        ctx.upload(0, *frames[0])
        ctx.upload(1, *frames[1])
        ctx.accumulate_centroid(0, poses[0])
        n = C.c_int64(0)
        abi.check(ctx._lib.tl3d_extract(ctx._h, 0, 1, 0, 1.0, None, None, 0, C.byref(n)))
        n_first = n.value
        a1, _ = ctx.extract()                                   # query + extraction back to back: counts reused
        assert len(a1) == n_first
        abi.check(ctx._lib.tl3d_extract(ctx._h, 0, 1, 0, 1.0, None, None, 0, C.byref(n)))       # query ...
        ctx.accumulate_centroid(1, poses[1])                    # ... grid changes ...
        ctx.backproject(0, pose=poses[0])                       # ... and the scratch is borrowed
        cap = n.value + 200000
        xyz, rgb = np.empty((cap, 3), np.float32), np.empty((cap, 3), np.uint8)
        abi.check(ctx._lib.tl3d_extract(ctx._h, 0, 1, 0, 1.0, abi.ptr(xyz), abi.ptr(rgb), cap, C.byref(n)))
    for (d, b), p in zip(frames, poses):
        orc.centroid_accumulate(d, b, p[0], p[1])
    ox, oc = orc.extract(0)
    assert n.value == len(ox) > n_first
    assert np.array_equal(xyz[:n.value], ox) and np.array_equal(rgb[:n.value], oc)


def test_tsdf_from_16bit_millimetre_frames_is_bit_exact():
    """Frames uploaded as uint16 millimetres (the reference's 16-bit PNG depth, D2R:85-90): the TSDF kernels gather from
    the 16-bit image and convert exactly as the upload conversion does, so the grid equals the oracle run
    on the converted f32 frames -- for every lane mapping, with holes (0) and saturated pixels (65535)."""
    from tl3d import synth
    rng = np.random.default_rng(9)
    cam = dict(width=200, height=152, fx=150.0, fy=160.0, cx=101.3, cy=70.7)
    ctx, orc = make_pair(cam=cam, dims=(64, 48, 56), voxel=0.05, centre=(0.1, -0.1, 0.3), trunc=0.12, n_slots=2, max_depth=60.0)
    scene = synth.object_scene(with_room=True)
    views = [((1.0, -0.2, 0.3), (0, 0, 0), (0, 1, 0)), ((0.2, -0.3, -1.2), (0, 0, 0), (0, 1, 0)),
             ((0.1, -1.1, 0.2), (0, 0, 0), (1, 0, 0)), ((0.9, -0.6, 0.7), (0.1, 0, 0), (0.3, 1, 0.2))]
    with ctx:
        for i, (eye, tgt, up) in enumerate(views):
            pose = synth.look_at(np.array(eye), np.array(tgt), up=np.array(up, float))
            depth, _ = synth.render(scene, pose, want_color=False, **cam)
            mm = np.clip(np.rint(depth * 1000.0), 0, 65535).astype(np.uint16)
            mm[rng.random(mm.shape) < 0.05] = 0
            mm[10:14, 20:60] = 65535
            ctx.upload(i & 1, mm, None)
            assert np.array_equal(ctx.download_depth(i & 1), mm.astype(np.float32) / np.float32(1000.0))
            ctx.integrate(i & 1, pose)
            orc.tsdf_integrate(mm.astype(np.float32) / np.float32(1000.0), pose[0], pose[1])
        g = ctx.download_grid(tl3d.CH_TSDF)
    assert orc.tsdf[:, 1].sum() > 50000
    assert np.array_equal(g, orc.tsdf)


def test_tsdf_int32_headroom_is_guarded_not_wrapped():
    """A voxel's sum of rint(tsdf * 32767) stays inside int32 for TL3D_TSDF_MAX_WEIGHT = 65536 observations (free-space
    voxels add +32767 every frame).  tl3d_integrate / tl3d_grid_add refuse the step that could wrap instead of wrapping."""
    from tl3d import _cabi as abi
    cam = dict(width=64, height=48, fx=60.0, fy=60.0, cx=31.5, cy=23.5)
    depth = np.full((48, 64), 3.0, np.float32)                        # a wall 3 m away: the whole grid is free space
    spec = tl3d.GridSpec.cube(16, 0.02, centre=(0.0, 0.0, 1.0), channels=tl3d.CH_TSDF)
    pose = (np.eye(3), np.zeros(3))
    with tl3d.FusionContext(n_slots=1, grid=spec, **cam) as ctx:
        ctx.upload(0, depth, None)
        ctx.integrate(0, pose)
        g = ctx.download_grid(tl3d.CH_TSDF)
        assert (g[:, 1] == 1).all() and (g[:, 0] == 32767).all() and ctx.max_weight() == 1
        g[:, 1] = abi.TSDF_MAX_WEIGHT - 1
        g[:, 0] = 32767 * (abi.TSDF_MAX_WEIGHT - 1)
        ctx.upload_grid(tl3d.CH_TSDF, g)                              # contents unknown to the library: re-measured lazily
        ctx.integrate(0, pose)                                        # 65536th observation: still representable
        g2 = ctx.download_grid(tl3d.CH_TSDF)
        assert (g2[:, 1] == abi.TSDF_MAX_WEIGHT).all() and (g2[:, 0].astype(np.int64) == 32767 * abi.TSDF_MAX_WEIGHT).all()
        with pytest.raises(tl3d.Tl3dError) as e:
            ctx.integrate(0, pose)
        assert e.value.code == abi.E_STATE and "overflow" in str(e.value)
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), g2)    # refused, nothing changed
        # merges: 40000 + 30000 observations do not fit, 30000 + 30000 do
        ctx.reset()
        a = np.zeros_like(g); a[:, 1] = 40000; a[:, 0] = 40000 * 32767 // 2
        b = np.zeros_like(g); b[:, 1] = 30000; b[:, 0] = -30000 * 32767 // 2
        ctx.upload_grid(tl3d.CH_TSDF, a)
        with pytest.raises(tl3d.Tl3dError) as e:
            ctx.add_grid(tl3d.CH_TSDF, b)
        assert e.value.code == abi.E_STATE
        ctx.upload_grid(tl3d.CH_TSDF, b)
        ctx.add_grid(tl3d.CH_TSDF, b)
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), 2 * b) and ctx.max_weight() == 60000


def test_slot_rewrites_are_ordered_behind_uncollected_icp_runs():
    """An ICP lane reads its source slot's depth and its target slot's normal map until it is collected.  Re-uploading
    into the source slot (a ring of fewer slots than frames) or rebuilding the target's normals meanwhile must not change
    the run's result: the library orders those writes behind the run on the device."""
    cam = dict(width=640, height=480, fx=525.0, fy=525.0, cx=320.0, cy=240.0)
    from tl3d import synth
    scene = synth.object_scene(with_room=True)
    poses = synth.orbit_poses(3, 1.0, 2.0)
    frames = [synth.render(scene, p, want_color=False, **cam)[0] for p in poses]
    other = np.roll(frames[2], 37, axis=1) * np.float32(1.3)          # something quite different
    kw = dict(iters=30, stride=1, max_dist=0.1, eps=0.0)
    with tl3d.FusionContext(n_slots=2, grid=None, **cam) as ctx:
        ctx.upload(0, frames[0], None)
        ctx.upload(1, frames[1], None)
        ctx.build_normals(1)
        ref = ctx.icp(0, 1, **kw)
        for _ in range(3):
            ctx.upload(0, frames[0], None)
            ctx.upload(1, frames[1], None)
            ctx.build_normals(1)
            ctx.icp_enqueue(3, 0, 1, **kw)
            ctx.upload(0, other, None)              # rewrites the depth the run is reading
            ctx.upload(1, other, None)
            ctx.build_normals(1)                    # rewrites the normal map the run is reading
            got = ctx.icp_collect(3)
            assert np.array_equal(got["T"], ref["T"]) and got["n_corr"] == ref["n_corr"] and got["iters_run"] == ref["iters_run"]


def test_streams_run_side_by_side_without_scheduler_stalls():
    """`import tl3d` sets GPU_MAX_HW_QUEUES = 16 before the process's first HIP call (the runtime reads it once).  With
    ROCm's default of 4 the 16 ICP lanes would share 4 hardware queues; from 24 live queues on, the GPU's queue slots are
    over-subscribed and every hand-over can stall for a ~10 ms scheduler quantum.  The probe runs
    n one-wave kernels of `spin` ms on n fresh streams: ceil(n / Q) rounds when healthy."""
    from tl3d import _cabi as abi
    one = abi.probe_hw_queues(0, 1, 0.5)
    assert 0.45 < one["elapsed_ms"] < 2.0, one                      # the spin kernel keeps the 100 MHz clock
    info = abi.probe_hw_queues(0, 16, 0.5)
    assert info["GPU_MAX_HW_QUEUES"] is not None and 1 <= info["effective_queues"] <= 16
    if int(info["GPU_MAX_HW_QUEUES"]) == 16:
        assert info["elapsed_ms"] < 1.6, info                       # 16 streams, 16 queues: one round (+ launch overhead)
        many = abi.probe_hw_queues(0, 48, 0.25)
        assert many["elapsed_ms"] < 4.0, many                       # 3 rounds; the over-subscribed regime takes 10-20 ms


def test_free_space_counters_fold_to_the_same_grid():
    """Free-space bricks are counted (one add per brick and frame) and folded into the records before anything reads the
    TSDF channel: download, merge, extraction, weight check and checkpoint see the oracle's grid bit for bit, and the counting
    mode reports which bricks were counted instead of streamed."""
    cam = dict(width=320, height=240, fx=280.0, fy=280.0, cx=159.5, cy=119.5)
    poses, frames = small_scene_frames(n=6, deg=9.0, scene=None, cam=cam)
    ctx, orc = make_pair(cam=cam, dims=(160, 136, 160), voxel=0.0125, centre=(0.0, -0.2, 0.0), n_slots=6, channels=tl3d.CH_TSDF)
    with ctx:
        ctx.set_profile(True, False)
        for i, (d, c) in enumerate(frames):
            ctx.upload(i, d, c)
        for i in range(3):
            ctx.integrate(i, poses[i])
            orc.tsdf_integrate(frames[i][0], poses[i][0], poses[i][1])
        st = ctx.stats()
        assert st["tsdf_bricks_free"] >= 300 and st["tsdf_bricks_free_counted"] == st["tsdf_bricks_free"], st["tsdf_bricks_free"]
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), orc.tsdf)                    # folded by the download
        for i in range(3, 6):                                                               # more frames on top of folded + new counts
            ctx.integrate(i, poses[i])
            orc.tsdf_integrate(frames[i][0], poses[i][0], poses[i][1])
        assert ctx.max_weight() == int(orc.tsdf[:, 1].max())                                # folded by the weight check
        xyz, _ = ctx.extract(tl3d.EXTRACT_TSDF, min_weight=2)                               # and by the extraction
        oxyz, _ = orc.extract(1, min_weight=2, use_centroid=False)
        assert len(xyz) > 100 and np.array_equal(xyz, oxyz)
        g = ctx.download_grid(tl3d.CH_TSDF)
        assert np.array_equal(g, orc.tsdf)
        ctx.add_grid(tl3d.CH_TSDF, g)                                                       # merge of two folded grids
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), 2 * orc.tsdf)
        ctx.reset()
        ctx.integrate(0, poses[0])
        o2 = c_oracle_like(orc)
        o2.tsdf_integrate(frames[0][0], poses[0][0], poses[0][1])
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), o2.tsdf)                     # counters were cleared by the reset


def c_oracle_like(orc):
    from oracle import c_oracle
    c = orc.cfg
    return c_oracle.Oracle(c.width, c.height, c.fx, c.fy, c.cx, c.cy, c.min_depth, c.max_depth, dims=(c.nx, c.ny, c.nz),
                           origin=tuple(c.origin), voxel_size=c.voxel_size, sdf_trunc=c.sdf_trunc)


def test_c_abi_rccl_merge_single_rank():
    """tl3d_rccl_unique_id / tl3d_rccl_init / tl3d_allreduce_grid: the merge step for hosts without torch.distributed, RCCL
    resolved at run time.  One rank on the one GPU of this box: the all-reduce is the identity, the int32 headroom check
    runs over 'all' ranks, the grid stays the oracle's; a second join is refused."""
    poses, frames = small_scene_frames(n=3, deg=6.0)
    ctx, orc = make_pair(dims=(64, 64, 64), voxel=0.04)
    with ctx:
        for i, (d, c) in enumerate(frames):
            ctx.upload(i, d, c)
            ctx.integrate(i, poses[i])
            ctx.accumulate_centroid(i, poses[i], subsample=2)
            orc.tsdf_integrate(d, poses[i][0], poses[i][1])
            orc.centroid_accumulate(d, c, poses[i][0], poses[i][1], subsample=2)
        with pytest.raises(tl3d.Tl3dError):
            ctx.allreduce_grid()                                   # no communicator yet
        uid = tl3d.FusionContext.rccl_unique_id()
        assert len(uid) == 128 and any(uid)
        ctx.rccl_init(1, 0, uid)
        ctx.allreduce_grid()                                       # both channels; folds the free-space counts first
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), orc.tsdf)
        assert np.array_equal(ctx.download_grid(tl3d.CH_CENTROID), orc.centroid)
        ctx.allreduce_grid(tl3d.CH_TSDF)
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), orc.tsdf)
        with pytest.raises(tl3d.Tl3dError):
            ctx.rccl_init(1, 0, uid)


def test_icp_batch_matches_oracle_and_the_per_level_calls():
    """tl3d_icp_batch_*: every pair through all levels in one launch = the per-level blocking calls a host would chain
    (same stop rule between levels) = the oracle; a pair's result does not depend on the batch it is in."""
    poses, frames = small_scene_frames(n=6, deg=1.5)
    ctx, orc = make_pair(channels=0, dims=(8, 8, 8), n_slots=7)
    levels = [dict(iters=6, stride=4, max_dist=0.2, eps=1e-7), dict(iters=10, stride=2, max_dist=0.1, eps=1e-7)]
    with ctx:
        for i, f in enumerate(frames):
            ctx.upload(i, *f)
            ctx.build_normals(i)
        ctx.upload(6, np.zeros_like(frames[0][0]), None)             # an empty frame: its pair must fail, not hang
        pairs = [(i, i + 1) for i in range(5)] + [(6, 0)]
        T0 = [None, np.eye(4), None, None, None, None]
        chained = []
        for k, (a, b) in enumerate(pairs):
            T, res = np.eye(4), None
            for lv in levels:
                res = ctx.icp(a, b, T_init=T, **lv)
                if res["status"] == 2 or res["n_corr"] < 8:
                    break
                T = res["T"]
            chained.append(res)
        got = ctx.icp_batch(pairs, levels, T_init=T0)
        alone = ctx.icp_batch([pairs[2]], levels)[0]
        single_wg = ctx.icp_batch([(0, 1)], [dict(iters=6, stride=4, max_dist=0.2)])[0]
        ref_single = ctx.icp(0, 1, iters=6, stride=4, max_dist=0.2)
        with pytest.raises(tl3d.Tl3dError):
            ctx.icp_batch_collect()                                  # nothing in flight
        onm = orc.normals(frames[1][0])
        T, ores = np.eye(4), None
        for lv in levels:
            ores = orc.icp(frames[0][0], onm, T_init=T, iters=lv["iters"], stride=lv["stride"], max_dist=lv["max_dist"], eps=lv["eps"])
            T = ores["T"]
    for a, b in zip(chained[:5], got[:5]):
        assert np.linalg.norm(a["T"] - b["T"]) <= 1e-9 and abs(a["n_corr"] - b["n_corr"]) <= 2 and a["n_src"] == b["n_src"]
        # (the sums are added over different trees in the two kernels: a pose entry that rounds to the neighbouring f32 moves rmse by ~1e-9)
        assert a["iters_run"] == b["iters_run"] and a["status"] == b["status"] and abs(a["rmse"] - b["rmse"]) < 1e-8
    # one workgroup per pair (1 200 samples at stride 4): no waiting side at all
    assert np.linalg.norm(single_wg["T"] - ref_single["T"]) <= 1e-9 and single_wg["iters_run"] == ref_single["iters_run"]
    assert got[5]["status"] == 2 and got[5]["n_corr"] == 0 and chained[5]["status"] == 2
    assert np.array_equal(alone["T"], got[2]["T"]) and alone["n_corr"] == got[2]["n_corr"]
    assert np.linalg.norm(got[0]["T"] - ores["T"]) <= 1e-9 and got[0]["iters_run"] == ores["iters_run"] and got[0]["n_src"] == ores["n_src"]


def test_batched_registration_is_bitwise_repeatable_poses_and_statistics():
    """The same batch launched again gives the same bits: poses AND the statistics the host reads afterwards (rmse, fitness,
    correspondences).  The statistics of a batch's LAST pairs used to differ from run to run with identical poses: the sums of
    consecutive passes of a pair are written by different workgroups (different XCDs), and as plain stores the line of the
    last-but-one pass could be written back after the final pass's when the launch ended (they are write-through now)."""
    from tl3d import synth
    W, H = 540, 960
    cam = dict(width=W, height=H, fx=859.5, fy=859.5, cx=270.0, cy=480.0)
    scene = synth.object_scene(with_room=True)
    n = 25
    poses = synth.orbit_poses(n, 1.0, 0.7)
    with tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], n_slots=n, grid=None) as ctx:
        for i, p in enumerate(poses):
            ctx.upload(i, synth.render(scene, p, want_color=False, **cam)[0], None)
        ctx.set_normal_smoothing(1)
        for i in range(n):
            ctx.build_normals(i)
        pairs = [(i - 1, i) for i in range(1, n)]
        for levels in ([dict(iters=10, stride=4, max_dist=0.05, eps=0.0)],
                       [dict(iters=10, stride=4, max_dist=0.2, eps=1e-7), dict(iters=15, stride=2, max_dist=0.05, eps=1e-7)]):
            ref = None
            for _ in range(8):
                res = ctx.icp_batch(pairs, levels)
                now = (np.stack([r["T"] for r in res]),
                       np.array([[r["rmse"], r["fitness"], r["n_corr"], r["n_src"], r["iters_run"], r["status"]] for r in res]))
                if ref is None:
                    ref = now
                assert np.array_equal(now[0], ref[0]) and np.array_equal(now[1], ref[1])
        assert ctx.stats()["icp_batch_timeouts"] == 0


def test_brick_merge_sends_free_space_as_counts_and_records_only_where_there_are_records():
    """The device form of the multi-GPU merge (tl3d.distributed.allreduce_context_grids) on one rank (gloo, world size 1: every
    sum is the identity, so the grid must come out as the oracle's bit for bit): the pending free-space counts are summed as 4 bytes
    per brick and stay pending, only bricks with records travel as records -- far fewer than the bricks a fold-first merge marks --
    and the int32 headroom check sees the counts without folding them."""
    import os
    import torch
    import torch.distributed as dist
    from tl3d.distributed import allreduce_context_grids
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
    from tl3d import synth
    poses, frames = small_scene_frames(n=6, deg=4.0, scene=synth.object_scene(with_room=True))     # every pixel valid: whole bricks of free space
    ctx, orc = make_pair(dims=(128, 128, 128), voxel=0.02, centre=(0.0, -0.1, 0.0), n_slots=6)
    with ctx:
        for i, ((depth, bgr), pose) in enumerate(zip(frames, poses)):
            ctx.upload(i, depth, bgr)
            ctx.integrate(i, pose)
            ctx.accumulate_centroid(i, pose, subsample=2)
            orc.tsdf_integrate(depth, pose[0], pose[1])
            orc.centroid_accumulate(depth, bgr, pose[0], pose[1], subsample=2)
        assert ctx.max_weight() == int(orc.tsdf[:, 1].max())          # counts included, nothing folded
        info = allreduce_context_grids(ctx, dist)
        dev = torch.device("cuda", ctx.device)
        folded = torch.zeros(ctx.n_bricks, dtype=torch.uint8, device=dev)
        g = ctx.download_grid(tl3d.CH_TSDF)                           # (folds)
        c = ctx.download_grid(tl3d.CH_CENTROID)
        ctx.touched_bricks(folded, tl3d.CH_TSDF | tl3d.CH_CENTROID)
        ctx.sync()
        n_folded = int(folded.sum().item())
    assert np.array_equal(g, orc.tsdf) and np.array_equal(c, orc.centroid)
    assert info["bricks_total"] == 16 ** 3 and 0 < info["bricks_sent"] < info["bricks_total"] // 2
    assert n_folded > info["bricks_sent"], (n_folded, info)           # a fold-first merge would also have packed the bricks that hold nothing but a count
    # what travelled: the counts (4 B per brick), two sub-brick maps (8 B per brick each), and the occupied SUB-BRICKS of each channel
    # (512 B / 2 KB each) -- a surface crosses a brick in a few of its eight sub-bricks: far fewer bytes than whole bricks
    nt, nc = info["sub_bricks_tsdf"], info["sub_bricks_centroid"]
    assert info["bytes"] == 4 * info["bricks_total"] + 16 * info["bricks_total"] + 512 * nt + 2048 * nc
    assert 0 < nc < nt <= 8 * info["bricks_sent"] and 2048 * nc < 0.5 * 16384 * info["bricks_sent"]


def test_tsdf_on_ragged_image_sizes_is_bit_exact():
    """Image widths that are not multiples of 4 (scalar loads in the tiles kernel), of 8 (partial level-0 tiles: their 8-B records
    feed the per-voxel test) or of 32 (partial regions), a one-region image, holes and out-of-range depths: f32 and 16-bit
    frames, a batch of several frames per launch, against the every-voxel oracle."""
    from helpers import c_oracle
    from tl3d import synth
    scene = synth.object_scene(with_room=True)
    rng = np.random.default_rng(5)
    for (w, h) in ((173, 131), (97, 64), (31, 29), (200, 152)):
        cam = dict(width=w, height=h, fx=0.9 * w, fy=0.9 * w, cx=0.5 * w - 0.5, cy=0.5 * h - 0.5)
        poses = synth.orbit_poses(5, 1.0, 7.0)
        frames = [synth.render(scene, p, **cam) for p in poses]
        for kind in ("f32", "u16"):
            ctx, orc = make_pair(cam=cam, dims=(96, 96, 96), voxel=0.03, centre=(0.0, -0.1, 0.0), n_slots=5, channels=tl3d.CH_TSDF)
            with ctx:
                for i, ((depth, bgr), pose) in enumerate(zip(frames, poses)):
                    d = depth.copy()
                    holes = rng.random(d.shape)
                    d[holes < 0.03] = 0.0
                    d[(holes > 0.03) & (holes < 0.05)] = 80.0                       # beyond max_depth
                    if kind == "u16":
                        mm = np.clip(np.rint(d * 1000.0), 0, 65535).astype(np.uint16)
                        ctx.upload(i, mm, bgr)
                        d = mm.astype(np.float32) / np.float32(1000.0)
                    else:
                        ctx.upload(i, d, bgr)
                    ctx.integrate(i, pose)
                    orc.tsdf_integrate(d, pose[0], pose[1])
                g = ctx.download_grid(tl3d.CH_TSDF)
            assert orc.tsdf[:, 1].sum() > 1000, (w, h, kind)
            assert np.array_equal(g, orc.tsdf), (w, h, kind)


def test_centroid_runs_of_every_length_and_the_point_list_path():
    """Runs of adjacent samples that share a voxel are summed in the wave (inside 16-lane rows) before the LDS table and
    the grid: voxels of 2 mm ... 40 cm give runs from 1 sample to whole rows (split at row boundaries), invalid pixels cut
    them, strides 1-4 take the table and the direct kernel; the accumulated grid stays the oracle's bit for bit.  Then the
    same cloud through tl3d_accumulate_points (the merge_pointclouds seam)."""
    rng = np.random.default_rng(11)
    poses, frames = small_scene_frames(n=2, deg=3.0)
    for voxel, dims in ((0.002, (192, 192, 192)), (0.05, (64, 64, 64)), (0.4, (16, 16, 16))):
        ctx, orc = make_pair(dims=dims, voxel=voxel, centre=(0.0, -0.2, 0.3), channels=tl3d.CH_CENTROID)
        with ctx:
            for i, ((depth, bgr), pose) in enumerate(zip(frames, poses)):
                depth = depth.copy()
                depth[rng.random(depth.shape) < 0.07] = 0.0                 # holes break runs at random places
                ctx.upload(i, depth, bgr)
                for sub in (1, 2, 3, 4):
                    ctx.accumulate_centroid(i, pose, subsample=sub)
                    orc.centroid_accumulate(depth, bgr, pose[0], pose[1], subsample=sub)
            g = ctx.download_grid(tl3d.CH_CENTROID)
            st = ctx.stats()
            assert np.array_equal(g, orc.centroid), voxel
            assert st["centroid_points"] == orc.n_acc.value and st["centroid_dropped"] == orc.n_drop.value
            # the point-list path: the same points as an explicit cloud
            pts, col = ctx.backproject(0, pose=poses[0], subsample=1)
            ctx.reset()
            ctx.accumulate_points(pts, col)
            g_pts = ctx.download_grid(tl3d.CH_CENTROID)
            ctx.reset()
            ctx.accumulate_centroid(0, poses[0], subsample=1)
            assert np.array_equal(g_pts, ctx.download_grid(tl3d.CH_CENTROID)), voxel


def test_a_batch_of_frames_per_launch_gives_the_oracle_grid_bit_for_bit():
    """tl3d_integrate collects up to 32 frames per update launch (every touched record is read and written once per batch):
    41 frames (a slow orbit, a jump back and two frames 20 degrees apart, then the start again) -> 2 launches, grid == oracle
    == one frame per launch; 16-bit frames take the same path; a batch holds one depth kind."""
    poses, frames = small_scene_frames(n=30, deg=1.0)
    far_poses, far_frames = small_scene_frames(n=2, deg=20.0)
    for as_u16 in (False, True):
        ctx, orc = make_pair(dims=(96, 96, 96), voxel=0.025, centre=(0.0, -0.2, 0.0), n_slots=41, channels=tl3d.CH_TSDF)
        with ctx:
            seq = list(zip(frames, poses)) + list(zip(far_frames, far_poses)) + list(zip(frames[:9], poses[:9]))
            for i, ((d, c), _) in enumerate(seq):
                if as_u16:
                    mm = np.clip(np.round(d * 1000.0), 0, 65535).astype(np.uint16)
                    ctx.upload(i, mm, None)
                    seq[i] = ((mm.astype(np.float32) / np.float32(1000.0), c), seq[i][1])
                else:
                    ctx.upload(i, d, None)
            grids, launches = [], []
            for batching in (True, False):
                ctx.reset()
                ctx.reset_stats()
                ctx.set_tsdf_pairing(batching)
                for i, (_, pose) in enumerate(seq):
                    ctx.integrate(i, pose)
                grids.append(ctx.download_grid(tl3d.CH_TSDF))
                launches.append(ctx.stats()["tsdf_launches"])
            for (d, _), pose in seq:
                orc.tsdf_integrate(d, pose[0], pose[1])
        assert launches == [2, 41], launches                          # 32 + 9 frames; one frame per launch
        assert np.array_equal(grids[0], orc.tsdf) and np.array_equal(grids[1], orc.tsdf), as_u16
    # float32 and 16-bit frames interleaved: the batch is cut at every change of kind, the grid does not care
    ctx, orc = make_pair(dims=(96, 96, 96), voxel=0.025, centre=(0.0, -0.2, 0.0), n_slots=6, channels=tl3d.CH_TSDF)
    with ctx:
        ds = []
        for i in range(6):
            d = frames[i][0]
            if i in (2, 3, 5):
                mm = np.clip(np.round(d * 1000.0), 0, 65535).astype(np.uint16)
                ctx.upload(i, mm, None)
                d = mm.astype(np.float32) / np.float32(1000.0)
            else:
                ctx.upload(i, d, None)
            ds.append(d)
        ctx.reset_stats()
        for i in range(6):
            ctx.integrate(i, poses[i])
            orc.tsdf_integrate(ds[i], poses[i][0], poses[i][1])
        g = ctx.download_grid(tl3d.CH_TSDF)
        assert ctx.stats()["tsdf_launches"] == 4                      # (0,1) (2,3) (4) (5)
    assert np.array_equal(g, orc.tsdf)


def test_fuse_frames_and_normals_many_equal_the_per_frame_calls():
    """tl3d_fuse_frames / tl3d_build_normals_many (one foreign call for a whole sequence) == the per-frame calls in order."""
    poses, frames = small_scene_frames(n=5, deg=2.0)
    grids, nmaps = [], []
    for batched in (False, True):
        ctx, _ = make_pair(dims=(96, 96, 96), voxel=0.025, centre=(0.0, -0.2, 0.0), n_slots=5)
        with ctx:
            for i, (d, c) in enumerate(frames):
                ctx.upload(i, d, c)
            scales = [1.0, 1.0, 0.98, 1.0, 1.02]
            if batched:
                ctx.build_normals_many(list(range(5)), scales)
                ctx.fuse_frames(list(range(5)), poses, scales, centroid_subsample=2)
            else:
                for i in range(5):
                    ctx.build_normals(i, scale=scales[i])
                for i in range(5):
                    ctx.integrate(i, poses[i], scale=scales[i])
                    ctx.accumulate_centroid(i, poses[i], scale=scales[i], subsample=2)
            grids.append((ctx.download_grid(tl3d.CH_TSDF), ctx.download_grid(tl3d.CH_CENTROID)))
            nmaps.append([ctx.download_normals(i) for i in range(5)])
    assert np.array_equal(grids[0][0], grids[1][0]) and np.array_equal(grids[0][1], grids[1][1])
    assert all(np.array_equal(a, b) for a, b in zip(nmaps[0], nmaps[1]))
    assert int((grids[0][0][:, 1] > 0).sum()) > 1000


def test_sim3_icp_matches_the_oracle_and_recovers_the_scale():
    """Row f3, scale as the 7th unknown of the registration: source depth maps divided by s in [0.7, 1.4] -- per-iteration
    kernel and batched kernel vs the C oracle (pose <= 1e-4 Frobenius, scale <= 1e-6 relative), and the recovered scale is
    the true one (<= 2e-3: pixel quantisation of the projective association)."""
    from tl3d import synth
    cam = dict(width=320, height=240, fx=280.0, fy=280.0, cx=159.5, cy=119.5)
    scene = synth.object_scene()
    poses = synth.orbit_poses(4, 1.0, 4.0)
    frames = [synth.render(scene, p, want_color=False, **cam)[0] for p in poses]
    true_s = [1.0, 0.8, 1.3, 0.72]
    rel = [(f / s).astype(np.float32) for f, s in zip(frames, true_s)]
    ctx, orc = make_pair(cam=cam, dims=(64, 64, 64), voxel=0.04, n_slots=4)
    lv = dict(iters=25, stride=2, max_dist=0.5, estimate_scale=True)
    with ctx:
        for i, d in enumerate(rel):
            ctx.upload(i, d, None)
        ctx.build_normals(0, scale=1.0)
        nm0 = orc.normals(rel[0], scale=1.0)
        batch = ctx.icp_batch([(i, 0) for i in (1, 2, 3)], [lv], scales=[1.0, 1.0, 1.0])
        for k, i in enumerate((1, 2, 3)):
            ores = orc.icp(rel[i], nm0, iters=25, stride=2, max_dist=0.5, scale_src=1.0, estimate_scale=True)
            one = ctx.icp(i, 0, iters=25, stride=2, max_dist=0.5, scale_src=1.0, estimate_scale=True)
            for res in (one, batch[k]):
                assert np.linalg.norm(res["T"] - ores["T"]) < 1e-4, (i, np.linalg.norm(res["T"] - ores["T"]))
                assert abs(res["scale"] - ores["scale"]) < 1e-6 * ores["scale"], (i, res["scale"], ores["scale"])
                assert abs(res["n_corr"] - ores["n_corr"]) <= 1e-3 * ores["n_corr"]      # (exp() of device and host differ in the last bit: border samples)
            assert abs(one["scale"] - true_s[i]) < 2e-3 * true_s[i], (i, one["scale"])
            if i == 1:                                                         # (views 2, 3 start 8 / 12 degrees off: parity only)
                r_rel, t_rel = synth.relative_pose(poses[i], poses[0])         # view i -> view 0
                assert np.linalg.norm(one["T"][:3, :3] - r_rel) < 3e-3 and np.linalg.norm(one["T"][:3, 3] - t_rel.ravel()) < 3e-3
        # without the flag the scale is the caller's and the run is the 6-unknown one, bit for bit
        a = ctx.icp(1, 0, iters=10, stride=2, max_dist=0.1, scale_src=0.8)
        b = orc.icp(rel[1], nm0, iters=10, stride=2, max_dist=0.1, scale_src=0.8)
        assert a["scale"] == 0.8 and np.linalg.norm(a["T"] - b["T"]) < 1e-9


def test_sim3_icp_keeps_the_prior_where_the_scale_is_not_observed():
    """A single plane seen from two positions: the plane's distance is one constraint on (translation along the normal,
    scale) -- the combination the data does not see is dropped by the eigenvalue cutoff, so a true prior is kept (and the
    solve takes the eigen path: a plane also leaves two translations and one rotation free)."""
    from tl3d import synth
    cam = dict(width=320, height=240, fx=280.0, fy=280.0, cx=159.5, cy=119.5)
    scene = synth.Scene(planes=[((0.0, 0.0, -1.0), -1.5)])                     # z = 1.5 seen from the origin
    poses = synth.dolly_poses(2, (0.0, 0.0, 0.0), (0.03, 0.0, 0.0))
    frames = [synth.render(scene, p, want_color=False, **cam)[0] for p in poses]
    ctx, orc = make_pair(cam=cam, dims=(64, 64, 64), voxel=0.04, n_slots=2)
    with ctx:
        for i, d in enumerate(frames):
            ctx.upload(i, d, None)
        ctx.build_normals(0)
        res = ctx.icp(1, 0, iters=15, stride=2, max_dist=0.2, scale_src=1.0, estimate_scale=True)
        ores = orc.icp(frames[1], orc.normals(frames[0]), iters=15, stride=2, max_dist=0.2, scale_src=1.0, estimate_scale=True)
    assert res["status"] in (0, 1) and abs(res["scale"] - 1.0) < 1e-4, res["scale"]
    assert abs(res["scale"] - ores["scale"]) < 1e-6 and np.linalg.norm(res["T"] - ores["T"]) < 1e-4
    assert np.linalg.norm(res["T"] - np.eye(4)) < 1e-3                         # sliding along the plane is not observed either


def test_a_timed_out_batched_registration_falls_back_to_the_per_iteration_kernel(tmp_path):
    """icp_batch_kernel's workgroups wait for one another inside the launch (bounded waits).  When a wait does time out the
    batch is registered again on the ICP lanes instead of failing the run; the rehearsal forces the time-out flag in the
    experiments flavour of the library (the shipped one has no such switch) and compares with the normal run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = os.path.join(root, "textureless-3d-reconstruction_amd", "libtl3d_exp.so")
    if not os.path.exists(exp):
        subprocess.check_call(["bash", os.path.join(root, "textureless-3d-reconstruction_amd", "csrc", "build.sh")],
                              env=dict(os.environ, TL3D_FLAVOUR="experiments"))
    code = r'''
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, tl3d
from helpers import make_pair, small_scene_frames
poses, frames = small_scene_frames(n=4, deg=2.0)
ctx, orc = make_pair(n_slots=4)
with ctx:
    for i, (d, c) in enumerate(frames):
        ctx.upload(i, d, None)
        ctx.build_normals(i)
    lv = [dict(iters=8, stride=4, max_dist=0.2), dict(iters=10, stride=2, max_dist=0.05)]
    res = ctx.icp_batch([(0, 1), (1, 2), (2, 3)], lv)
    st = ctx.stats()
print(json.dumps(dict(T=[r["T"].tolist() for r in res], n=[r["n_corr"] for r in res], timeouts=st["icp_batch_timeouts"],
                      fallback=st["icp_batch_fallback_pairs"])))
''' % (root, os.path.join(root, "tests"))
    outs = []
    for env in (dict(os.environ), dict(os.environ, TL3D_LIB=exp, TL3D_ICP_FORCE_TIMEOUT="1")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r.stderr))
    (a, _), (b, err) = outs
    assert a["timeouts"] == 0 and a["fallback"] == 0
    assert b["timeouts"] == 1 and b["fallback"] == 3 and "re-registering 3 pairs" in err
    assert a["n"] == b["n"] and np.abs(np.asarray(a["T"]) - np.asarray(b["T"])).max() < 1e-8


def test_reset_followed_at_once_by_integrate_on_a_large_grid():
    """tl3d_grid_reset clears 1 GiB of records and the free-space counters on the main stream; the classification kernels of
    the next frames add to those counters on a side stream.  They are ordered behind the clear (an event the prep chains wait
    for): on a 512^3 grid, where the clear takes long enough to lose that race, reset -> integrate -> download == oracle."""
    cam = dict(width=320, height=240, fx=280.0, fy=280.0, cx=159.5, cy=119.5)
    poses, frames = small_scene_frames(n=3, deg=5.0, cam=cam)
    ctx, orc = make_pair(cam=cam, dims=(512, 512, 512), voxel=0.005, centre=(0.0, -0.1, 0.0), n_slots=3, channels=tl3d.CH_TSDF)
    with ctx:
        for i, (d, c) in enumerate(frames):
            ctx.upload(i, d, None)
        for i in range(3):                                  # something to clear, counters included
            ctx.integrate(i, poses[i])
        ctx.sync()
        for rep in range(2):
            ctx.reset()
            for i in range(3):
                ctx.integrate(i, poses[i])
            g = ctx.download_grid(tl3d.CH_TSDF)
            if rep == 0:
                for i in range(3):
                    orc.tsdf_integrate(frames[i][0], poses[i][0], poses[i][1])
            assert np.array_equal(g, orc.tsdf), rep
    assert int(orc.tsdf[:, 1].sum()) > 10 ** 6


def test_sparse_grid_equals_the_dense_grid_bit_for_bit():
    """A sparse grid (records only for the bricks the data touches, handed out on first touch through a brick table; bricks that
    are wholly free space in every frame that sees them keep a 4-byte count) holds what the dense grid holds: both channels, both
    extractions and the merge of two grids equal the oracle's / the dense context's bit for bit, with a fraction of the bricks
    allocated."""
    poses, frames = small_scene_frames(n=5, deg=4.0)
    dims, voxel, centre = (96, 96, 96), 0.025, (0.0, -0.2, 0.0)
    nbr = 96 ** 3 // 512
    ctx, orc = make_pair(dims=dims, voxel=voxel, centre=centre, n_slots=5)
    origin = tuple(centre[i] - 0.5 * dims[i] * voxel for i in range(3))
    # the sparse context starts WITHOUT a grid: the frames are uploaded, the bricks they will touch are counted (tl3d_count_bricks:
    # the fusion's own classification, no records), and the grid is attached with pools of exactly that size
    sp = tl3d.FusionContext(SMALL["width"], SMALL["height"], SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"], n_slots=5, grid=None)
    with ctx, sp:
        for c in (ctx, sp):
            for i, (d, col) in enumerate(frames):
                c.upload(i, d, col)
        geom = tl3d.GridSpec(dims, origin, voxel, 4 * voxel, tl3d.CH_TSDF | tl3d.CH_CENTROID)
        nt, nc = sp.count_bricks(geom, list(range(5)), poses, centroid_subsample=1)
        assert 0 < nc < nbr and 0 < nt < nbr
        spec = tl3d.GridSpec(dims, origin, voxel, 4 * voxel, tl3d.CH_TSDF | tl3d.CH_CENTROID, pool_tsdf=nt + 8, pool_centroid=nc + 8)
        sp.attach_grid(spec)
        for c in (ctx, sp):
            for i in range(5):
                c.integrate(i, poses[i])
                c.accumulate_centroid(i, poses[i], subsample=1)
        for i, (d, col) in enumerate(frames):
            orc.tsdf_integrate(d, poses[i][0], poses[i][1])
            orc.centroid_accumulate(d, col, poses[i][0], poses[i][1], subsample=1)
        st = sp.stats()
        # the count is exactly what the fusion takes (one slot per brick, whoever touches it first and however many at once)
        assert st["pool_refused"] == 0 and st["pool_slots_tsdf"] == nt and st["pool_slots_centroid"] == nc
        assert ctx.stats()["pool_slots_tsdf"] == nbr                            # dense: every brick has records
        gt, gc = sp.download_grid(tl3d.CH_TSDF), sp.download_grid(tl3d.CH_CENTROID)
        assert np.array_equal(gc, orc.centroid)
        # TSDF: the oracle's grid, every voxel.  (Until round 3 bricks that saw nothing but free space through a footprint with
        # holes / beyond the image border got no records in a sparse grid: up to a fifth of this scene's voxels differed, and a
        # band voxel of such a brick lost the free-space observations of the OTHER views.  This scene has both kinds: the object
        # alone, most rays miss.)
        seen_free_through_holes = (orc.tsdf[:, 1] > 0) & (orc.tsdf[:, 0] == 32767 * orc.tsdf[:, 1])
        assert seen_free_through_holes.sum() > 10000
        assert np.array_equal(gt, orc.tsdf)
        gt_dense = orc.tsdf
        assert np.array_equal(ctx.download_grid(tl3d.CH_TSDF), orc.tsdf)
        assert sp.max_weight() == int(gt[:, 1].max()) and ctx.max_weight() == int(orc.tsdf[:, 1].max())
        for mode, kw in ((tl3d.EXTRACT_CENTROID, dict(min_count=1)), (tl3d.EXTRACT_CENTROID, dict(min_count=2, min_weight=2, max_abs_tsdf=0.9)),
                         (tl3d.EXTRACT_TSDF, dict(min_weight=2))):
            a, ac = sp.extract(mode, **kw)
            b, bc = ctx.extract(mode, **kw)
            assert len(a) > 100 and np.array_equal(a, b) and np.array_equal(ac, bc), (mode, kw)
        # merge: grid += dense image of another grid (records appear where the other grid has something)
        sp.add_grid(tl3d.CH_TSDF, gt)
        sp.add_grid(tl3d.CH_CENTROID, gc)
        assert np.array_equal(sp.download_grid(tl3d.CH_TSDF), 2 * gt) and np.array_equal(sp.download_grid(tl3d.CH_CENTROID), 2 * orc.centroid)
        sp.reset()
        assert sp.stats()["pool_slots_tsdf"] == 0 and not sp.download_grid(tl3d.CH_TSDF).any()
        sp.upload_grid(tl3d.CH_TSDF, gt_dense)
        assert np.array_equal(sp.download_grid(tl3d.CH_TSDF), orc.tsdf)                 # (an uploaded image is kept whole)
        with pytest.raises(tl3d.Tl3dError):
            sp.grid_ptr(tl3d.CH_TSDF)                                           # no dense layout to point at


def test_a_full_brick_pool_refuses_bricks_and_says_so():
    poses, frames = small_scene_frames(n=2, deg=4.0)
    dims, voxel, centre = (96, 96, 96), 0.025, (0.0, -0.2, 0.0)
    origin = tuple(centre[i] - 0.5 * dims[i] * voxel for i in range(3))
    spec = tl3d.GridSpec(dims, origin, voxel, 4 * voxel, tl3d.CH_TSDF | tl3d.CH_CENTROID, pool_tsdf=16, pool_centroid=8)
    with tl3d.FusionContext(SMALL["width"], SMALL["height"], SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"], n_slots=2, grid=spec) as sp:
        for i, (d, col) in enumerate(frames):
            sp.upload(i, d, col)
            sp.integrate(i, poses[i])
            sp.accumulate_centroid(i, poses[i])
        st = sp.stats()
        assert st["pool_slots_tsdf"] == 16 and st["pool_slots_centroid"] == 8 and st["pool_refused"] > 100
        xyz, _ = sp.extract(tl3d.EXTRACT_CENTROID)
        assert 0 < len(xyz) <= 8 * 512
