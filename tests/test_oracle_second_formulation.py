"""CPU: the C oracle (oracle/tl3d_oracle.c -- what the HIP kernels are compared with) against a SECOND formulation of the same
contracts written in a different shape (oracle/second_numpy.py: whole-array numpy, its own fma emulation and record-order
arithmetic).  The reference holds neither TSDF nor ICP, so nothing reference-held can pin these rows;
this is the guard against a misreading shared by kernel and oracle.  TSDF grids and normal maps bit for bit, ICP normal equations
(and the Sim(3) column) to 1e-12 relative."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import second_numpy as sn
from tl3d import synth

CAM = dict(width=160, height=120, fx=140.0, fy=140.0, cx=79.5, cy=59.5)


def _frames(n, deg, scene=None, noise=0.0, cam=CAM, radius=1.0, height=0.0):
    scene = scene or synth.object_scene()
    poses = synth.orbit_poses(n, radius, deg, height=height)
    return poses, [synth.render(scene, p, cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], noise_sigma=noise,
                                seed=i, want_color=False)[0] for i, p in enumerate(poses)]


def test_fma32_is_a_single_rounding():
    rng = np.random.default_rng(0)
    a = rng.standard_normal(200000).astype(np.float32)
    b = rng.standard_normal(200000).astype(np.float32)
    c = (rng.standard_normal(200000) * 1e-3).astype(np.float32)
    # exact reference with Python integers / fractions on a sample, plus the known double-rounding trap
    from fractions import Fraction
    got = sn.fma32(a[:2000], b[:2000], c[:2000])
    for i in range(2000):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        lo = np.float32(float(exact))                             # float(Fraction) rounds once to float64, then float32: may be the trap
        cands = [np.nextafter(lo, np.float32(-np.inf)), lo, np.nextafter(lo, np.float32(np.inf))]
        best = min(cands, key=lambda x: (abs(Fraction(float(x)) - exact), int(np.float32(x).view(np.uint32)) & 1))
        assert got[i] == best, i
    # the trap itself: a*b + c lands on a float32 midpoint in float64 while the exact value lies just above it
    a1, b1 = np.float32(1.0 + 2.0 ** -12), np.float32(1.0 + 2.0 ** -12)          # product = 1 + 2^-11 + 2^-24 exactly: a midpoint of float32
    c1 = np.float32(2.0 ** -60)
    assert sn.fma32(a1, b1, c1) == np.float32(1.0 + 2.0 ** -11 + 2.0 ** -23)       # up; float64 sum then cast would round to even (down)
    assert sn.fma32(a1, b1, -c1) == np.float32(1.0 + 2.0 ** -11)
    assert np.float32(np.float64(a1) * np.float64(b1) + np.float64(c1)) == np.float32(1.0 + 2.0 ** -11)    # the two-step answer differs


def test_record_order_is_the_oracles():
    orc = c_oracle.Oracle(8, 8, 1, 1, 0, 0, dims=(24, 16, 32))
    rng = np.random.default_rng(1)
    ijk = np.stack([rng.integers(0, 24, 500), rng.integers(0, 16, 500), rng.integers(0, 32, 500)], axis=1)
    mine = sn.record_index(ijk[:, 0], ijk[:, 1], ijk[:, 2], 3, 2)
    theirs = [orc.vox_index(int(i), int(j), int(k)) for i, j, k in ijk]
    assert np.array_equal(mine, np.asarray(theirs))
    assert len(set(sn.record_index(*np.meshgrid(np.arange(24), np.arange(16), np.arange(32), indexing="ij"), 3, 2).ravel().tolist())) == 24 * 16 * 32


@pytest.mark.parametrize("case", ["orbit", "rolled_close", "scaled_limits"])
def test_tsdf_grid_of_both_formulations_is_identical(case):
    dims, voxel = (64, 64, 64), 0.03
    origin = (-0.96, -1.06, -0.96)
    kw = dict(min_depth=0.1, max_depth=50.0)
    scale = 1.0
    if case == "orbit":
        poses, frames = _frames(4, 9.0, scene=synth.object_scene(with_room=True))
    elif case == "rolled_close":
        # camera inside the grid, rolled about its axis and looking down: voxels behind the camera, at z ~ 0, outside the image
        poses, frames = _frames(3, 25.0, radius=0.7, height=-0.3)
        roll = np.array([[np.cos(0.5), -np.sin(0.5), 0], [np.sin(0.5), np.cos(0.5), 0], [0, 0, 1.0]])
        poses = [(roll @ r, roll @ t) for r, t in poses]
        frames = [synth.render(synth.object_scene(), p, want_color=False, **CAM)[0] for p in poses]
    else:
        poses, frames = _frames(3, 6.0)
        kw = dict(min_depth=0.9, max_depth=1.6)                    # limits that cut through the object
        scale = 1.25
        frames = [f / np.float32(scale) for f in frames]
    orc = c_oracle.Oracle(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], dims=dims, origin=origin, voxel_size=voxel,
                          sdf_trunc=4 * voxel, **kw)
    g = sn.Geometry(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], dims=dims, origin=origin, voxel_size=voxel,
                    sdf_trunc=4 * voxel, **kw)
    grid = np.zeros((64 ** 3, 2), np.int32)
    n_upd = 0
    for d, (R, t) in zip(frames, poses):
        orc.tsdf_integrate(d, R, t, scale=scale)
        n_upd += sn.tsdf_integrate(g, grid, d, R, t, scale=scale)
    assert n_upd > 10000
    band = np.abs(grid[:, 0]) < 32767 * np.maximum(grid[:, 1], 1)
    assert band.sum() > 1500                                       # voxels with a real signed distance, not only free space
    assert np.array_equal(grid, orc.tsdf)


def test_normal_maps_and_smoothed_depth_of_both_formulations_are_identical():
    poses, frames = _frames(2, 5.0, scene=synth.object_scene(with_room=True), noise=0.001)
    _, holes = _frames(1, 0.0)                                      # object only: most pixels invalid
    orc = c_oracle.Oracle(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"])
    g = sn.Geometry(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"])
    for d in (frames[0], holes[0]):
        assert np.array_equal(sn.normals(g, d), orc.normals(d))
        for radius in (1, 2):
            sd, nm = orc.normals_smooth(d, radius=radius)
            mine = sn.smooth_depth(g, d, radius=radius)
            assert np.array_equal(mine, sd), radius
            assert np.array_equal(sn.normals(g, mine, step=radius), nm), radius
    d = frames[0] / np.float32(1.3)
    assert np.array_equal(sn.normals(g, d, scale=1.3, depth_jump=0.02), orc.normals(d, scale=1.3, depth_jump=0.02))


def _unpack21(a21):
    A = np.zeros((6, 6))
    m = 0
    for a in range(6):
        for b in range(a, 6):
            A[a, b] = A[b, a] = a21[m]
            m += 1
    return A


@pytest.mark.parametrize("stride,scale_src", [(1, 1.0), (2, 1.0), (4, 0.8)])
def test_icp_normal_equations_of_both_formulations_agree(stride, scale_src):
    poses, frames = _frames(2, 3.0, scene=synth.object_scene(with_room=True))
    orc = c_oracle.Oracle(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"])
    g = sn.Geometry(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"])
    nm = orc.normals(frames[1])
    r_rel, t_rel = synth.relative_pose(poses[0], poses[1])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = r_rel, t_rel.ravel()
    T[:3, 3] += [0.004, -0.003, 0.002]                            # off the optimum: non-zero right-hand side
    src = frames[0] / np.float32(scale_src)
    out, cnt, nsrc = orc.icp_sums_scale(src, nm, T, stride=stride, max_dist=0.05, scale_src=scale_src)
    A, b, e, n, ns, c6, cc, bc = sn.icp_sums(g, src, nm, T, stride=stride, max_dist=0.05, scale_src=scale_src, with_scale_column=True)
    assert n == cnt and ns == nsrc and n > 1000
    A0 = _unpack21(out[:21])
    tol = 1e-12
    assert np.abs(A - A0).max() <= tol * np.abs(A0).max()
    assert np.abs(b - out[21:27]).max() <= tol * max(np.abs(out[21:27]).max(), 1e-30) + 1e-18
    assert abs(e - out[27]) <= tol * out[27]
    assert np.abs(c6 - out[29:35]).max() <= tol * np.abs(out[29:35]).max()
    assert abs(cc - out[35]) <= tol * out[35] and abs(bc - out[36]) <= tol * abs(out[36]) + 1e-18
    # and without the column the pose block is the same pass
    out29, cnt2, _ = orc.icp_sums(src, nm, T, stride=stride, max_dist=0.05, scale_src=scale_src)
    assert cnt2 == cnt and np.array_equal(out29[:28], out[:28])


def test_one_gauss_newton_step_from_the_second_formulation_matches_the_oracles_first_iteration():
    """orc_icp with one iteration from T0 == exp(x) T0 with x = -(A + lam I)^-1 b, lam = damping * trace(A) / 6 (the damping is
    relative to the mean diagonal), solved here with numpy (well-conditioned scene: no eigen-direction is truncated), composed as
    T <- [exp(omega) | tau] T."""
    poses, frames = _frames(2, 3.0, scene=synth.object_scene(with_room=True))
    orc = c_oracle.Oracle(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"])
    g = sn.Geometry(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"])
    nm = orc.normals(frames[1])
    T0 = np.eye(4)
    res = orc.icp(frames[0], nm, T_init=T0, iters=1, stride=2, max_dist=0.1, damping=1e-6, eps=0.0)
    A, b, e, n, ns = sn.icp_sums(g, frames[0], nm, T0, stride=2, max_dist=0.1)
    x = -np.linalg.solve(A + 1e-6 * (np.trace(A) / 6.0) * np.eye(6), b)
    w, tau = x[:3], x[3:]
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Rx = np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * (K @ K) if th > 0 else np.eye(3)
    T1 = np.eye(4)
    T1[:3, :3] = Rx @ T0[:3, :3]
    T1[:3, 3] = Rx @ T0[:3, 3] + tau
    assert np.abs(T1 - res["T"]).max() < 1e-9, np.abs(T1 - res["T"]).max()
