"""CPU: the parts of the oracle the reference cannot pin (TSDF, ICP, voxel centroid, outlier filter) against
analytic known answers and against each other.  These define what the GPU path is compared with."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import ref_numpy as rn
from tl3d import synth

CAM = dict(width=160, height=120, fx=140.0, fy=140.0, cx=79.5, cy=59.5)


def _oracle(dims=(64, 64, 64), voxel=0.02, centre=(0.0, 0.0, 0.0), **kw):
    origin = tuple(centre[i] - 0.5 * dims[i] * voxel for i in range(3))
    return c_oracle.Oracle(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], dims=dims,
                           origin=origin, voxel_size=voxel, sdf_trunc=4 * voxel, **kw)


def test_centroid_accumulators_equal_open3d_voxel_down_sample():
    scene = synth.object_scene()
    poses = synth.orbit_poses(3, 1.0, 6.0)
    orc = _oracle(dims=(128, 128, 128), voxel=0.02)
    clouds = []
    for p in poses:
        d, c = synth.render(scene, p, **CAM)
        orc.centroid_accumulate(d, c, p[0], p[1], subsample=1)
        clouds.append(rn.backproject(d, c, CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], pose=p))
    # the C back-projection inside the accumulator and the numpy restatement see the same points
    cp, cc = orc.backproject(*synth.render(scene, poses[0], **CAM), poses[0][0], poses[0][1])
    assert np.array_equal(cp, clouds[0][0]) and np.array_equal(cc, clouds[0][1])
    xyz, rgb = orc.extract(0)
    pts = np.vstack([c[0] for c in clouds])
    col = np.vstack([c[1] for c in clouds])
    # Open3D semantics with the grid's origin standing in for min_bound - voxel/2
    o = np.array(orc.cfg.origin[:])
    idx = np.floor((pts.astype(np.float64) - o) / 0.02).astype(np.int64)
    assert orc.n_drop.value == 0 and orc.n_acc.value == len(pts)
    uniq, inv, cnt = np.unique(idx, axis=0, return_inverse=True, return_counts=True)
    assert len(xyz) == len(uniq)
    psum = np.zeros((len(uniq), 3)); np.add.at(psum, inv.reshape(-1), pts.astype(np.float64))
    csum = np.zeros((len(uniq), 3)); np.add.at(csum, inv.reshape(-1), col.astype(np.float64))
    mean = psum / cnt[:, None]
    # extraction order is brick-major; compare as sets through the voxel index
    gidx = np.floor((xyz.astype(np.float64) - o) / 0.02).astype(np.int64)
    og, orf = np.lexsort(gidx.T[::-1]), np.lexsort(uniq.T[::-1])
    assert np.array_equal(gidx[og], uniq[orf])
    assert np.abs(xyz[og] - mean[orf]).max() < 0.02 / 4096 + 2e-7            # offset quantum voxel/4096 + f32 output
    assert np.array_equal(rgb[og], np.floor(csum[orf] / cnt[orf][:, None]).astype(np.uint8))


def test_tsdf_zero_crossing_lies_on_the_analytic_plane():
    n = np.array([0.0, 0.0, -1.0])
    scene = synth.Scene(planes=[(tuple(n), float(n @ np.array([0.0, 0.0, 0.30])))])      # plane z = 0.30 in the world
    orc = _oracle(dims=(64, 64, 64), voxel=0.02, centre=(0.0, 0.0, 0.3))
    for p in synth.orbit_poses(5, 1.0, 4.0, target=(0.0, 0.0, 0.3)):
        d, _ = synth.render(scene, p, want_color=False, **CAM)
        orc.tsdf_integrate(d, p[0], p[1])
    w = orc.tsdf[:, 1]
    assert w.max() == 5 and (w > 0).sum() > 20000
    mean = orc.tsdf[:, 0] / np.maximum(w, 1) / 32767.0
    assert mean.max() <= 1.0 and mean[w > 0].min() >= -1.0
    xyz, rgb = orc.extract(1, min_weight=3, use_centroid=False)
    assert len(xyz) > 500 and np.all(rgb == 128)
    assert np.abs(xyz[:, 2] - 0.30).max() < 0.004                  # well inside one voxel (20 mm)
    assert abs(np.mean(xyz[:, 2]) - 0.30) < 1e-3
    # free space in front of the plane is exactly +1, and nothing more than trunc behind it was touched
    i, j = 32, 32
    col = np.array([orc.vox_index(i, j, k) for k in range(64)])
    zc = -0.34 + (np.arange(64) + 0.5) * 0.02
    assert np.all(orc.tsdf[col[zc < 0.15], 0] == 32767 * orc.tsdf[col[zc < 0.15], 1])
    assert np.all(orc.tsdf[col[zc > 0.30 + 0.08 + 0.03], 1] == 0)


def test_icp_recovers_analytic_relative_pose():
    scene = synth.object_scene()
    poses = synth.orbit_poses(2, 1.0, 2.0)
    f0 = synth.render(scene, poses[0], **CAM)
    f1 = synth.render(scene, poses[1], **CAM)
    orc = _oracle(dims=(8, 8, 8))
    nm = orc.normals(f1[0])
    assert (nm[..., 3] > 0).mean() > 0.6
    lens = np.linalg.norm(nm[..., :3][nm[..., 3] > 0], axis=1)
    assert np.abs(lens - 1).max() < 1e-5
    res = orc.icp(f0[0], nm, iters=25, stride=1, max_dist=0.1)
    r_rel, t_rel = synth.relative_pose(poses[0], poses[1])
    T_true = np.eye(4); T_true[:3, :3] = r_rel; T_true[:3, 3] = t_rel.ravel()
    assert res["fitness"] > 0.8 and res["rmse"] < 2e-3
    assert np.linalg.norm(res["T"] - T_true) < 4e-3
    assert np.linalg.norm(res["T"] - T_true) < 0.1 * np.linalg.norm(np.eye(4) - T_true)     # 10x closer than the start
    # a perfect initial guess is a fixed point
    res2 = orc.icp(f0[0], nm, T_init=res["T"], iters=3, stride=1, max_dist=0.1)
    assert np.linalg.norm(res2["T"] - res["T"]) < 1e-4


def test_icp_leaves_unobservable_motion_at_the_prior():
    """A fronto-parallel plane observes z, rx, ry only: x/y translation and in-plane rotation must not drift (H3)."""
    n = np.array([0.0, 0.0, -1.0])
    scene = synth.Scene(planes=[(tuple(n), float(n @ np.array([0.0, 0.0, 1.0])))])
    p0 = synth.look_at((0.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    p1 = synth.look_at((0.0, 0.0, 0.02), (0.0, 0.0, 1.0))           # 2 cm towards the plane
    d0, _ = synth.render(scene, p0, want_color=False, **CAM)
    d1, _ = synth.render(scene, p1, want_color=False, **CAM)
    orc = _oracle(dims=(8, 8, 8))
    res = orc.icp(d0, orc.normals(d1), iters=10, stride=2, max_dist=0.1)
    T = res["T"]
    assert abs(T[2, 3] + 0.02) < 1e-4                                # observed: the approach along z
    assert np.abs(T[:2, 3]).max() < 1e-6 and abs(T[0, 1]) < 1e-6      # unobserved: stays at the identity prior
    assert res["status"] in (0, 1)


def test_statistical_outlier_restatement():
    rng = np.random.default_rng(0)
    pts = np.vstack([rng.normal(0, 0.01, (500, 3)), [[1.0, 1.0, 1.0]], [[-1.0, 0.5, 0.2]]])
    keep = rn.statistical_outlier_open3d(pts, 20, 2.0)
    assert not keep[-1] and not keep[-2] and keep[:500].mean() > 0.9
    assert rn.statistical_outlier_open3d(np.zeros((0, 3))).shape == (0,)
    # duplicates only: every mean distance is 0 -> nothing is valid (Open3D keeps mean > 0 only)
    assert not rn.statistical_outlier_open3d(np.ones((30, 3)), 20, 2.0).any()


def test_chamfer_metric():
    a = np.array([[0, 0, 0], [1, 0, 0.0]])
    assert rn.chamfer_mean(a, a) == 0
    assert abs(rn.chamfer_mean(a, a + [0, 0.001, 0]) - 0.001) < 1e-12
