"""CPU: the oracle against vectors captured from the reference itself (tests/golden/make_golden.py).
Pins rows a3-a9 of SURVEY.md section 8a.  Nothing here touches a GPU."""
import os

import numpy as np
import pytest

import inputs as gi
import make_golden as mg
from helpers_cpu import ulp_diff
from oracle import c_oracle
from oracle import ref_numpy as rn


@pytest.fixture(scope="module")
def bp(golden_dir):
    return np.load(os.path.join(golden_dir, "backproject.npz"))


@pytest.fixture(scope="module")
def misc(golden_dir):
    return np.load(os.path.join(golden_dir, "misc.npz"))


def _inputs(case):
    name, seed, h, w, kname, api, pose, tshape, scale, sub, limits, store = case
    depth, color, p, (kind, sc) = mg.case_inputs(case)
    if kind == "premul":
        return depth * sc, color, p, 1.0
    return depth, color, p, sc


@pytest.mark.parametrize("case", mg.BP_CASES, ids=[c[0] for c in mg.BP_CASES])
def test_numpy_oracle_bit_exact_vs_reference(case, bp):
    name, seed, h, w, kname, api, pose, tshape, scale, sub, limits, store = case
    K = getattr(gi, kname)
    d_raw, c_raw, _, _ = mg.case_inputs(case)
    assert gi.digest(d_raw, c_raw) == str(bp[f"{name}/in_digest"]), "seeded inputs drifted from the golden run"
    depth, color, p, sc = _inputs(case)
    pts, col = rn.backproject(depth, color, K["fx"], K["fy"], K["cx"], K["cy"], pose=p, scale=sc, subsample=sub,
                              min_depth=limits[0], max_depth=limits[1])
    assert len(pts) == int(bp[f"{name}/n"])
    if store == "full":
        assert np.array_equal(col, bp[f"{name}/colors"])
        assert np.array_equal(pts.view(np.int32), bp[f"{name}/points"].view(np.int32))          # bit for bit
    else:
        st = int(bp[f"{name}/stride"])
        assert np.array_equal(pts[::st].view(np.int32), bp[f"{name}/points_strided"].view(np.int32))
        assert np.array_equal(pts[:16], bp[f"{name}/points_head"]) and np.array_equal(pts[-16:], bp[f"{name}/points_tail"])
        assert np.array_equal(col[::st], bp[f"{name}/colors_strided"])
        assert np.array_equal(col.astype(np.int64).sum(0), bp[f"{name}/colors_sum"])
        assert np.allclose(pts.astype(np.float64).sum(0), bp[f"{name}/points_sum"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", [c for c in mg.BP_CASES if c[2] <= 481], ids=[c[0] for c in mg.BP_CASES if c[2] <= 481])
def test_c_oracle_backprojection_matches_reference(case, bp):
    """The C restatement (what the GPU grids are compared with) against the same goldens: <= 1 ulp of f32."""
    name, seed, h, w, kname, api, pose, tshape, scale, sub, limits, store = case
    K = getattr(gi, kname)
    depth, color, p, sc = _inputs(case)
    kind = scale[0]
    orc = c_oracle.Oracle(w, h, K["fx"], K["fy"], K["cx"], K["cy"], limits[0], limits[1])
    R, t = (None, None) if p is None else p
    pts, col = orc.backproject(depth, color, R, t, scale=float(sc), flags=1 if kind == "np64" else 0, subsample=sub)
    assert len(pts) == int(bp[f"{name}/n"])
    if store == "full":
        assert np.array_equal(col, bp[f"{name}/colors"])
        ud = ulp_diff(pts, bp[f"{name}/points"])
        assert ud.max() <= 1 and (ud > 0).mean() < 1e-3
    else:
        st = int(bp[f"{name}/stride"])
        assert ulp_diff(pts[::st], bp[f"{name}/points_strided"]).max() <= 1
        assert np.array_equal(col[::st], bp[f"{name}/colors_strided"])


def test_threshold_rows(misc):
    row, row2, col = misc["thr/row"], misc["thr/row2"], misc["thr/col"]
    p, c = rn.backproject(row, col, 10.0, 10.0, 4.0, 0.0)
    assert np.array_equal(p, misc["thr/d2r_points"]) and np.array_equal(c, misc["thr/d2r_colors"])
    assert p[:, 2].tolist() == [np.float32(0.10000001), np.float32(49.999996), 1.0]      # SURVEY section 8a probe
    p, c = rn.backproject(row, col, 10.0, 10.0, 4.0, 0.0, scale=np.float64(1.0))
    assert np.array_equal(p, misc["thr/d2r64_points"])
    p, c = rn.backproject(row2, col, 10.0, 10.0, 4.0, 0.0, min_depth=0.1, max_depth=100.0)
    assert np.array_equal(p, misc["thr/der_points"]) and np.array_equal(c, misc["thr/der_colors"])


def test_projection_factors(misc):
    xf, yf = rn.projection_factors(6, 9, **gi.K_S)
    assert xf.dtype == np.float64 and np.array_equal(xf, misc["factors/xf"]) and np.array_equal(yf, misc["factors/yf"])


def test_estimate_scale_cases(misc):
    for name, (p3, p2, dm) in mg.scale_cases().items():
        for variant in ("d2r", "der"):
            got = rn.estimate_scale(p3, p2, dm, variant)
            assert np.float64(got) == misc[f"scale/{name}_{variant}"], (name, variant)
    assert misc["scale/too_few_d2r"] == 1.0 and misc["scale/four_der"] == 1.0 and misc["scale/four_d2r"] != 1.0


def test_merge_vstack_and_empty(misc):
    p1 = np.array([[0.1, 0.2, 0.3], [1, 2, 3]], np.float32)
    c1 = np.array([[1, 2, 3], [4, 5, 6]], np.uint8)
    p2 = np.array([[-1.5, 0.25, 7.125]], np.float32)
    c2 = np.array([[255, 0, 128]], np.uint8)
    e = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8))
    mp, mc = rn.merge_vstack([(p1, c1), e, (p2, c2)])
    assert np.array_equal(mp, misc["merge/points"]) and np.array_equal(mc, misc["merge/colors"])
    ep, ec = rn.merge_vstack([e])
    assert list(ep.shape) == misc["merge/empty_shape0"].tolist() and str(ep.dtype) == str(misc["merge/empty_dtype"])


def test_ascii_ply_text(misc, golden_dir):
    text = rn.ply_ascii_text(misc["ply/points"], misc["ply/colors"])
    assert text == str(misc["ply/d2r_text"]) == str(misc["ply/der_text"])
    with open(os.path.join(golden_dir, "ascii_d2r.ply")) as f:
        assert f.read() == text
    assert str(misc["ply/empty_stdout"]) == "No points to save\n"


def test_reference_defaults(misc):
    assert misc["defaults/d2r"].tolist() == [1719.0, 1719.0, 540.0, 960.0, 0.1, 50.0, 0.005, 2.0]
