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
