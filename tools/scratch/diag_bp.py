import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import numpy as np
import tl3d
import inputs as gi
import make_golden as mg
from oracle import ref_numpy as rn

for case in mg.BP_CASES:
    name, seed, h, w, kname, api, pose, tshape, scale, sub, limits, store = case
    if h < 1000:
        continue
    K = getattr(gi, kname)
    depth, color, p, (kind, sc) = mg.case_inputs(case)
    if kind == "premul":
        depth_in, s, f64 = depth * sc, 1.0, False
    else:
        depth_in, s, f64 = depth, float(sc), kind == "np64"
    for rep in range(3):
        with tl3d.FusionContext(w, h, K["fx"], K["fy"], K["cx"], K["cy"], min_depth=limits[0], max_depth=limits[1], n_slots=1) as ctx:
            ctx.upload(0, depth_in, color)
            pts, col = ctx.backproject(0, pose=p, scale=s, subsample=sub, scale_f64=f64)
            pts, col = pts.copy(), col.copy()
        op, oc = rn.backproject(depth_in, color, K["fx"], K["fy"], K["cx"], K["cy"], pose=p, scale=(np.float64(s) if f64 else s), subsample=sub,
                                min_depth=limits[0], max_depth=limits[1])
        n = len(op)
        badc = np.nonzero((col[:n] != oc).any(1))[0] if len(col) == n else None
        badp = np.nonzero((np.abs(pts[:n] - op) > 1e-5).any(1))[0] if len(pts) == n else None
        print(name, "rep", rep, "n", len(pts), n, "bad colours", None if badc is None else len(badc), "bad points", None if badp is None else len(badp))
        if badc is not None and len(badc):
            # tile of each output index
            ds = depth_in[::sub, ::sub].reshape(-1).astype(np.float64) * s
            valid = (ds > limits[0]) & (ds < limits[1])
            cum = np.cumsum(valid)
            tile_start = np.concatenate([[0], cum[2047::2048]])
            t = np.searchsorted(tile_start, badc, side="right") - 1
            print("  first bad", badc[:12], "tiles", t[:12], "offset in tile", (badc - tile_start[t])[:12])
            print("  bad index runs:", [(int(a), int(b)) for a, b in zip(badc[np.r_[True, np.diff(badc) > 1]][:10], badc[np.r_[np.diff(badc) > 1, True]][:10])])
            i = badc[0]
            print("  got", col[i - 2:i + 3].tolist(), "want", oc[i - 2:i + 3].tolist())
            print("  tile starts mod 16 (x3):", [(int(3 * tile_start[k]) % 16) for k in sorted(set(t[:12]))])
