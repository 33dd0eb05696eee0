"""DepthToReconstructionPipeline: the reference's orchestration (depth_to_reconstruction.py:423-703) with the
SfM pose front end replaced by on-device point-to-plane ICP and the vstack + Open3D merge replaced by on-device
voxel fusion.

Kept from the reference: load_data()'s file rules and messages, cam0 = (I, 0), the pose chain
R_c = R_rel R_prev, t_c = R_rel t_prev + t_rel (D2R:618-620), frames whose registration fails are skipped and
do not extend the pose list (D2R:598-615), subsample_factor / voxel_size semantics, the progress lines
(`Camera i: n points`, `Final reconstruction: ...`), save_reconstruction().
Not kept (out of scope): SIFT / essential-matrix poses; the sparse
triangulation that feeds estimate_scale -- depth is taken as metric (scale = config.depth_scale), which is
what D2R itself falls back to (D2R:555-558).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import time

import numpy as np

from . import _cabi as abi
from . import fileio
from .config import ReconstructionConfig
from .dense import DenseReconstructor
from .fusion import FusionContext, GridSpec


def compose(r_rel, t_rel, r_prev, t_prev):
    """D2R:619-620."""
    r_rel = np.asarray(r_rel, np.float64)
    return r_rel @ np.asarray(r_prev, np.float64), r_rel @ np.asarray(t_prev, np.float64).reshape(3, 1) + np.asarray(t_rel, np.float64).reshape(3, 1)


def plan_grid(bounds_min, bounds_max, voxel_size, grid_dim, channels=abi.CH_TSDF | abi.CH_CENTROID, trunc_voxels=4.0,
              max_voxels=None, sparse_bytes=None):
    """Grid with Open3D's voxel origin (min_bound - voxel/2) covering the bounds.

    Up to a voxel BUDGET (default grid_dim^3 voxels in total; not a cube: a 2 m x 2.4 m x 12 m corridor at 5 mm becomes
    400 x 480 x 2400 voxels) the grid is dense.  Beyond it the grid is SPARSE, as the reference's hash-map merge is
    (D2R:404-410): the same dims, records only for the bricks the data touches, pools sized by sparse_bytes (default: the dense
    budget's bytes) -- a GUESS when nothing is known about the frames; choose_layout() replaces it by a count.  Only a scene of
    more than 2^32 voxels is shrunk about its centre -- longest axis first -- and `clipped` returned True (points outside are
    dropped and counted by the accumulation kernels; the caller prints the warning)."""
    v = float(voxel_size)
    mn, mx = np.asarray(bounds_min, np.float64), np.asarray(bounds_max, np.float64)
    origin = mn - 0.5 * v
    dims = np.floor((mx - origin) / v).astype(np.int64) + 1
    dims = np.maximum(8, ((dims + 7) // 8) * 8)
    budget = int(grid_dim) ** 3 if max_voxels is None else int(max_voxels)
    budget = max(512, min(budget, 1 << 32))
    clipped = False
    want = dims.copy()
    while int(dims[0]) * int(dims[1]) * int(dims[2]) > (1 << 32):    # shave the (currently) longest axis, 8 voxels at a time
        a = int(np.argmax(dims))
        if dims[a] <= 8:
            break
        dims[a] -= 8
        clipped = True
    for a in range(3):
        if dims[a] < want[a]:
            origin[a] = 0.5 * (mn[a] + mx[a]) - 0.5 * dims[a] * v
    nvox = int(dims[0]) * int(dims[1]) * int(dims[2])
    pool_t = pool_c = 0
    if nvox > budget:
        per_vox = (8 if channels & abi.CH_TSDF else 0) + (32 if channels & abi.CH_CENTROID else 0)
        mem = int(sparse_bytes) if sparse_bytes is not None else budget * per_vox
        # surfaces: the TSDF band is ~3 bricks thick where the centroid channel holds one layer: 12 KB + 16 KB per surface brick
        unit = (3 * 4096 if channels & abi.CH_TSDF else 0) + (16384 if channels & abi.CH_CENTROID else 0)
        surf = max(4096, mem // unit)
        pool_t = int(min(nvox // 512, 3 * surf)) if channels & abi.CH_TSDF else 0
        pool_c = int(min(nvox // 512, surf)) if channels & abi.CH_CENTROID else 0
    return GridSpec(tuple(int(d) for d in dims), tuple(float(o) for o in origin), v, trunc_voxels * v, channels,
                    pool_tsdf=pool_t, pool_centroid=pool_c), clipped


DENSE_WITHOUT_ASKING = 1 << 30        # bytes: below this a dense grid is allocated (and cleared) faster than its occupancy is counted


def layout_from_counts(grid: GridSpec, bricks_tsdf: int, bricks_centroid: int) -> GridSpec:
    """Per channel: a pool of the counted bricks (+ 3 % + 2048: a slot lost to a race between two waves is not reused) when that
    is less than half of the dense channel, else the dense channel."""
    nbr = grid.nvox // 512
    pool_t = pool_c = 0
    if grid.channels & abi.CH_TSDF:
        want = int(bricks_tsdf * 1.03) + 2048
        pool_t = want if 2 * want < nbr else 0
    if grid.channels & abi.CH_CENTROID:
        want = int(bricks_centroid * 1.03) + 2048
        pool_c = want if 2 * want < nbr else 0
    return GridSpec(grid.dims, grid.origin, grid.voxel_size, grid.sdf_trunc, grid.channels, pool_tsdf=pool_t, pool_centroid=pool_c)


def choose_layout(ctx: FusionContext, grid: GridSpec, slots, poses, scales, centroid_subsample, dist=None, log=print) -> GridSpec:
    """Dense or sparse, from what the frames will really touch (tl3d_count_bricks: the fusion's own classification, no records).

    The reference's merge is a hash map over occupied voxels (D2R:404-410): memory follows the surfaces, whatever the extent.
    Here a grid is allocated (and cleared) before the first frame is fused, so its layout is a decision -- taken from a count, not
    from the extent.  Small grids are dense without asking.  With `dist` (reconstruct_sharded) every rank counts its own frames and
    all take the SUM (an upper bound of the union the merge will bring to every rank), so that all ranks choose alike."""
    nbr = grid.nvox // 512
    dense = GridSpec(grid.dims, grid.origin, grid.voxel_size, grid.sdf_trunc, grid.channels)
    if dense.device_bytes() <= DENSE_WITHOUT_ASKING:
        return dense
    nt, nc = ctx.count_bricks(grid, slots, poses, scales, centroid_subsample=centroid_subsample) if len(slots) else (0, 0)
    if dist is not None:
        from . import distributed as dd
        nt, nc = dd.allreduce_counts([nt, nc], dist)
    out = layout_from_counts(grid, nt, nc)
    log(f"  Occupancy: {nt} TSDF bricks, {nc} centroid bricks of {nbr}: "
        f"{'sparse' if out.sparse else 'dense'} volume, {out.device_bytes() / 2**30:.2f} GiB (dense: {dense.device_bytes() / 2**30:.2f} GiB)")
    return out


class ScaleTracker:
    """The reference's depth-scale bookkeeping (row f3), with the anchor source left to the caller.

    depth_to_reconstruction.py estimates, per view, the median of Z_anchor / depth[pixel] over sparse anchors
    (estimate_scale, D2R:297-326), averages the first two views (D2R:552-554) and then runs
    avg = 0.7 * avg + 0.3 * scale_i (D2R:650); views without enough anchors keep the running value (D2R:645-647).
    In the reference the anchors are triangulated SIFT matches -- the SfM front end that ICP replaces here -- so
    they are an input: any sparse metric source (fiducials, a laser range, an external SLAM map) can provide
    {frame_index: (points3d [n,3], pixels [n,2])}.  Without anchors the depth is taken as metric (D2R:555-558)."""

    def __init__(self, default: float = 1.0):
        self.avg = float(default)
        self.history = []

    def first_pair(self, s0: float, s1: float) -> float:
        self.avg = (s0 + s1) / 2
        self.history = [self.avg, self.avg]
        return self.avg

    def update(self, scale_i=None, weight: float = 0.3) -> float:
        """D2R:650 with weight = 0.3; weight = 1 takes every view's own estimate as it comes."""
        if scale_i is not None:
            self.avg = (1.0 - weight) * self.avg + weight * scale_i
        self.history.append(self.avg)
        return self.avg


def per_frame_scales(depths, anchors, default: float = 1.0, fetch=None):
    """Scale of every frame from sparse anchors, exactly as the reference sequences it.  `fetch(i)` returns frame i's
    depth map when `depths[i]` is not held on the host (streaming)."""
    from .dense import estimate_scale_d2r
    n = len(depths)
    tr = ScaleTracker(default)
    if not anchors:
        return [float(default)] * n

    def depth_of(i):
        return depths[i] if depths[i] is not None else fetch(i)

    def est(i):
        if i not in anchors:
            return None
        p3, p2 = anchors[i]
        if len(p3) < 3:
            print("  Warning: Not enough valid points for scale, using previous")
            return None
        return estimate_scale_d2r(np.asarray(p3), np.asarray(p2), depth_of(i))

    s0, s1 = est(0), est(1) if n > 1 else None
    if s0 is not None and s1 is not None:
        tr.first_pair(s0, s1)
    else:
        print("Warning: Not enough sparse points for scale estimation")
        tr.history = [tr.avg, tr.avg]
    print(f"Average scale: {tr.avg:.6f}")
    for i in range(2, n):
        tr.update(est(i))
    return tr.history[:n]


def align_grid_to_open3d(grid: GridSpec, min_bound) -> GridSpec:
    """Shift a grid by less than one voxel so its lattice coincides with Open3D's (voxel origin = min_bound - voxel/2,
    depth_to_reconstruction.py:410).  With the lattices in phase the fused centroids are the reference's centroids up
    to the accumulator quantum; out of phase, two 5 mm samplings of one surface sit 1-2 mm apart."""
    v = float(grid.voxel_size)
    o3d = np.asarray(min_bound, np.float64) - 0.5 * v
    org = np.asarray(grid.origin, np.float64)
    shift = np.mod(o3d - org, v)                       # in [0, v)
    return GridSpec(grid.dims, tuple(float(x) for x in org + shift - v), grid.voxel_size, grid.sdf_trunc, grid.channels)


class DepthToReconstructionPipeline:
    def __init__(self, config: ReconstructionConfig = None):
        self.config = config or ReconstructionConfig()
        self.dense = DenseReconstructor(self.config)
        self.images: List[np.ndarray] = []
        self.image_names: List[str] = []
        self.depths: List[np.ndarray] = []
        self.camera_poses: List[Tuple[np.ndarray, np.ndarray]] = []
        self.frame_index: List[int] = []          # which loaded frame each pose belongs to
        self.icp_log: List[dict] = []
        self.stats: dict = {}
        self.timings: dict = {}                   # wall seconds per stage of the last reconstruct()
        self.grid: Optional[GridSpec] = None      # the fusion volume of the last reconstruct()

    # ---- a2 --------------------------------------------------------------------------------------
    def load_data(self, rgb_folder: str, depth_folder: str) -> int:
        self.images, self.depths, self.image_names = fileio.load_data(rgb_folder, depth_folder)
        self._files = None
        return len(self.images)

    def load_data_streaming(self, rgb_folder: str, depth_folder: str) -> int:
        """Same pairing rules and messages as load_data(), but only the file names are kept: reconstruct() then decodes
        on worker threads into pinned buffers and uploads asynchronously (fileio.FramePrefetcher), so host RAM never
        holds the whole sequence (the reference keeps every frame in two lists, D2R:434-437)."""
        from pathlib import Path
        rgb_path, depth_path = Path(rgb_folder), Path(depth_folder)
        files = sorted(f for f in rgb_path.iterdir() if f.suffix.lower() in fileio.IMAGE_SUFFIXES)
        print(f"Found {len(files)} RGB images")
        pairs = []
        for f in files:
            d = fileio.DepthImageLoader.find_matching_depth(f.name, depth_path)
            if d is None:
                print(f"  Warning: No depth found for {f.name}")
                continue
            pairs.append((f, d))
        print(f"Loaded {len(pairs)} image-depth pairs")
        self._files = pairs
        self.image_names = [f.name for f, _ in pairs]
        if pairs:
            first = fileio.read_image_bgr(pairs[0][0])
            self._frame_shape = first.shape[:2]
        self.images, self.depths = [None] * len(pairs), [None] * len(pairs)
        return len(pairs)

    def set_frames(self, images, depths, names=None):
        """Same state load_data() leaves, from arrays already in memory."""
        self.images, self.depths = list(images), list(depths)
        self.image_names = list(names) if names is not None else [f"frame_{i:04d}" for i in range(len(self.images))]
        return len(self.images)

    # ---- poses: ICP replaces detect_and_match / compute_pose ---------------------------------------
    def _register(self, ctx: FusionContext, scales, init_poses=None):
        """Frame-to-frame registration.  Consecutive pairs are independent: they go to the device in batches, every pair of
        a batch through all its levels and iterations inside one launch.  A failed pair drops its frame (reference rule,
        D2R:598-615): the following pair is then re-registered against the last kept frame."""
        cfg = self.config
        n = len(self.depths)
        if not isinstance(scales, (list, tuple)):
            scales = [float(scales)] * n
        # coarse-to-fine: each level is (iterations, pixel stride, correspondence gate); every level starts from the
        # previous level's pose.  A wide first gate takes frame steps of 0.5 m / 30 degrees that the 5 cm gate alone loses
        # from 15 cm / 8 degrees on (tools/icp_basin.py); a level that has converged stops after one iteration.
        levels = [tuple(l) for l in cfg.icp_coarse] + [(cfg.icp_iters, cfg.icp_stride, cfg.icp_max_dist)]
        common = dict(damping=cfg.icp_damping, eig_rel=cfg.icp_eig_rel, eps=cfg.icp_eps)

        def level_kw(lv):
            return dict(iters=int(lv[0]), stride=int(lv[1]), max_dist=float(lv[2]), **common)

        level_list = [level_kw(lv) for lv in levels]

        def blocking(src, cur, T0):
            return ctx.icp_batch([(src, cur)], level_list, T_init=[T0], scales=[scales[src]])[0]
        poses = [(np.eye(3), np.zeros((3, 1)))]
        index = [0]
        prev = 0
        ctx.build_normals_many(list(range(n)), scales)
        T_guess = np.eye(4)

        def prior(a, b_):
            if init_poses is None:
                return T_guess
            r0, t0 = init_poses[a]
            r1, t1 = init_poses[b_]
            rr = np.asarray(r1) @ np.asarray(r0).T
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = rr, (np.asarray(t1).reshape(3) - rr @ np.asarray(t0).reshape(3))
            return T

        i = 1
        while i < n:
            # every pair of a batch runs all its levels inside ONE launch (tl3d_icp_batch_*); batches only exist so that the
            # constant-velocity prior of the next one can come from the last pose found (a short first batch gets one early)
            batch = list(range(i, min(n, i + (16 if i == 1 else 128))))
            srcs = [prev if cur == batch[0] else cur - 1 for cur in batch]
            T0s = [prior(src, cur) for src, cur in zip(srcs, batch)]
            results = ctx.icp_batch(list(zip(srcs, batch)), level_list, T_init=T0s, scales=[scales[src] for src in srcs])
            for lane, cur in enumerate(batch):
                print(f"\nProcessing image {cur}...")
                res = results[lane]
                src = prev if cur == batch[0] else cur - 1
                if src != prev:                       # the frame this run started from was dropped: redo against `prev`
                    res = blocking(prev, cur, prior(prev, cur))
                self.icp_log.append(dict(frame=cur, against=prev, **{k: res[k] for k in ("fitness", "rmse", "n_corr", "iters_run", "status")}))
                if res["status"] == 2 or res["n_corr"] < 8:
                    print(f"  Skipping - registration failed (correspondences: {res['n_corr']})")
                    continue
                T = res["T"]
                r_c, t_c = compose(T[:3, :3], T[:3, 3], *poses[-1])
                poses.append((r_c, t_c))
                index.append(cur)
                print(f"  ICP: fitness {res['fitness']:.3f}, rmse {res['rmse'] * 1e3:.2f} mm, {res['iters_run']} iterations")
                T_guess = T                  # constant-velocity prior for the next batch
                prev = cur
            i = batch[-1] + 1
        return poses, index

    def _sim3_levels(self):
        cfg = self.config
        wide = [(15, max(2, int(cfg.icp_stride) * 2), 1.0)] + [tuple(l) for l in cfg.icp_coarse] + [(cfg.icp_iters, cfg.icp_stride, cfg.icp_max_dist)]
        common = dict(damping=cfg.icp_damping, eig_rel=cfg.icp_eig_rel, eps=cfg.icp_eps, estimate_scale=True)
        return [dict(iters=int(lv[0]), stride=int(lv[1]), max_dist=float(lv[2]), **common) for lv in wide][-abi.ICP_MAX_LEVELS:]

    def _sim3_chain(self, ctx: FusionContext, frames, slot_of, state, init_poses=None, weight: float = 0.3):
        """One stretch of the Sim(3) registration chain: the views `frames` (ascending global indices), one after the other,
        each against the last kept view.  `state` = dict(prev, prev_scale, avg, T_guess) is where the chain stands -- the last kept
        view (resident in slot_of, its normals built with prev_scale), the running scale and the constant-velocity prior -- and
        comes back advanced, so that the next stretch (the next rank, reconstruct_sharded) continues exactly where this one ends.
        Returns {cur: row}; row: T (prev -> cur, the (R_rel, t_rel) the reference chains), against, ok, scale_raw (the
        registration's own estimate), scale (the running value the view is fused with), statistics."""
        level_list = self._sim3_levels()
        rows = {}
        prev, avg, T_guess = int(state["prev"]), float(state["avg"]), np.asarray(state["T_guess"], np.float64).reshape(4, 4)
        for cur in frames:
            print(f"\nProcessing image {cur}...")
            T0 = T_guess
            if init_poses is not None:
                r0, t0 = init_poses[prev]
                r1, t1 = init_poses[cur]
                rr = np.asarray(r0) @ np.asarray(r1).T       # cur -> prev
                T0 = np.eye(4)
                T0[:3, :3], T0[:3, 3] = rr, (np.asarray(t0).reshape(3) - rr @ np.asarray(t1).reshape(3))
            res = ctx.icp_batch([(slot_of[cur], slot_of[prev])], level_list, T_init=[T0], scales=[avg])[0]
            ok = not (res["status"] == 2 or res["n_corr"] < 8 or not np.isfinite(res["scale"]) or not (1e-3 < res["scale"] < 1e3))
            row = dict(T=np.eye(4), against=prev, ok=ok, scale_raw=float(res["scale"]) if np.isfinite(res["scale"]) else 0.0, scale=avg,
                       **{k: res[k] for k in ("fitness", "rmse", "n_corr", "iters_run", "status")})
            rows[cur] = row
            if not ok:
                print(f"  Skipping - registration failed (correspondences: {res['n_corr']})")
                continue
            avg = (1.0 - weight) * avg + weight * res["scale"]               # D2R:650 with weight = 0.3
            T = res["T"]
            Ti = np.eye(4)
            Ti[:3, :3] = T[:3, :3].T
            Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]              # prev -> cur = (R_rel, t_rel)
            row["T"], row["scale"] = Ti, avg
            print(f"  ICP: fitness {res['fitness']:.3f}, rmse {res['rmse'] * 1e3:.2f} mm, {res['iters_run']} iterations, scale {res['scale']:.6f} -> {avg:.6f}")
            ctx.build_normals(slot_of[cur], scale=avg)
            T_guess = T
            prev = cur
            state.update(prev=prev, prev_scale=avg)
        state.update(avg=avg, T_guess=T_guess)
        return rows

    def _register_with_scale(self, ctx: FusionContext, scale0: float, init_poses=None, weight: float = 0.3):
        """Registration of RELATIVE depth maps (row f3): every view's metric scale is the 7th unknown of its registration
        against the previous kept view (Sim(3) point-to-plane ICP, tl3d_icp_params.estimate_scale), which replaces the
        reference's median of Z_triangulated / depth over SIFT points (D2R:297-326, DER:659-697).  View 0 fixes the gauge
        (scale0 = config.depth_scale, or the anchors' estimate); the running scale follows the reference's rule
        avg = (1 - w) avg + w scale_i (D2R:650, w = 0.3; config.scale_update_weight = 1 trusts every view's own estimate).
        The source of a run is the NEW view (unknown scale), the target the previous one (scale known, normals built with
        it), so the run returns cur -> prev; its inverse is the (R_rel, t_rel) the reference chains (D2R:618-620).
        Sequential by nature: a view's target needs that view's scale.  (Both views of a run read window-averaged depth when
        config.icp_smooth_radius > 0: the target through its normal map, the source through the averaged map its own normals were
        built from -- a view is a source BEFORE its scale is known, so its map is built with the running value first and rebuilt
        with its own scale once that is known; the averaged depth itself does not depend on the scale.)"""
        n = len(self.depths)
        ctx.build_normals_many(list(range(n)), [float(scale0)] * n)       # every view's averaged depth (what a SOURCE reads); normals are rebuilt per view below
        state = dict(prev=0, prev_scale=float(scale0), avg=float(scale0), T_guess=np.eye(4))
        rows = self._sim3_chain(ctx, list(range(1, n)), {g: g for g in range(n)}, state, init_poses, weight)
        return self._sim3_finish(rows, n, scale0)

    def _sim3_finish(self, rows, n, scale0):
        """(poses, kept frame indices, per-frame scales) from the rows of every view 1 .. n-1: cam0 = (I, 0), then
        R = R_rel R_prev, t = R_rel t_prev + t_rel over the kept views (D2R:618-620)."""
        poses, index, scales = [(np.eye(3), np.zeros((3, 1)))], [0], [float(scale0)] * n
        for cur in range(1, n):
            row = rows[cur]
            self.icp_log.append(dict(frame=cur, against=row["against"], scale=row["scale_raw"],
                                     **{k: row[k] for k in ("fitness", "rmse", "n_corr", "iters_run", "status")}))
            if not row["ok"]:
                continue
            scales[cur] = row["scale"]
            r_c, t_c = compose(row["T"][:3, :3], row["T"][:3, 3], *poses[-1])
            poses.append((r_c, t_c))
            index.append(cur)
        return poses, index, scales

    # ---- reconstruct ---------------------------------------------------------------------------------
    def reconstruct(self, grid: Optional[GridSpec] = None, init_poses=None, poses=None, anchors=None, estimate_scale: bool = False):
        """(points, colors, camera_poses) like D2R:479-671.

        grid: fix the fusion volume (else planned from the data with Open3D's voxel origin).
        init_poses: optional per-frame pose priors for ICP; poses: skip registration and fuse with these poses.
        anchors: {frame: (points3d, pixels)} sparse metric anchors -> per-frame depth scale by the reference's rule
        (ScaleTracker); default: config.depth_scale for every frame (metric depth).
        estimate_scale: relative depth with no anchors -- every view's scale is estimated by its registration (Sim(3) ICP,
        _register_with_scale); view 0 keeps config.depth_scale (or its anchors' estimate).
        """
        if len(self.images) < 2:
            print("Need at least 2 images")
            return None, None, None
        cfg = self.config
        print("\n" + "=" * 70)
        print("DEPTH-ENHANCED RECONSTRUCTION PIPELINE (MI355X: ICP + voxel fusion)")
        print("=" * 70)
        streaming = getattr(self, "_files", None) is not None
        h, w = self._frame_shape if streaming else self.depths[0].shape
        n = len(self.depths)
        scale = float(cfg.depth_scale)
        if not anchors:
            print(f"Using depth scale = {scale} (depth assumed metric, as D2R:555-558)")
        # one context: every frame is uploaded once and stays resident in HBM through registration, bounding and fusion
        # (288 GB holds ~19 000 frames of 1080x1920 depth+colour); the grid is attached once the scene bounds are known
        clock, marks = time.perf_counter, [("start", time.perf_counter())]
        ctx = FusionContext(w, h, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.min_depth, cfg.max_depth, n_slots=n, grid=None,
                            device=cfg.device)
        ctx.set_normal_smoothing(int(getattr(cfg, "icp_smooth_radius", 0)))
        try:
            if streaming:
                pre = fileio.FramePrefetcher(ctx, [f for f, _ in self._files], [d for _, d in self._files])
                try:
                    for _i, _slot in pre:
                        pass
                finally:
                    # where a file-fed run's time goes: decode (summed over the worker threads) against the wall time of the stage
                    self.decode_stats = dict(workers=pre.workers, decode_thread_seconds=round(pre.decode_s, 3),
                                             decode_ms_per_frame_and_thread=round(1e3 * pre.decode_s / max(1, n), 2))
                    pre.close()
            else:
                for i in range(n):
                    if self.depths[i].shape != (h, w):
                        raise ValueError(f"frame {i} is {self.depths[i].shape}, expected {(h, w)}")
                    ctx.upload(i, self.depths[i], self.images[i])
            # depth maps may be GPU tensors (depth estimated in this process): the scale rule then reads them back per frame
            host_depths = [d if isinstance(d, np.ndarray) else None for d in self.depths]
            self.scales = per_frame_scales(host_depths, anchors, default=scale, fetch=ctx.download_depth)
            marks.append(("upload", clock()))
            if poses is not None:
                self.camera_poses, self.frame_index = list(poses), list(range(len(poses)))
            else:
                if estimate_scale:
                    print("\n--- Step 1: Register frames (Sim(3) point-to-plane ICP: pose and depth scale, frame to frame) ---")
                    self.camera_poses, self.frame_index, self.scales = self._register_with_scale(
                        ctx, self.scales[0], init_poses, weight=float(getattr(cfg, "scale_update_weight", 0.3)))
                else:
                    print("\n--- Step 1: Register frames (point-to-plane ICP, frame to frame) ---")
                    self.camera_poses, self.frame_index = self._register(ctx, self.scales, init_poses)
            marks.append(("register", clock()))
            if len(self.camera_poses) < 2:
                print("Pose estimation failed")
                return None, None, None
            if grid is None:
                print("\n--- Step 2: Bound the scene ---")
                # extent of the union of the frames' clouds, on the device
                mn, mx = ctx.frames_bounds(self.frame_index, self.camera_poses, [self.scales[fi] for fi in self.frame_index],
                                           subsample=cfg.subsample_factor)
                if not np.all(np.isfinite(mn)):
                    print("Reconstruction failed")
                    return None, None, None
                grid, clipped = plan_grid(mn, mx, cfg.voxel_size, cfg.grid_dim, trunc_voxels=cfg.sdf_trunc_voxels)
                if clipped:
                    print(f"  Warning: scene extent {np.round(mx - mn, 3)} m at {cfg.voxel_size} m voxels exceeds the budget of "
                          f"{cfg.grid_dim}^3 voxels; the grid {grid.dims} is centred on the scene and points outside it are dropped "
                          "(raise --grid or --voxel-size)")
                # dense or sparse: from the bricks these frames will really touch, not from the extent
                grid = choose_layout(ctx, grid, self.frame_index, self.camera_poses, [self.scales[fi] for fi in self.frame_index],
                                     cfg.subsample_factor)
            print(f"  Grid {grid.dims} @ {grid.voxel_size * 1e3:g} mm, origin {np.round(grid.origin, 4)}")
            self.grid = grid
            ctx.attach_grid(grid)
            marks.append(("bound_and_allocate", clock()))
            print("\n--- Step 3: Fuse depth frames (TSDF + voxel centroids) ---")
            ctx.fuse_frames(self.frame_index, self.camera_poses, [self.scales[fi] for fi in self.frame_index],
                            centroid_subsample=cfg.subsample_factor)           # TSDF + centroids of every kept frame, in order
            for fi in self.frame_index:
                print(f"Camera {fi}: fused")
            st = ctx.stats()
            marks.append(("fuse", clock()))
            print("\n--- Step 4: Extract and clean point cloud ---")
            xyz, rgb = ctx.extract(abi.EXTRACT_CENTROID, min_count=1, min_weight=cfg.tsdf_min_weight,
                                   max_abs_tsdf=cfg.tsdf_max_abs)
            n_vox = len(xyz)
            if len(xyz) > 0 and cfg.outlier_filter:         # D2R:413-415; DER's merge has none (DER:615-645)
                keep = ctx.statistical_outlier(xyz, cfg.outlier_nb_neighbors, cfg.outlier_std_ratio, cell_size=2.0 * grid.voxel_size)
                xyz, rgb = xyz[keep], rgb[keep]
            self.stats = dict(points_accumulated=st["centroid_points"], points_dropped=st["centroid_dropped"],
                              voxels=n_vox, after_outlier_filter=len(xyz), sparse=bool(grid.sparse),
                              bricks_tsdf=st["pool_slots_tsdf"], bricks_centroid=st["pool_slots_centroid"], pool_refused=st["pool_refused"])
            if grid.sparse:
                print(f"  Sparse volume: {st['pool_slots_tsdf']} TSDF bricks and {st['pool_slots_centroid']} centroid bricks hold records "
                      f"(of {grid.nvox // 512}); {grid.device_bytes() / 2**30:.2f} GiB")
                if st["pool_refused"]:
                    print(f"  Warning: {st['pool_refused']} bricks found the record pool full and are missing from the result "
                          "(raise --grid, the memory budget of the volume)")
            marks.append(("extract_and_filter", clock()))
            # wall time of each stage of this call (host clock; stages end at a point where the host has the stage's result)
            self.timings = {name + "_s": round(t1 - t0, 4) for (name, t1), (_, t0) in zip(marks[1:], marks[:-1])}
        finally:
            ctx.close()
        print(f"\nFinal reconstruction: {len(xyz)} points, {len(self.camera_poses)} cameras")
        return xyz.astype(np.float64), rgb, self.camera_poses

    # ---- multi-GPU: frames shard across ranks, one exchange step at merge time -------------
    def _register_pairs(self, ctx: FusionContext, pairs, slot_of, scales, init_poses=None, T_guess=None):
        """Independent registrations (src, cur) -> result dict, every pair through the coarse-to-fine levels inside one
        launch per batch.  slot_of maps a global frame index to its resident slot."""
        cfg = self.config
        levels = [tuple(l) for l in cfg.icp_coarse] + [(cfg.icp_iters, cfg.icp_stride, cfg.icp_max_dist)]
        common = dict(damping=cfg.icp_damping, eig_rel=cfg.icp_eig_rel, eps=cfg.icp_eps)
        out = {}

        def prior(a, b_):
            if init_poses is None:
                return np.eye(4) if T_guess is None else T_guess
            r0, t0 = init_poses[a]
            r1, t1 = init_poses[b_]
            rr = np.asarray(r1) @ np.asarray(r0).T
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = rr, (np.asarray(t1).reshape(3) - rr @ np.asarray(t0).reshape(3))
            return T

        level_list = [dict(iters=int(lv[0]), stride=int(lv[1]), max_dist=float(lv[2]), **common) for lv in levels]
        for i0 in range(0, len(pairs), 256):
            batch = pairs[i0:i0 + 256]
            results = ctx.icp_batch([(slot_of[a], slot_of[b_]) for a, b_ in batch], level_list, T_init=[prior(a, b_) for a, b_ in batch],
                                    scales=[scales[a] for a, _ in batch])
            for (a, b_), res in zip(batch, results):
                out[b_] = dict(res, against=a)
        return out

    def reconstruct_sharded(self, dist, grid: Optional[GridSpec] = None, init_poses=None, poses=None, anchors=None, estimate_scale: bool = False):
        """reconstruct() over the ranks of an initialised torch.distributed group, one process per GPU:

          1. rank r uploads its contiguous frame range [lo, hi) plus the one-frame halo lo - 1 and registers every pair whose
             later frame it owns (the pair that crosses a shard boundary belongs to the later rank), on its ICP lanes;
          2. the relative transforms are exchanged (a few hundred bytes) and every rank chains the same global poses with
             the same fp64 products as D2R:618-620; a frame whose registration failed is dropped and its successor is
             re-registered against the last kept frame by the successor's owner (the reference's skip rule, D2R:598-615);
          3. scene bounds: per-rank min / max, MIN / MAX all-reduce -> every rank plans the same grid;
          4. rank r fuses its own frames into its private grid;
          5. ONE sum all-reduce of the integer grids (RCCL over xGMI with the nccl backend) -> every rank holds the merged
             grid, bit-identical to a single-GPU run GIVEN THE SAME POSES (pairs are registered independently here, from
             identity or init_poses; reconstruct() seeds later batches with a constant-velocity prior, so the two can converge
             to poses that differ in the last bits); rank 0 extracts, filters and returns the cloud (other ranks return
             (None, None, poses)).
        estimate_scale (relative depth, no anchors: reconstruct()'s Sim(3) registration): a view's target needs that view's scale,
        so the chain is sequential -- the ranks take turns, each continues it over its own views from the state the previous rank
        hands over (last kept view and its scale, running scale, motion prior: 19 doubles), and the poses, the scales and the
        cloud are those of the single-process run bit for bit; fusion and merge stay parallel.
        Every frame must be loaded on every rank's host (load_data); only the rank's own range goes to its GPU."""
        from . import distributed as dd
        world, rank = dist.get_world_size(), dist.get_rank()
        if len(self.images) < 2:
            print("Need at least 2 images")
            return None, None, None
        if getattr(self, "_files", None) is not None:
            raise ValueError("reconstruct_sharded needs load_data(): frames decoded on the host of every rank")
        cfg = self.config
        n = len(self.depths)
        h, w = self.depths[0].shape
        lo, hi = dd.shard_range(n, world, rank)
        first = max(0, lo - 1) if hi > lo else lo                  # halo frame in front of the range
        resident = list(range(first, hi))
        slot_of = {g: k for k, g in enumerate(resident)}
        say = print if rank == 0 else (lambda *a, **k: None)
        say("\n" + "=" * 70)
        say(f"DEPTH-ENHANCED RECONSTRUCTION PIPELINE (MI355X x{world}: ICP + voxel fusion, frames sharded by rank)")
        say("=" * 70)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO() if rank else None) if rank else contextlib.nullcontext():
            self.scales = per_frame_scales(self.depths, anchors, default=float(cfg.depth_scale))
        spare = len(resident)                                       # one more slot for out-of-range sources of repairs
        ctx = FusionContext(w, h, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.min_depth, cfg.max_depth, n_slots=len(resident) + 1, grid=None,
                            device=cfg.device)
        ctx.set_normal_smoothing(int(getattr(cfg, "icp_smooth_radius", 0)))
        try:
            for g in resident:
                ctx.upload(slot_of[g], self.depths[g], self.images[g])
            if poses is not None:
                self.camera_poses, self.frame_index = list(poses), list(range(len(poses)))
            elif estimate_scale:
                say("\n--- Step 1: Register frames (Sim(3) point-to-plane ICP: pose and depth scale; the ranks take turns along the chain) ---")
                scale0 = float(self.scales[0])
                weight = float(getattr(cfg, "scale_update_weight", 0.3))
                for g in resident:
                    ctx.build_normals(slot_of[g], scale=scale0)           # every view's averaged depth (what a SOURCE reads)
                state = dict(prev=0, prev_scale=scale0, avg=scale0, T_guess=np.eye(4))
                rows = {}
                for turn in range(world):
                    mine_now = list(range(max(lo, 1), hi)) if turn == rank else []
                    if mine_now:
                        slots = dict(slot_of)
                        if state["prev"] not in slots:                    # the last kept view lives on another rank's GPU: bring it here
                            ctx.upload(spare, self.depths[state["prev"]], self.images[state["prev"]])
                            slots[state["prev"]] = spare
                        ctx.build_normals(slots[state["prev"]], scale=state["prev_scale"])     # the target's normals, with ITS scale
                        with contextlib.redirect_stdout(io.StringIO()) if rank else contextlib.nullcontext():
                            rows = self._sim3_chain(ctx, mine_now, slots, state, init_poses, weight)
                    state = dd.handover_sim3_state(state, turn, dist)
                table = dd.exchange_sim3_rows(rows, n, dist)
                self.camera_poses, self.frame_index, self.scales = self._sim3_finish(table, n, scale0)
            else:
                say("\n--- Step 1: Register frames (point-to-plane ICP, frame to frame, pairs sharded by rank) ---")
                for g in resident:
                    ctx.build_normals(slot_of[g], scale=self.scales[g])
                own = [(a, b_) for a, b_ in dd.pairs_for_rank(n, world, rank)]
                local = self._register_pairs(ctx, own, slot_of, self.scales, init_poses)
                table = dd.exchange_registrations(local, n, dist)             # every rank: {cur: T, against, ok, statistics}
                # the reference's skip rule: a failed frame is dropped and its successor is re-registered against the last
                # kept frame -- by the successor's owner, one exchange per repair (failures are rare); every rank takes the
                # same decisions from the same table
                for _ in range(n):
                    kept, redo = dd.resolve_chain(table, n)
                    if redo is None:
                        break
                    want, cur = redo
                    fixed = {}
                    if lo <= cur < hi:
                        src_slot = slot_of
                        if want not in slot_of:                               # the source frame lives on another rank's GPU: bring it here
                            ctx.upload(spare, self.depths[want], self.images[want])
                            src_slot = {**slot_of, want: spare}
                        fixed = self._register_pairs(ctx, [(want, cur)], src_slot, self.scales, init_poses)
                    dd.exchange_registrations(fixed, n, dist, into=table)
                self.camera_poses, self.frame_index, self.icp_log = dd.chain_from_table(table, n)
                for e in self.icp_log:
                    say(f"\nProcessing image {e['frame']}...")
                    if e["dropped"]:
                        say(f"  Skipping - registration failed (correspondences: {e['n_corr']})")
                    else:
                        say(f"  ICP: fitness {e['fitness']:.3f}, rmse {e['rmse'] * 1e3:.2f} mm, {e['iters_run']} iterations")
            if len(self.camera_poses) < 2:
                say("Pose estimation failed")
                return None, None, None
            pose_of = dict(zip(self.frame_index, self.camera_poses))
            mine = [g for g in range(lo, hi) if g in pose_of]
            if grid is None:
                say("\n--- Step 2: Bound the scene ---")
                mn, mx = np.full(3, np.inf), np.full(3, -np.inf)
                if mine:
                    mn, mx = ctx.frames_bounds([slot_of[g] for g in mine], [pose_of[g] for g in mine], [self.scales[g] for g in mine],
                                               subsample=cfg.subsample_factor)
                mn, mx = dd.allreduce_bounds(mn, mx, dist)
                if not np.all(np.isfinite(mn)):
                    say("Reconstruction failed")
                    return None, None, None
                grid, clipped = plan_grid(mn, mx, cfg.voxel_size, cfg.grid_dim, trunc_voxels=cfg.sdf_trunc_voxels)
                if clipped:
                    say(f"  Warning: scene extent {np.round(mx - mn, 3)} m at {cfg.voxel_size} m voxels exceeds the budget of "
                        f"{cfg.grid_dim}^3 voxels; the grid {grid.dims} is centred on the scene and points outside it are dropped "
                        "(raise --grid or --voxel-size)")
                grid = choose_layout(ctx, grid, [slot_of[g] for g in mine], [pose_of[g] for g in mine], [self.scales[g] for g in mine],
                                     cfg.subsample_factor, dist=dist if world > 1 else None, log=say)
            say(f"  Grid {grid.dims} @ {grid.voxel_size * 1e3:g} mm, origin {np.round(grid.origin, 4)}")
            self.grid = grid
            ctx.attach_grid(grid)
            say(f"\n--- Step 3: Fuse depth frames (TSDF + voxel centroids), {len(mine)} of {len(pose_of)} on this rank ---")
            for g in mine:
                if grid.channels & abi.CH_TSDF:
                    ctx.integrate(slot_of[g], pose_of[g], scale=self.scales[g])
                ctx.accumulate_centroid(slot_of[g], pose_of[g], scale=self.scales[g], subsample=cfg.subsample_factor)
            say("\n--- Step 4: Merge the per-GPU grids (integer sum all-reduce) ---")
            info = dd.merge_context_grids(ctx, dist)
            if info:
                say(f"  {info['bricks_sent']} of {info['bricks_total']} bricks travelled ({info['bytes'] / 1e6:.1f} MB per rank and direction)")
            st = ctx.stats()
            tot = dd.allreduce_counts([st["centroid_points"], st["centroid_dropped"]], dist)
            xyz = rgb = None
            if rank == 0:
                say("\n--- Step 5: Extract and clean point cloud ---")
                xyz, rgb = ctx.extract(abi.EXTRACT_CENTROID, min_count=1, min_weight=cfg.tsdf_min_weight, max_abs_tsdf=cfg.tsdf_max_abs)
                n_vox = len(xyz)
                if len(xyz) > 0 and cfg.outlier_filter:
                    keep = ctx.statistical_outlier(xyz, cfg.outlier_nb_neighbors, cfg.outlier_std_ratio, cell_size=2.0 * grid.voxel_size)
                    xyz, rgb = xyz[keep], rgb[keep]
                self.stats = dict(points_accumulated=tot[0], points_dropped=tot[1], voxels=n_vox, after_outlier_filter=len(xyz))
                say(f"\nFinal reconstruction: {len(xyz)} points, {len(self.camera_poses)} cameras")
                xyz = xyz.astype(np.float64)
        finally:
            ctx.close()
        return xyz, rgb, self.camera_poses

    def export_frame_clouds(self, out_dir, subsample: int = 1, min_depth=None, max_depth=None):
        """Per-frame camera-space clouds like depth_processor.py writes them (DP:371-422 back-projection, DP:923-934 file):
        `<out_dir>/pointclouds/<image stem>.ply`, one per loaded frame.  Returns the number of files written."""
        from pathlib import Path
        cfg = self.config
        out = Path(out_dir) / "pointclouds"
        out.mkdir(parents=True, exist_ok=True)
        if not self.depths:
            return 0
        h, w = self.depths[0].shape
        n_written = 0
        with FusionContext(w, h, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.min_depth if min_depth is None else min_depth,
                           cfg.max_depth if max_depth is None else max_depth, n_slots=1, grid=None, device=cfg.device) as ctx:
            for name, d, img in zip(self.image_names, self.depths, self.images):
                ctx.upload(0, d, img)
                pts, col = ctx.backproject(0, pose=None, scale=float(cfg.depth_scale), subsample=int(subsample))
                if len(pts) == 0:
                    continue                                         # DP:929-931: nothing to save
                fileio.write_ply_binary(out / f"{Path(name).stem}.ply", pts.astype(np.float64), col)
                n_written += 1
        return n_written

    def save_reconstruction(self, points, colors, output_path: str, ascii: bool = False):
        fileio.save_reconstruction(points, colors, output_path, ascii=ascii)
