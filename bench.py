#!/usr/bin/env python3
"""bench.py -- depth frames/sec fused into the TSDF (1080x1920), 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of --frames-per-step synthetic frames per GPU (512 = one turn of a
512-frame orbit, frames 0.7 degrees apart): each frame, already resident in HBM, is integrated into that GPU's 512^3 TSDF
grid with ONE foreign call per 64 frames (tl3d_fuse_frames); the library updates 32 frames per launch
(roofline.frames_per_sweep = 32; the same frames at one per launch are measured beside it:
roofline.single_frame_per_sweep).  With N > 1 the frames shard across ranks (weak scaling: per-GPU batch fixed) and the
per-GPU grids are summed with the product's merge inside the timed region.  Rank 0 prints ONE JSON line.

roofline: algorithmic bytes per launch are COUNTED by the kernel (8 B x (records read + records written),
SURVEY.md section 8d) plus the depth bytes of the launch's frames; time is hipEvent time on the launching stream over
the timed region.  roofline.bound / achieved / peak / frac price the algorithmic bytes against HBM, as the measurement contract
defines them; roofline.limited_by names what really limits the kernel (vector-instruction issue: DESIGN.md 7.5).  roofline.traffic and the
two sub-objects scattered_reads / vector_alu come from the builder's own rocprofv3 --pmc passes (profiles/pmc_traffic.json,
keyed by workload): counters need passes of their own, they are NOT counters of this run, and the line says so.
cpu_baseline: the C oracle (oracle/tl3d_oracle.c, OpenMP) on the host cores, bounded sample.  rows: the other rows of the
path, each timed over 5 repetitions (median; [min, max] in rows.spread).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames-per-step", type=int, default=512)
    ap.add_argument("--resident-frames", type=int, default=512,
                    help="distinct synthetic frames kept in HBM per GPU (512 = one frame per 0.7 degrees of the orbit, the sequence SURVEY.md 8d names)")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--voxel", type=float, default=0.005)
    ap.add_argument("--width", type=int, default=1080)
    ap.add_argument("--height", type=int, default=1920)
    ap.add_argument("--depth-format", choices=["f32", "u16"], default="f32",
                    help="frames as float32 metres (BASELINE.md's headline input) or as 16-bit millimetres (the reference "
                         "loader's PNG depth, D2R:85-90; the TSDF kernels then gather from the 16-bit image)")
    ap.add_argument("--centroid", action="store_true", help="also accumulate the voxel-centroid channel each frame")
    ap.add_argument("--icp", action="store_true", help="also run frame-to-frame ICP each frame (poses still analytic)")
    ap.add_argument("--icp-iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the one-frame-per-launch measurement beside the roofline (profiling runs)")
    ap.add_argument("--no-rows", action="store_true", help="skip the per-row rates of the rest of the path (N = 1 only)")
    ap.add_argument("--cpu-merge-frames", type=int, default=6, help="frames of the numpy back-project -> vstack -> voxel centroid -> SOR sample")
    ap.add_argument("--force-dist", action="store_true", help="init torch.distributed (RCCL) even with one rank: exercises the merge path")
    ap.add_argument("--cpu-frames", type=int, default=4, help="distinct frames the CPU baseline cycles over")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline sample")
    return ap.parse_args()


def load_launcher():
    """textureless-3d-reconstruction_amd/launch.py by path: the parent of `--gpus N` imports nothing that could load the HIP runtime"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tl3d_launch", os.path.join(ROOT, "textureless-3d-reconstruction_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def spawn_ranks(n):
    """--gpus N outside torchrun: start N fresh copies of this command line, one per GPU, BEFORE this process has made any GPU
    call (the parent only watches its ranks and passes rank 0's line through).  A rank that dies -- by a signal too -- ends the
    group with a non-zero status instead of leaving the others in a collective; the wait is bounded (TL3D_RANK_TIMEOUT_S)."""
    return load_launcher().spawn_ranks(os.path.abspath(__file__), sys.argv[1:], n)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # stdout carries exactly one line, the JSON; libraries that print banners with printf (RCCL at communicator creation)
    # are sent to stderr at the file-descriptor level for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # before anything can initialise the HIP runtime (it reads the variable once; `import tl3d` sets the same default)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import tl3d
    import torch
    from tl3d import synth
    from tl3d import _cabi as abi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # TL3D_SHARE_DEVICE=1 + TL3D_DIST_BACKEND=gloo: several ranks on ONE GPU (rehearsal of the N > 1 path on a one-GPU box)
    if os.environ.get("TL3D_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = os.environ.get("TL3D_DIST_BACKEND", "nccl")
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                      # single-process rehearsal of the merge path
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    H, W = args.height, args.width
    sx = W / 1080.0
    cam = dict(width=W, height=H, fx=1719.0 * sx, fy=1719.0 * sx, cx=W / 2.0, cy=H / 2.0)
    n = args.grid
    F = args.frames_per_step
    n_res = min(args.resident_frames, F * max(1, args.steps))
    total_frames_rank = F * args.steps
    # every rank flies the SAME kind of turn: resident_frames frames, 360 / resident_frames degrees apart (0.70 at 512), so the
    # per-GPU step at N = 8 is the step the N = 1 line times; rank r's turn is shifted by r / N of one frame spacing, i.e. the
    # ranks hold frames r, r + N, r + 2N, ... of one N x finer sequence
    deg = 360.0 / max(1, args.resident_frames)
    scene = synth.object_scene(with_room=True)
    poses = synth.orbit_poses(n_res, 1.0, deg, start_deg=rank * deg / world)
    want_rows = world == 1 and not args.no_rows
    channels = tl3d.CH_TSDF | (tl3d.CH_CENTROID if (args.centroid or want_rows) else 0)
    spec = tl3d.GridSpec.cube(n, args.voxel, centre=(0.0, -0.1, 0.0), channels=channels)
    # one explicit stream shared by torch (frame synthesis, RCCL) and the library, so every hand-over is ordered
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = tl3d.FusionContext(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], min_depth=0.1, max_depth=50.0,
                             n_slots=n_res, grid=spec, device=local_rank, stream=stream.cuda_stream)

    t_gen = time.perf_counter()
    host_keep = []
    n_invalid = 0
    for i, p in enumerate(poses):
        d, c = synth.render(scene, p, W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], xp=torch, device=dev)
        d, c = d.contiguous(), c.contiguous()
        n_invalid += int(((d <= 0.1) | (d >= 50.0)).sum().item())
        if args.depth_format == "u16":            # what a 16-bit PNG holds: round(metres * 1000), converted back by / 1000
            mm = torch.clamp(torch.round(d * 1000.0), 0, 65535).to(torch.int32).to(torch.uint16).contiguous()
            ctx.upload(i, mm, c)
            d_host = (mm.cpu().numpy().astype(np.float32) / np.float32(1000.0)) if (rank == 0 and i < args.cpu_frames) else None
            del mm
        else:
            ctx.upload(i, d, c)
            d_host = d.cpu().numpy() if (rank == 0 and i < args.cpu_frames) else None
        stream.synchronize()                      # the tensors are freed right after: finish the copy first
        if d_host is not None:
            host_keep.append((d_host, p))
        del d, c
    torch.cuda.synchronize(dev)
    t_gen = time.perf_counter() - t_gen
    invalid_frac = n_invalid / float(max(1, n_res) * H * W)
    if args.icp or want_rows:
        for i in range(n_res):
            ctx.build_normals(i)

    T_rel = []
    if args.icp or want_rows:
        for k in range(n_res):                       # analytic inter-frame motion as the ICP prior
            r_rel, t_rel = synth.relative_pose(poses[(k - 1) % n_res], poses[k])
            T0 = np.eye(4)
            T0[:3, :3], T0[:3, 3] = r_rel, t_rel.ravel()
            T_rel.append(T0)

    icp_level = [dict(iters=args.icp_iters, stride=4, max_dist=0.05)]

    # the sequence as the arrays tl3d_fuse_frames takes (packed once: the timed loop makes one foreign call per 64 frames)
    R_all = np.ascontiguousarray(np.stack([np.asarray(p[0], np.float64).reshape(9) for p in poses]))
    t_all = np.ascontiguousarray(np.stack([np.asarray(p[1], np.float64).reshape(3) for p in poses]))
    ones_all = np.ones(n_res)

    def step(s, icp=args.icp, centroid=args.centroid, frames=F):
        # registration of group g+1 (one batched launch: every pair through all its iterations) runs while group g is fused
        G = 64
        groups = [[(s * frames + j) % n_res for j in range(j0, min(frames, j0 + G))] for j0 in range(0, frames, G)]

        def enqueue(gi):
            ks = groups[gi]
            ctx.icp_batch_enqueue([((k - 1) % n_res, k) for k in ks], icp_level, T_init=[T_rel[k] for k in ks])

        if icp:
            enqueue(0)
        for gi, ks in enumerate(groups):
            if icp:
                ctx.icp_batch_collect()
                if gi + 1 < len(groups):
                    enqueue(gi + 1)
            idx = np.asarray(ks, np.int32)
            ctx.fuse_frames_packed(idx, np.ascontiguousarray(R_all[idx]), np.ascontiguousarray(t_all[idx]), ones_all[:len(ks)],
                                   centroid_subsample=2 if centroid else 0)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    merge_ms, merge_info = [], []

    def merge():
        # the product's merge: int32-headroom check + RCCL sum all-reduce on the library's own grid memory (zero-copy)
        if dist is not None:
            from tl3d.distributed import allreduce_context_grids, merge_context_grids
            torch.cuda.synchronize(dev)                         # (not ctx.sync(): that would fold the free-space counts, which the merge sends as counts)
            tm = time.perf_counter()
            if world == 1:                                      # --force-dist: the RCCL path with one rank
                info = allreduce_context_grids(ctx, dist)
            else:
                info = merge_context_grids(ctx, dist)
            merge_info.append(info)
            ctx.sync()
            merge_ms.append(1e3 * (time.perf_counter() - tm))

    # int32 headroom of the TSDF sums: a voxel may hold TL3D_TSDF_MAX_WEIGHT (65 536) observations, the merged grid included.
    # A job longer than that is a sequence of scans: merge, hand the grid on (here: drop it), start the next scan.  All of it
    # stays inside the timed region.
    scan_frames = max(F, abi.TSDF_MAX_WEIGHT // world)
    for s in range(args.warmup):
        step(s)
    if dist is not None and args.warmup > 0:
        merge()                                                 # warm the communicator
        merge_ms.clear()
        merge_info.clear()
    ctx.reset()
    barrier()
    ctx.event_record(0)
    t0 = time.perf_counter()
    in_scan, n_merges = 0, 0
    for s in range(args.steps):
        if in_scan + F > scan_frames:
            merge()
            ctx.reset()
            in_scan, n_merges = 0, n_merges + 1
        step(s)
        in_scan += F
    ctx.event_record(1)
    merge()
    n_merges += 1
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    dev_ms = ctx.event_elapsed_ms()
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the dominant kernel (tsdf_integrate), rank 0 ----------------------------------------
    depth_bpp = 2.0 if args.depth_format == "u16" else 4.0

    def measure_roofline(c_, steps_, nres_=None, F_=None):
        """Counted algorithmic bytes per launch (a counting pass over the resident frames) and the kernel's own duration
        (hipEvent pairs around every batch of back-to-back launches, on the launching stream) over `steps_` steps.  A launch
        may update two consecutive frames (frames_per_sweep = frames / launches): the bytes are what the launch moves."""
        nres_ = n_res if nres_ is None else nres_
        F_ = F if F_ is None else F_
        c_.reset_stats()
        c_.set_profile(count_records=True, time_kernels=False)
        for j in range(nres_):
            c_.integrate(j, poses[j])
        st = c_.stats()
        nl = max(1, st["tsdf_launches"])
        fps_ = nres_ / nl
        rec_per_launch = (st["tsdf_records_read"] + st["tsdf_records_written"]) / nl
        counted_free = st["tsdf_bricks_free_counted"] / nl
        c_.reset_stats()
        c_.reset()
        c_.set_profile(count_records=False, time_kernels=True)
        for s_ in range(steps_):
            for j in range(F_):
                c_.integrate((s_ * F_ + j) % nres_, poses[(s_ * F_ + j) % nres_])
        st2 = c_.stats()
        c_.set_profile(False, False)
        c_.reset()
        k_ms = st2["tsdf_kernel_ms"] / max(1, st2["tsdf_kernel_timed"])
        # voxel records read + written (8 B each), one 4-byte counter read + written per free-space brick that was counted
        # instead of streamed, and the depth frame(s) of the launch read once
        bytes_launch = 8.0 * rec_per_launch + 8.0 * counted_free + fps_ * depth_bpp * H * W
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        return {"achieved": round(achieved, 1), "frac": round(achieved / 8000.0, 4), "bytes_per_launch": int(bytes_launch),
                "records_per_launch": int(rec_per_launch), "bricks_visited_per_launch": int(st["tsdf_bricks_visited"] / nl),
                "free_space_bricks_per_launch": int(st["tsdf_bricks_free"] / nl), "free_space_bricks_counted_per_launch": int(counted_free),
                "ms_per_launch": round(k_ms, 4), "frames_per_sweep": round(fps_, 3), "us_per_frame": round(1e3 * k_ms / fps_, 2)}

    roof = None
    if rank == 0:
        launches = total_frames_rank
        m = measure_roofline(ctx, args.steps)
        region_ms = dev_ms / launches
        counters_on = m["free_space_bricks_counted_per_launch"] > 0
        traffic = None
        pmc_extra = {}
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pj):
            try:
                with open(pj) as f:
                    tj = json.load(f)
                # a PMC measurement only speaks for the kernel and input it was taken on
                if (tj.get("grid") == n and tj.get("width") == W and tj.get("height") == H and tj.get("depth_format", "f32") == args.depth_format
                        and bool(tj.get("free_space_counters", False)) == counters_on
                        and int(tj.get("frames_per_sweep", 1)) == int(round(m["frames_per_sweep"]))):
                    traffic = tj.get("hbm_bytes_per_launch")
                    pmc_extra = {k: tj.get(k) for k in ("l2_read_requests_per_launch", "scattered_read_roof_requests_per_s", "valu_instructions_per_launch")}
            except Exception:
                traffic = None
        paired = m["frames_per_sweep"] > 1.01
        roof = {"bound": "hbm",                               # the roof achieved / peak / frac are priced against (the measurement contract's field) ...
                "limited_by": "vector-instruction issue, not HBM: the update kernel and the batch as a whole (prep chain + update) run at "
                              "~85 % of the chip's measured issue rate of 4.4 cycles per wave instruction per SIMD (DESIGN.md 7.5, "
                              "tools/ubench_issue.hip); see vector_alu",     # ... and what really limits the kernel
                "kernel": "tsdf_update_pairs_kernel", "achieved": m["achieved"], "peak": 8000.0,
                "unit": "GB/s", "frac": m["frac"], "traffic": traffic,
                "traffic_source": ("profiles/pmc_traffic.json: the builder's own rocprofv3 --pmc passes over this workload (separate runs; "
                                   "not counters of this run)") if traffic is not None else None,
                "bytes_per_launch": m["bytes_per_launch"], "records_per_launch": m["records_per_launch"],
                "dense_sweep_bytes": int(16.0 * n ** 3 + 4.0 * H * W), "bricks_visited_per_launch": m["bricks_visited_per_launch"],
                "free_space_bricks_per_launch": m["free_space_bricks_per_launch"],
                "free_space_bricks_counted_per_launch": m["free_space_bricks_counted_per_launch"],
                "ms_per_launch": m["ms_per_launch"], "us_per_frame": m["us_per_frame"], "ms_per_frame_all_kernels": round(region_ms, 4),
                "launches": int(round(launches / m["frames_per_sweep"])), "frames_per_sweep": m["frames_per_sweep"]}
        if pmc_extra.get("l2_read_requests_per_launch") and m["ms_per_launch"] > 0:
            # what the kernel is really bound by, from the committed PMC passes of this workload and this run's timing:
            # 64-B read requests that leave the L2 per second against the chip's measured rate for scattered 4-B reads, and the share
            # of the launch the vector ALU needs at full issue rate (4 cycles per instruction, 4 SIMDs on each of 256 CUs, 2.4 GHz)
            rate = pmc_extra["l2_read_requests_per_launch"] / (m["ms_per_launch"] * 1e-3)
            roof["scattered_reads"] = {"source": "profiles/pmc_traffic.json (builder PMC passes) + this run's timing",
                                       "l2_read_requests_per_launch": pmc_extra["l2_read_requests_per_launch"],
                                       "achieved_G_per_s": round(rate * 1e-9, 2), "roof_G_per_s": round(pmc_extra["scattered_read_roof_requests_per_s"] * 1e-9, 2),
                                       "frac": round(rate / pmc_extra["scattered_read_roof_requests_per_s"], 4)}
            if pmc_extra.get("valu_instructions_per_launch"):
                valu_ms = pmc_extra["valu_instructions_per_launch"] * 4.0 / (1024 * 2.4e9) * 1e3
                roof["vector_alu"] = {"source": "profiles/pmc_traffic.json (builder PMC passes) + this run's timing",
                                      "instructions_per_launch": pmc_extra["valu_instructions_per_launch"], "ms_at_full_issue_rate": round(valu_ms, 4),
                                      "frac_of_launch": round(valu_ms / m["ms_per_launch"], 4)}
        if paired and not args.no_single:
            # the same frames, one per launch (F = 1: what the fraction was quoted on before), same context, same buffers
            ctx.set_tsdf_pairing(False)
            roof["single_frame_per_sweep"] = measure_roofline(ctx, max(1, min(args.steps, 4)))
            ctx.set_tsdf_pairing(True)

    # ---- CPU baseline: the oracle on the host cores, bounded sample, rank 0 at N=1 only --------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        c_oracle.Oracle.set_threads(c_oracle.usable_cpus())       # the box shows 256 CPUs and grants a quota of 16
        orc = c_oracle.Oracle(W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"], 0.1, 50.0, dims=spec.dims,
                              origin=spec.origin, voxel_size=spec.voxel_size, sdf_trunc=spec.sdf_trunc)
        orc.tsdf_integrate(host_keep[0][0], host_keep[0][1][0], host_keep[0][1][1])     # page in the 1 GiB grid
        tc0 = time.perf_counter()
        done = 0
        while time.perf_counter() - tc0 < args.cpu_seconds:
            d, p = host_keep[done % len(host_keep)]
            orc.tsdf_integrate(d, p[0], p[1])
            done += 1
        tc = time.perf_counter() - tc0
        cpu = {"value": round(done / tc, 3), "unit": "frames/s", "cores": orc.threads, "kind": "port",
               "sample": f"{done} integrations cycling over {len(host_keep)} of the same {W}x{H} frames into the same {n}^3 "
                         f"grid, oracle/tl3d_oracle.c orc_tsdf_integrate (OpenMP over z-slabs, every voxel visited), "
                         f"{tc:.1f} s wall"}
        # the reference's own per-frame work (numpy back-projection, its default subsample=2) for context
        from oracle import ref_numpy
        tb = time.perf_counter()
        ref_numpy.backproject(host_keep[0][0], np.zeros((H, W, 3), np.uint8), cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                              pose=host_keep[0][1], subsample=2)
        cpu["reference_numpy_backproject_s2_fps"] = round(1.0 / (time.perf_counter() - tb), 2)
        del orc
        # the reference's own CPU path for these frames, restated in numpy (oracle/ref_numpy.py, bit-checked against the
        # reference's back-projection; Open3D's voxel_down_sample / remove_statistical_outlier restated): per frame
        # depth_to_pointcloud at the D2R stride 2 (D2R:328-384), then ONE merge_pointclouds = vstack -> voxel centroid at 5 mm ->
        # 20-NN / 2 sigma outlier filter (D2R:386-420).  Bounded sample: a few frames, single process.
        nm = max(2, min(args.cpu_merge_frames, len(host_keep) * 4))
        tm0 = time.perf_counter()
        clouds = []
        for i in range(nm):
            d, p = host_keep[i % len(host_keep)]
            clouds.append(ref_numpy.backproject(d, np.zeros((H, W, 3), np.uint8), cam["fx"], cam["fy"], cam["cx"], cam["cy"], pose=p, subsample=2))
        tm1 = time.perf_counter()
        mp_, _ = ref_numpy.merge_open3d(clouds, args.voxel, sor=True)
        tm2 = time.perf_counter()
        cpu["restated_reference_path"] = {
            "kind": "restated reference", "value": round(nm / (tm2 - tm0), 3), "unit": "frames/s", "cores": 1,
            "sample": f"{nm} of the same {W}x{H} frames: numpy depth_to_pointcloud (stride 2) {tm1 - tm0:.1f} s + vstack -> voxel centroid "
                      f"@ {args.voxel * 1e3:g} mm -> 20-NN outlier filter {tm2 - tm1:.1f} s, {len(mp_)} points out (oracle/ref_numpy.py, "
                      "scipy cKDTree)"}

    # ---- the rest of the path north_star names, measured in this same run (N = 1): registration in the loop, the
    # voxel-centroid channel, back-projection to a point list, extraction, outlier filter ---------------------------
    rows = None
    if want_rows:
        def timed(fn, reps):
            fn(0)
            ctx.sync()
            tq = time.perf_counter()
            for k in range(reps):
                fn(k + 1)
            ctx.sync()
            return (time.perf_counter() - tq) / reps

        REPS = 5                                                   # every row: median of 5 samples, [min, max] in rows.spread
        spread = {}

        def sampled(key, fn, reps, to_value, digits=1):
            vals = sorted(to_value(timed(fn, reps)) for _ in range(REPS))
            rows[key] = round(vals[REPS // 2], digits)
            spread[key] = [round(vals[0], digits), round(vals[-1], digits)]

        rows = {}
        nf = 256
        ctx.reset()
        sampled("tsdf_plus_centroid_s2_fps", lambda k: step(k, icp=False, centroid=True, frames=nf), 2, lambda t_: nf / t_)
        ctx.reset()
        sampled("icp_in_loop_tsdf_fps", lambda k: step(k, icp=True, centroid=False, frames=nf), 2, lambda t_: nf / t_)
        ctx.reset()
        sampled("icp_in_loop_tsdf_plus_centroid_fps", lambda k: step(k, icp=True, centroid=True, frames=nf), 2, lambda t_: nf / t_)
        # the centroid channel alone (stride 2, the reference's default D2R:65): counted bytes per frame = the samples read
        # ((4 + 3) B each) + 2 x 32 B per record update (counted by the kernel: one per distinct voxel and 32 x 32 tile of samples)
        ctx.reset()
        ctx.reset_stats()
        sampled("centroid_s2_us_per_frame", lambda k: [ctx.accumulate_centroid((k * nf + j) % n_res, poses[(k * nf + j) % n_res], subsample=2) for j in range(nf)],
                2, lambda t_: 1e6 * t_ / nf, digits=2)
        stc = ctx.stats()
        n_acc = max(1, stc["centroid_launches"])
        upd_per_frame = stc["centroid_record_updates"] / max(1.0, (REPS * 3.0 * nf))
        cen_bytes = (4 + 3) * (H // 2) * (W // 2) + 64.0 * upd_per_frame
        rows["centroid"] = {"subsample": 2, "us_per_frame": rows["centroid_s2_us_per_frame"], "record_updates_per_frame": int(upd_per_frame),
                            "bytes_per_frame": int(cen_bytes), "achieved_GBps": round(cen_bytes / (rows["centroid_s2_us_per_frame"] * 1e-6) / 1e9, 1),
                            "frac_of_8TBps": round(cen_bytes / (rows["centroid_s2_us_per_frame"] * 1e-6) / 8e12, 4), "frames_per_launch": 32,
                            "launches": int(n_acc)}
        ctx.reset()
        rows["icp"] = {"iters": args.icp_iters, "stride": 4, "pairs_per_launch": 64, "prior": "analytic inter-frame motion",
                       "workgroups_per_pair": int(min(32, -(-((W + 3) // 4) * ((H + 3) // 4) // 8192))), "pair_slots_per_workgroup": 2,
                       "algorithmic_bytes_per_pair_iteration": 20 * ((W + 3) // 4) * ((H + 3) // 4)}
        sampled("icp_single_chain_us_per_iteration",
                lambda k: ctx.icp((k - 1) % n_res, k % n_res, T_init=T_rel[k % n_res], iters=args.icp_iters, stride=4, max_dist=0.05, eps=0.0), 24,
                lambda t_: 1e6 * t_ / (args.icp_iters + 1), digits=2)
        # batched registration alone: every resident pair in ONE launch (all iterations inside the kernel), and the pipeline's
        # two-level coarse-to-fine schedule (10 x stride 4 @ 20 cm, then 15 x stride 2 @ 5 cm, stop at a 1e-7 update)
        all_pairs = [((k - 1) % n_res, k) for k in range(n_res)]
        fixed = [dict(iters=args.icp_iters, stride=4, max_dist=0.05, eps=0.0)]
        sampled("icp_batch_pairs_per_s", lambda k: ctx.icp_batch(all_pairs, fixed, T_init=T_rel), 2, lambda t_: n_res / t_)
        rows["icp_batch_us_per_pair_iteration"] = round(1e6 / rows["icp_batch_pairs_per_s"] / (args.icp_iters + 1), 3)
        # 20 B per sampled pixel (4 B source depth + 16 B target normal and depth), SURVEY.md section 8d
        rows["icp"]["batch_achieved_GBps"] = round(rows["icp"]["algorithmic_bytes_per_pair_iteration"] / rows["icp_batch_us_per_pair_iteration"] / 1e3, 1)
        rows["icp"]["batch_frac_of_8TBps"] = round(rows["icp"]["batch_achieved_GBps"] / 8000.0, 4)
        two = [dict(iters=10, stride=4, max_dist=0.2, eps=1e-7), dict(iters=15, stride=2, max_dist=0.05, eps=1e-7)]
        sampled("icp_batch_two_level_from_identity_pairs_per_s", lambda k: ctx.icp_batch(all_pairs, two), 2, lambda t_: n_res / t_)
        sampled("icp_batch_one_pair_us_per_iteration", lambda k: ctx.icp_batch(all_pairs[1:2], fixed, T_init=T_rel[1:2]), 16,
                lambda t_: 1e6 * t_ / (args.icp_iters + 1), digits=2)
        st_icp = ctx.stats()
        rows["icp"]["batch_timeouts"] = int(st_icp.get("icp_batch_timeouts", 0))          # in-kernel waits that hit their bound (0: none)
        rows["icp"]["batch_fallback_pairs"] = int(st_icp.get("icp_batch_fallback_pairs", 0))
        cap_pts = H * W
        xyz_d = torch.empty((cap_pts, 3), dtype=torch.float32, device=dev)
        rgb_d = torch.empty((cap_pts, 3), dtype=torch.uint8, device=dev)
        n_d = torch.zeros(1, dtype=torch.int64, device=dev)
        for sub in (1, 2):
            t_b = timed(lambda k: ctx.backproject_device(k % n_res, xyz_d, rgb_d, n_d, pose=poses[k % n_res], subsample=sub), 64)
            npts = int(n_d.item())
            rows[f"backproject_s{sub}_device_us"] = round(1e6 * t_b, 2)
            rows[f"backproject_s{sub}_GBps"] = round(((4 + 3) * (H // sub) * (W // sub) + 15 * npts) / t_b / 1e9, 1)
        del xyz_d, rgb_d
        # a fused grid to extract from: one pass over the resident frames (TSDF + centroids at the D2R stride)
        ctx.reset()
        for k in range(n_res):
            ctx.integrate(k, poses[k])
            ctx.accumulate_centroid(k, poses[k], subsample=2)
        ctx.sync()
        tq = time.perf_counter()
        pts, _ = ctx.extract(tl3d.EXTRACT_CENTROID)
        rows["extract_centroid_to_host_ms"] = round(1e3 * (time.perf_counter() - tq), 2)
        rows["extract_points"] = int(len(pts))
        tq = time.perf_counter()
        keep = ctx.statistical_outlier(pts, 20, 2.0, cell_size=2.0 * args.voxel)
        rows["outlier_filter_k20_ms"] = round(1e3 * (time.perf_counter() - tq), 2)
        rows["outlier_filter_kept"] = int(keep.sum())
        rows["spread"] = spread
        rows["repetitions"] = REPS
        ctx.reset()

    if rank == 0:
        total = world * total_frames_rank
        # effective stream concurrency of THIS process, measured after everything that is timed (the probe creates streams)
        hwq = abi.probe_hw_queues(local_rank, 16, 1.0)
        out = {
            "metric": "depth frames/sec fused into TSDF (1080x1920)",
            "value": round(total / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{W}x{H} ray-cast orbit (sphere-union object in a closed room: every pixel valid, the heavier "
                                   f"case -- ray-miss pixels only remove bricks; camera radius 1 m) "
                                   f"integrated into a {n}^3 TSDF @ {args.voxel * 1e3:g} mm, 8 B/voxel, "
                                   f"{args.resident_frames} frames per turn of the orbit ({360.0 / max(1, args.resident_frames):.2f} degrees apart on every rank), "
                                   f"{roof['frames_per_sweep'] if roof else 2:g} frames per update launch; frames resident in HBM as "
                                   + ("16-bit millimetres (PNG depth)" if args.depth_format == "u16" else "float32 metres"),
                       "depth_format": args.depth_format,
                       "frames_per_step_per_gpu": F, "resident_frames_per_gpu": n_res, "grid": n,
                       "voxel_m": args.voxel, "centroid_channel": bool(args.centroid), "icp_in_loop": bool(args.icp),
                       "parallelism": f"frame-shard x{world}" + (f" + {n_merges} RCCL all-reduce(s) of the grid" if world > 1 else ""),
                       "grid_merges_in_timed_region": n_merges if dist is not None else 0,
                       "merge_ms": [round(x, 2) for x in merge_ms],
                       "merge_bricks_sent_of_total": [[m["bricks_sent"], m["bricks_total"]] for m in merge_info if m],
                       "merge_bytes": [m["bytes"] for m in merge_info if m],
                       "dist_backend": (backend if dist is not None else None),
                       "dist_ranks": (dist.get_world_size() if dist is not None else 1),
                       "invalid_pixel_fraction": round(invalid_frac, 4),
                       "hw_queues": hwq, "hip": dict(abi.RUNTIME),
                       "setup_s": round(t_gen, 1)},
            "roofline": roof, "cpu_baseline": cpu, "rows": rows,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
