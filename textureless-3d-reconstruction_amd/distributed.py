"""Multi-GPU: frames shard across ranks, one exchange step at merge time.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests):
  1. rank r registers the consecutive pairs of its frame range [lo, hi) (plus the pair that crosses into it:
     a one-frame halo), locally -- each pair needs only its two depth maps;
  2. all-gather of the relative transforms (a few hundred bytes), then every rank composes the global poses by
     the same ordered prefix product as depth_to_reconstruction.py:618-620, in fp64 on the host;
  3. rank r fuses its own frames into its private grid;
  4. all-reduce(sum) of the grids: the accumulators are integers (int32 TSDF pairs, uint64-packed centroid
     sums), so the merged grid is bit-identical to a single-GPU run whatever the rank count or order.
The reference has no distributed code; nothing here translates an NCCL call pattern.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import _cabi as abi


def shard_range(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous frame range [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pairs_for_rank(n_frames: int, world: int, rank: int) -> List[Tuple[int, int]]:
    """(prev, curr) pairs a rank registers: every pair whose `curr` lies in its range (so the pair that crosses
    the shard boundary belongs to the later rank -- the one-frame halo)."""
    lo, hi = shard_range(n_frames, world, rank)
    return [(i - 1, i) for i in range(max(lo, 1), hi)]


def chain_poses(relative: Sequence[np.ndarray], first=None):
    """Global world->camera poses from relative transforms T_i (camera i-1 -> camera i), i = 1..n-1:
    R_i = R_rel R_{i-1}, t_i = R_rel t_{i-1} + t_rel (D2R:618-620)."""
    r, t = (np.eye(3), np.zeros((3, 1))) if first is None else (np.asarray(first[0], np.float64), np.asarray(first[1], np.float64).reshape(3, 1))
    poses = [(r, t)]
    for T in relative:
        T = np.asarray(T, np.float64).reshape(4, 4)
        r, t = T[:3, :3] @ r, T[:3, :3] @ t + T[:3, 3:4]
        poses.append((r, t))
    return poses


def all_gather_relative(local: dict, n_frames: int, dist=None) -> List[np.ndarray]:
    """local: {curr_index: 4x4}.  Returns the list T_1..T_{n-1} on every rank (identity where a pair failed)."""
    import torch
    buf = torch.zeros((n_frames, 17), dtype=torch.float64)
    for i, T in local.items():
        buf[i, :16] = torch.from_numpy(np.asarray(T, np.float64).reshape(16))
        buf[i, 16] = 1.0
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dev = None
        if dist.get_backend() == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            buf = buf.to(dev)
        dist.all_reduce(buf)                  # disjoint rows: a sum is a gather
        buf = buf.cpu()
    out = []
    for i in range(1, n_frames):
        out.append(buf[i, :16].numpy().reshape(4, 4).copy() if buf[i, 16] > 0.5 else np.eye(4))
    return out


def allreduce_grid_arrays(tsdf: np.ndarray = None, centroid: np.ndarray = None, dist=None, sparse: bool = True):
    """Host-array form of the merge (gloo tests, checkpoints): in-place sum over ranks.  sparse: as allreduce_context_grids --
    one byte per brick MAX-all-reduced, then only the bricks any rank holds something in are summed.  Returns the account
    (bricks_sent, bricks_total, bytes)."""
    import torch
    arrs = [(a, dt) for a, dt in ((tsdf, torch.int32), (centroid, torch.int64)) if a is not None]
    if not arrs:
        return None
    nbr = (tsdf.shape[0] if tsdf is not None else centroid.shape[0]) // 512
    row_bytes = (4096 if tsdf is not None else 0) + (16384 if centroid is not None else 0)
    idx = None
    if sparse:
        m = np.zeros(nbr, np.uint8)
        if tsdf is not None:
            m |= (tsdf.reshape(nbr, 512, 2)[:, :, 1] != 0).any(1).astype(np.uint8)
        if centroid is not None:
            m |= ((centroid.reshape(nbr, 512, 4)[:, :, 1] >> np.uint64(32)) != 0).any(1).astype(np.uint8)
        mt = torch.from_numpy(m)
        dist.all_reduce(mt, op=dist.ReduceOp.MAX)
        idx = np.nonzero(m)[0]
        if 2 * len(idx) >= nbr:
            idx = None
    for arr, dt in arrs:
        flat = arr.view(np.int32 if dt == torch.int32 else np.int64).reshape(nbr, -1)
        if idx is None:
            dist.all_reduce(torch.from_numpy(flat))
        elif len(idx):
            block = torch.from_numpy(np.ascontiguousarray(flat[idx]))
            dist.all_reduce(block)
            flat[idx] = block.numpy()
    n = nbr if idx is None else len(idx)
    return dict(bricks_sent=int(n), bricks_total=int(nbr), bytes=int(n * row_bytes + (nbr if idx is not None else 0)))


def _headroom_check(ctx, dist):
    """int32 headroom (tl3d.h: TL3D_TSDF_MAX_WEIGHT): the sum over ranks of each rank's largest voxel weight bounds the merged
    grid's; refuse the merge instead of wrapping."""
    import torch
    w = _reduce(torch.tensor([ctx.max_weight()], dtype=torch.int64), dist)
    if int(w.item()) > abi.TSDF_MAX_WEIGHT:
        raise OverflowError(f"merged TSDF grid could hold {int(w.item())} observations per voxel (limit {abi.TSDF_MAX_WEIGHT}): "
                            "merge more often or extract between scans")


def allreduce_context_grids(ctx, dist, sparse: bool = True) -> dict:
    """Device form of the merge: all-reduce on the library's own grid memory.

    sparse (default): what travels is what any rank OBSERVED, at the granularity of a 4x4x4 sub-brick (64 contiguous records) and
    per channel: one byte per sub-brick MAX-all-reduced gives every rank the same list, the listed sub-bricks are packed into one
    block (tl3d_grid_pack_bricks with TL3D_CH_SUB), SUM-all-reduced and unpacked.  A frame-sharded run of a few dozen frames per
    rank touches a few per cent of a 1024^3 grid (BASELINE config 5: 32 frames per rank at 8 GPUs), and of a touched brick the
    centroid channel fills one or two sub-bricks (a surface crosses it): whole 16-KB bricks were 6.3 of the 7.9 GB a rank sent
    in round 3.  A channel of which more than half is touched is all-reduced whole (zero-copy view; a sparse grid: every
    sub-brick it holds).  Works with any backend: RCCL reduces the device tensors in place; gloo takes them through the host.

    Ordering: every torch operation here runs on the CONTEXT's stream (torch.cuda.ExternalStream over tl3d_get_stream), so the
    library's kernels and torch's stay in program order without a device-wide wait between them (round 3 bracketed each step with
    torch.cuda.synchronize()); a collective is ordered against that stream by torch.distributed as against any current stream.
    Returns what went over the wire: sub_bricks_tsdf / sub_bricks_centroid (eighths of a brick), bricks_sent (the larger of the two,
    in bricks), bricks_total, bytes (records + maps + counts)."""
    import torch
    dev = torch.device("cuda", ctx.device)
    on_device = dist.get_backend() == "nccl"
    ch = ctx.grid.channels
    nbr = ctx.n_bricks
    stream = torch.cuda.ExternalStream(ctx.stream_ptr(), device=dev)

    def reduce_(t, op=None):
        kw = {} if op is None else {"op": op}
        if on_device:
            dist.all_reduce(t, **kw)
            return t
        h = t.cpu()                                           # (a blocking copy on the current = the context's stream)
        dist.all_reduce(h, **kw)
        t.copy_(h)
        return t

    with torch.cuda.stream(stream):
        if ch & abi.CH_TSDF:
            _headroom_check(ctx, dist)
        # The free-space observations of a frame-sharded run cover most of the volume between the cameras and the surfaces, but they
        # are ONE count per brick (tl3d.h: TL3D_CH_FREE), pending until something reads the channel.  They travel as what they are --
        # 4 bytes per brick, summed -- and only sub-bricks with RECORDS (a surface came within the truncation band, or a point fell in)
        # travel as records.
        free_apart = bool(ch & abi.CH_TSDF)
        fl = abi.CH_FREE if free_apart else 0
        sent = {}
        nbytes = 0
        if free_apart:
            cnt = ctx.grid_tensor(abi.CH_FREE)
            reduce_(cnt)                                      # every rank now holds the scan's counts, still pending
            nbytes += 4 * nbr
        for channel, row_words, dt in ((abi.CH_TSDF, 128, torch.int32), (abi.CH_CENTROID, 256, torch.int64)):
            if not ch & channel:
                continue
            chf = channel | (fl if channel == abi.CH_TSDF else 0)
            row_bytes = row_words * (4 if dt == torch.int32 else 8)
            idx = None
            if sparse:
                m = torch.zeros(8 * nbr, dtype=torch.uint8, device=dev)
                ctx.touched_bricks(m, chf | abi.CH_SUB)
                reduce_(m, dist.ReduceOp.MAX)
                nbytes += 8 * nbr
                idx = torch.nonzero(m, as_tuple=False).flatten().to(torch.int32)
                del m
                if 2 * idx.numel() >= 8 * nbr and not ctx.grid.sparse:
                    idx = None
            if idx is None:
                if ctx.grid.sparse:                           # no dense layout to view: every sub-brick of the channel
                    idx = torch.arange(8 * nbr, dtype=torch.int32, device=dev)
                else:
                    t = ctx.grid_tensor(channel)              # (TSDF: folds the pending counts into the records first, on the context's stream)
                    reduce_(t)
                    sent[channel] = 8 * nbr
                    nbytes += 8 * nbr * row_bytes
                    continue
            n = int(idx.numel())
            # blocks of at most 2 GiB: a 1024^3 scan's union can be several GB per channel, and the block is a copy
            step = max(1, (2 << 30) // row_bytes)
            for i0 in range(0, n, step):
                part = idx[i0:i0 + step]
                block = torch.empty((int(part.numel()), row_words), dtype=dt, device=dev)
                ctx.pack_bricks(chf | abi.CH_SUB, part, block)
                reduce_(block)
                ctx.unpack_bricks(chf | abi.CH_SUB, part, block)
                del block
            sent[channel] = n
            nbytes += n * row_bytes
        stream.synchronize()                                  # the caller's own torch work may sit on another stream
    return dict(bricks_sent=(max(sent.values(), default=0) + 7) // 8, bricks_total=nbr, bytes=int(nbytes),
                sub_bricks_tsdf=int(sent.get(abi.CH_TSDF, 0)), sub_bricks_centroid=int(sent.get(abi.CH_CENTROID, 0)))


# ---------------------------------------------------------------------------------------------------------------------
# product path: DepthToReconstructionPipeline.reconstruct_sharded (pipeline.py) is built from these
# ---------------------------------------------------------------------------------------------------------------------
_COLS = 24          # 16 T | against | ok | fitness | rmse | n_corr | iters_run | status | present


def _world(dist):
    return 1 if dist is None or not dist.is_initialized() else dist.get_world_size()


def _reduce(t, dist, op=None):
    """all_reduce of a CPU tensor through whatever backend the group has (RCCL wants it on the GPU)."""
    import torch
    if _world(dist) == 1:
        return t
    if dist.get_backend() == "nccl":
        d = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(d, op=op if op is not None else dist.ReduceOp.SUM)
        return d.cpu()
    dist.all_reduce(t, op=op if op is not None else dist.ReduceOp.SUM)
    return t


def exchange_registrations(local: dict, n_frames: int, dist=None, into: dict = None):
    """local: {curr: result dict of FusionContext.icp_collect + 'against'} for the pairs THIS rank registered.  Every rank
    gets the union (disjoint rows of one small tensor, sum = gather).  Returns the table {curr: entry}; with `into`, the
    received rows replace the table's and the return value tells whether anything arrived."""
    import torch
    buf = torch.zeros((n_frames, _COLS), dtype=torch.float64)
    for cur, r in local.items():
        row = buf[cur]
        row[:16] = torch.from_numpy(np.ascontiguousarray(np.asarray(r["T"], np.float64).reshape(16)))
        ok = not (r["status"] == 2 or r["n_corr"] < 8)
        row[16:24] = torch.tensor([float(r["against"]), float(ok), float(r["fitness"]), float(r["rmse"]), float(r["n_corr"]),
                                   float(r["iters_run"]), float(r["status"]), 1.0], dtype=torch.float64)
    buf = _reduce(buf, dist).numpy()
    table = {} if into is None else into
    got = False
    for cur in range(n_frames):
        if buf[cur, 23] > 0.5:
            got = True
            table[cur] = dict(T=buf[cur, :16].reshape(4, 4).copy(), against=int(buf[cur, 16]), ok=bool(buf[cur, 17] > 0.5), fitness=float(buf[cur, 18]),
                              rmse=float(buf[cur, 19]), n_corr=int(buf[cur, 20]), iters_run=int(buf[cur, 21]), status=int(buf[cur, 22]))
    return got if into is not None else table


def handover_sim3_state(state: dict, src: int, dist=None) -> dict:
    """The state of the Sim(3) registration chain (pipeline._sim3_chain: last kept view and its scale, running scale, motion prior)
    as rank `src` leaves it, on every rank: the next rank continues the chain from it.  19 doubles."""
    import torch
    if _world(dist) == 1:
        return state
    buf = torch.zeros(19, dtype=torch.float64)
    if dist.get_rank() == src:
        buf[0], buf[1], buf[2] = float(state["prev"]), float(state["prev_scale"]), float(state["avg"])
        buf[3:19] = torch.from_numpy(np.ascontiguousarray(np.asarray(state["T_guess"], np.float64).reshape(16)))
    buf = _reduce(buf, dist).numpy()                       # (zeros from everybody else: a sum is a broadcast)
    return dict(prev=int(round(buf[0])), prev_scale=float(buf[1]), avg=float(buf[2]), T_guess=buf[3:19].reshape(4, 4).copy())


def exchange_sim3_rows(local: dict, n_frames: int, dist=None) -> dict:
    """The rows of pipeline._sim3_chain every rank produced for the views it owns -> the whole table on every rank."""
    import torch
    cols = 27          # 16 T | against | ok | fitness | rmse | n_corr | iters_run | status | scale_raw | scale | present
    buf = torch.zeros((n_frames, cols), dtype=torch.float64)
    for cur, r in local.items():
        buf[cur, :16] = torch.from_numpy(np.ascontiguousarray(np.asarray(r["T"], np.float64).reshape(16)))
        buf[cur, 16:27] = torch.tensor([float(r["against"]), float(r["ok"]), float(r["fitness"]), float(r["rmse"]), float(r["n_corr"]),
                                        float(r["iters_run"]), float(r["status"]), float(r["scale_raw"]), float(r["scale"]), 0.0, 1.0],
                                       dtype=torch.float64)
    buf = _reduce(buf, dist).numpy()
    table = {}
    for cur in range(n_frames):
        if buf[cur, 26] > 0.5:
            table[cur] = dict(T=buf[cur, :16].reshape(4, 4).copy(), against=int(buf[cur, 16]), ok=bool(buf[cur, 17] > 0.5), fitness=float(buf[cur, 18]),
                              rmse=float(buf[cur, 19]), n_corr=int(buf[cur, 20]), iters_run=int(buf[cur, 21]), status=int(buf[cur, 22]),
                              scale_raw=float(buf[cur, 23]), scale=float(buf[cur, 24]))
    return table


def resolve_chain(table: dict, n_frames: int):
    """The reference's skip rule (D2R:598-615) applied to a table of pair registrations: frames are visited in order; a
    frame must have been registered against the LAST KEPT frame; a failed registration drops the frame.  Returns (kept frame
    indices, (want, curr) of the first frame that still has to be re-registered or None).  Pure function of the table, so
    every rank takes the same decisions."""
    kept = [0]
    for cur in range(1, n_frames):
        e = table.get(cur)
        if e is None or e["against"] != kept[-1]:
            return kept, (kept[-1], cur)
        if e["ok"]:
            kept.append(cur)
    return kept, None


def chain_from_table(table: dict, n_frames: int):
    """(poses, frame_index, log) of the kept frames: cam0 = (I, 0), then R = R_rel R_prev, t = R_rel t_prev + t_rel (D2R:618-620)."""
    kept, redo = resolve_chain(table, n_frames)
    assert redo is None, "chain_from_table needs a resolved table"
    pose = {0: (np.eye(3), np.zeros((3, 1)))}
    log = []
    for cur in range(1, n_frames):
        e = table[cur]
        log.append(dict(frame=cur, against=e["against"], fitness=e["fitness"], rmse=e["rmse"], n_corr=e["n_corr"], iters_run=e["iters_run"],
                        status=e["status"], dropped=not e["ok"]))
        if e["ok"]:
            r_prev, t_prev = pose[e["against"]]
            T = e["T"]
            pose[cur] = (T[:3, :3] @ r_prev, T[:3, :3] @ t_prev + T[:3, 3:4])
    return [pose[k] for k in kept], kept, log


def allreduce_bounds(mn, mx, dist=None):
    """Scene bounds over all ranks (exact: MIN / MAX), so that every rank plans the same grid."""
    import torch
    if _world(dist) == 1:
        return np.asarray(mn, np.float64), np.asarray(mx, np.float64)
    lo = _reduce(torch.from_numpy(np.asarray(mn, np.float64).copy()), dist, dist.ReduceOp.MIN).numpy()
    hi = _reduce(torch.from_numpy(np.asarray(mx, np.float64).copy()), dist, dist.ReduceOp.MAX).numpy()
    return lo, hi


def allreduce_counts(values, dist=None):
    import torch
    t = _reduce(torch.tensor([int(v) for v in values], dtype=torch.int64), dist)
    return [int(v) for v in t.tolist()]


def merge_context_grids(ctx, dist=None, sparse: bool = True):
    """Sum the per-rank grids into every rank's grid (integer sums: the result does not depend on the backend, the rank count,
    the order or the sparse / dense form).  Returns allreduce_context_grids' account of what travelled (None with one rank)."""
    if _world(dist) == 1:
        return None
    return allreduce_context_grids(ctx, dist, sparse=sparse)
