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


def allreduce_grid_arrays(tsdf: np.ndarray = None, centroid: np.ndarray = None, dist=None):
    """Host-array form of the merge (gloo tests, checkpoints): in-place sum over ranks."""
    import torch
    for arr, dt in ((tsdf, torch.int32), (centroid, torch.int64)):
        if arr is None:
            continue
        t = torch.from_numpy(arr.view(np.int32 if dt == torch.int32 else np.int64))
        dist.all_reduce(t)
    return tsdf, centroid


def allreduce_context_grids(ctx, dist) -> None:
    """Device form: RCCL all-reduce directly on the library's grid memory (zero-copy torch views)."""
    import torch
    ctx.sync()
    if ctx.grid.channels & abi.CH_TSDF:
        # int32 headroom (tl3d.h: TL3D_TSDF_MAX_WEIGHT): the sum over ranks of each rank's largest voxel weight bounds the
        # merged grid's; refuse the merge instead of wrapping
        w = torch.tensor([ctx.max_weight()], dtype=torch.int64, device=torch.device("cuda", ctx.device))
        dist.all_reduce(w)
        if int(w.item()) > abi.TSDF_MAX_WEIGHT:
            raise OverflowError(f"merged TSDF grid could hold {int(w.item())} observations per voxel (limit {abi.TSDF_MAX_WEIGHT}): "
                                "merge more often or extract between scans")
        dist.all_reduce(ctx.grid_tensor(abi.CH_TSDF))
    if ctx.grid.channels & abi.CH_CENTROID:
        dist.all_reduce(ctx.grid_tensor(abi.CH_CENTROID))
    torch.cuda.synchronize()
