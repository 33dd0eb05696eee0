"""CPU, world_size 2 over gloo: the N > 1 path -- frame shards, the relative-pose gather + ordered prefix product,
and the integer grid all-reduce whose result must equal the single-process grid bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAM = dict(width=96, height=72, fx=84.0, fy=84.0, cx=47.5, cy=35.5)
DIMS, VOXEL = (64, 64, 64), 0.04
N_FRAMES = 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frames():
    from tl3d import synth
    scene = synth.object_scene()
    poses = synth.orbit_poses(N_FRAMES, 1.0, 3.0)
    return poses, [synth.render(scene, p, **CAM) for p in poses]


def _oracle():
    from oracle import c_oracle
    origin = tuple(-0.5 * d * VOXEL for d in DIMS)
    return c_oracle.Oracle(CAM["width"], CAM["height"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], dims=DIMS, origin=origin,
                           voxel_size=VOXEL, sdf_trunc=4 * VOXEL)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tl3d import synth
    from tl3d.distributed import all_gather_relative, allreduce_grid_arrays, chain_poses, pairs_for_rank, shard_range
    poses, frames = _frames()
    orc = _oracle()
    # step 1: local registration of the pairs this rank owns (the oracle's ICP stands in for the device here)
    local = {}
    for prev, curr in pairs_for_rank(N_FRAMES, world, rank):
        r_rel, t_rel = synth.relative_pose(poses[prev], poses[curr])
        T0 = np.eye(4); T0[:3, :3] = r_rel; T0[:3, 3] = t_rel.ravel()
        res = orc.icp(frames[prev][0], orc.normals(frames[curr][0]), T_init=T0, iters=4, stride=2, max_dist=0.1)
        local[curr] = res["T"]
    # collective A: relative poses -> every rank chains the same global poses
    rel = all_gather_relative(local, N_FRAMES, dist)
    chained = chain_poses(rel, first=poses[0])
    # step 2: fuse own frames; collective B: integer grid sum
    lo, hi = shard_range(N_FRAMES, world, rank)
    for i in range(lo, hi):
        orc.tsdf_integrate(frames[i][0], chained[i][0], chained[i][1])
        orc.centroid_accumulate(frames[i][0], frames[i][1], chained[i][0], chained[i][1], subsample=2)
    dense_t, dense_c = orc.tsdf.copy(), orc.centroid.copy()
    info = allreduce_grid_arrays(orc.tsdf, orc.centroid, dist)               # sparse: only the bricks some rank touched travel
    info_d = allreduce_grid_arrays(dense_t, dense_c, dist, sparse=False)
    assert np.array_equal(dense_t, orc.tsdf) and np.array_equal(dense_c, orc.centroid)
    assert info["bricks_total"] == info_d["bricks_sent"] == 512 and 0 < info["bricks_sent"] < 256 and info["bytes"] < info_d["bytes"] / 2
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), tsdf=orc.tsdf, centroid=orc.centroid,
             poses=np.array([np.hstack([r, t.reshape(3, 1)]) for r, t in chained]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_merge_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(a["tsdf"], b["tsdf"]) and np.array_equal(a["centroid"], b["centroid"])
    assert np.array_equal(a["poses"], b["poses"])
    # single process, same poses
    sys.path.insert(0, ROOT)
    poses, frames = _frames()
    orc = _oracle()
    for i in range(N_FRAMES):
        r, t = a["poses"][i][:, :3], a["poses"][i][:, 3]
        orc.tsdf_integrate(frames[i][0], r, t)
        orc.centroid_accumulate(frames[i][0], frames[i][1], r, t, subsample=2)
    assert orc.tsdf[:, 1].sum() > 10000
    assert np.array_equal(a["tsdf"], orc.tsdf) and np.array_equal(a["centroid"], orc.centroid)      # bit-identical merge
    # the chained poses follow the analytic orbit
    for i in range(N_FRAMES):
        assert np.linalg.norm(a["poses"][i][:, :3] - poses[i][0]) + np.linalg.norm(a["poses"][i][:, 3] - poses[i][1].ravel()) < 5e-3


def _chain_worker(rank, world, port, out_dir):
    """The product path's exchange protocol (tl3d.distributed: exchange_registrations / resolve_chain / chain_from_table /
    allreduce_bounds / allreduce_counts) with a stand-in registration: the analytic relative pose, and frame 3 made to fail."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tl3d import distributed as dd, synth
    n = 7
    poses = synth.orbit_poses(n, 1.0, 5.0)

    def register(a, b, fail=False):
        r, t = synth.relative_pose(poses[a], poses[b])
        T = np.eye(4); T[:3, :3] = r; T[:3, 3] = t.ravel()
        return dict(T=T, against=a, fitness=0.9, rmse=1e-4, n_corr=3 if fail else 5000, n_src=6000, iters_run=4, status=0)

    lo, hi = dd.shard_range(n, world, rank)
    local = {b: register(a, b, fail=(b == 3)) for a, b in dd.pairs_for_rank(n, world, rank)}
    table = dd.exchange_registrations(local, n, dist)
    rounds = 0
    for _ in range(n):
        kept, redo = dd.resolve_chain(table, n)
        if redo is None:
            break
        want, cur = redo
        fixed = {cur: register(want, cur)} if lo <= cur < hi else {}       # the owner of `cur` repairs
        dd.exchange_registrations(fixed, n, dist, into=table)
        rounds += 1
    chained, index, log = dd.chain_from_table(table, n)
    mn, mx = dd.allreduce_bounds(np.array([rank, -rank, 5.0]), np.array([rank + 1.0, 0.5, 5.0 + rank]), dist)
    cnt = dd.allreduce_counts([10 + rank, 1], dist)
    # the Sim(3) chain's hand-over (reconstruct_sharded(estimate_scale=True)): the ranks take turns, each advances the state over its
    # own views and hands it on; every rank ends with the same table of rows
    state = dict(prev=0, prev_scale=1.0, avg=1.0, T_guess=np.eye(4))
    rows = {}
    for turn in range(world):
        if turn == rank:
            for cur in range(max(lo, 1), hi):
                ok = cur != 3
                if ok:
                    state.update(prev=cur, prev_scale=1.0 + 0.01 * cur, avg=1.0 + 0.01 * cur, T_guess=register(cur - 1, cur)["T"])
                rows[cur] = dict(T=register(state["prev"] if not ok else cur - 1, cur)["T"], against=cur - 1, ok=ok, fitness=0.5, rmse=1e-3, n_corr=99, iters_run=3,
                                 status=0, scale_raw=1.0 + 0.01 * cur, scale=state["avg"])
        state = dd.handover_sim3_state(state, turn, dist)
    sim3 = dd.exchange_sim3_rows(rows, n, dist)
    np.savez(os.path.join(out_dir, f"chain{rank}.npz"), index=np.array(index), rounds=rounds, mn=mn, mx=mx, cnt=np.array(cnt),
             sim3_prev=state["prev"], sim3_avg=state["avg"], sim3_T=state["T_guess"], sim3_scales=np.array([sim3[c]["scale"] for c in range(1, n)]),
             sim3_ok=np.array([sim3[c]["ok"] for c in range(1, n)]),
             poses=np.array([np.hstack([r, t.reshape(3, 1)]) for r, t in chained]), dropped=np.array([e["dropped"] for e in log]),
             against=np.array([e["against"] for e in log]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_registration_exchange_and_skip_rule(tmp_path):
    """Two ranks register the pairs they own, exchange them, agree on the dropped frame (D2R:598-615: frame 3 fails, frame 4 is
    re-registered against frame 2 by its owner) and chain the same poses; bounds and counters reduce exactly."""
    world = 2
    mp.spawn(_chain_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "chain0.npz"), np.load(tmp_path / "chain1.npz")
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k                                  # every rank took the same decisions
    assert a["index"].tolist() == [0, 1, 2, 4, 5, 6] and int(a["rounds"]) == 1
    assert a["dropped"].tolist() == [False, False, True, False, False, False] and a["against"].tolist() == [0, 1, 2, 2, 4, 5]
    sys.path.insert(0, ROOT)
    from tl3d import synth
    poses = synth.orbit_poses(7, 1.0, 5.0)
    r0, t0 = poses[0]
    for row, k in zip(a["poses"], a["index"]):                               # the analytic orbit, relative to camera 0
        rg = poses[k][0] @ r0.T
        tg = poses[k][1].reshape(3) - rg @ t0.reshape(3)
        assert np.linalg.norm(row[:, :3] - rg) + np.linalg.norm(row[:, 3] - tg) < 1e-12
    assert a["mn"].tolist() == [0.0, -1.0, 5.0] and a["mx"].tolist() == [2.0, 0.5, 6.0] and a["cnt"].tolist() == [21, 2]
    # Sim(3) hand-over: the state after the last rank's turn, and the whole table, on both ranks (compared key by key above)
    assert int(a["sim3_prev"]) == 6 and abs(float(a["sim3_avg"]) - 1.06) < 1e-15
    assert a["sim3_ok"].tolist() == [True, True, False, True, True, True]
    assert np.allclose(a["sim3_scales"], [1.01, 1.02, 1.02, 1.04, 1.05, 1.06], atol=1e-15)
