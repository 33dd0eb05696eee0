"""Worker of test_two_ranks_merge_their_device_grids_to_the_oracle_sum: one rank of a 2-rank gloo group, both ranks on ONE GPU
(TL3D_SHARE_DEVICE=1).  Each rank fuses its half of the frames into its own context (dense, or sparse with pools counted for
the UNION -- a rank also receives the bricks only the other one touched), the product's merge sums the device grids, and every
rank compares what it then holds with the C oracle run over ALL frames.  argv: out_dir layout(dense|sparse)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, layout = sys.argv[1], sys.argv[2]
    import tl3d
    import torch
    import torch.distributed as dist
    from helpers import SMALL, small_scene_frames
    from oracle import c_oracle
    from tl3d import synth
    from tl3d.distributed import merge_context_grids, shard_range
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    n = 8
    poses, frames = small_scene_frames(n=n, deg=6.0, scene=synth.object_scene(with_room=True))
    dims, voxel, centre = (128, 128, 128), 0.02, (0.0, -0.1, 0.0)
    origin = tuple(centre[i] - 0.5 * dims[i] * voxel for i in range(3))
    lo, hi = shard_range(n, world, rank)
    ctx = tl3d.FusionContext(SMALL["width"], SMALL["height"], SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"], n_slots=n, grid=None)
    with ctx:
        for i in range(n):                                     # (every frame resident: the union is counted below)
            ctx.upload(i, frames[i][0], frames[i][1])
        geom = tl3d.GridSpec(dims, origin, voxel, 4 * voxel, tl3d.CH_TSDF | tl3d.CH_CENTROID)
        if layout == "sparse":
            nt, nc = ctx.count_bricks(geom, list(range(n)), poses, centroid_subsample=2)
            geom = tl3d.GridSpec(dims, origin, voxel, 4 * voxel, geom.channels, pool_tsdf=nt + 4, pool_centroid=nc + 4)
        ctx.attach_grid(geom)
        for i in range(lo, hi):
            ctx.integrate(i, poses[i])
            ctx.accumulate_centroid(i, poses[i], subsample=2)
        own = ctx.stats()
        info = merge_context_grids(ctx, dist)
        t, c = ctx.download_grid(tl3d.CH_TSDF), ctx.download_grid(tl3d.CH_CENTROID)
        st = ctx.stats()
    orc = c_oracle.Oracle(SMALL["width"], SMALL["height"], SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"], dims=dims, origin=origin,
                          voxel_size=voxel, sdf_trunc=4 * voxel)
    for i in range(n):
        orc.tsdf_integrate(frames[i][0], poses[i][0], poses[i][1])
        orc.centroid_accumulate(frames[i][0], frames[i][1], poses[i][0], poses[i][1], subsample=2)
    ok = bool(np.array_equal(t, orc.tsdf) and np.array_equal(c, orc.centroid))
    grew = st["pool_slots_centroid"] > own["pool_slots_centroid"] if layout == "sparse" else True      # bricks only the other rank saw arrived
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{int(ok)} {int(grew)} {st['pool_refused']} {info['sub_bricks_tsdf']} {info['sub_bricks_centroid']} {info['bricks_total']} {info['bytes']}\n")
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
