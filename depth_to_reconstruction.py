#!/usr/bin/env python3
"""Drop-in command line for the reference's depth_to_reconstruction.py, running its dense back end on an MI355X.

    python depth_to_reconstruction.py --rgb-folder R --depth-folder D --output out.ply --fx 525 --fy 525 --cx 320 --cy 240

Same flags and defaults as the reference's main() (depth_to_reconstruction.py:770-786) and the same .ply out.
Poses come from on-device point-to-plane ICP (the reference used SIFT + essential matrix), fusion is on-device
voxel accumulation (the reference used np.vstack + Open3D).  Additive flags configure those two stages.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _spawn_ranks(n, argv):
    """--gpus N outside torchrun: start N copies of this command line, one per GPU, BEFORE anything touches a GPU here
    (children are fresh processes; the parent only watches them: the first rank that fails -- by a signal too -- ends the group
    with a non-zero status, the wait is bounded).  Rendezvous on 127.0.0.1."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("tl3d_launch", os.path.join(here, "textureless-3d-reconstruction_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.spawn_ranks(os.path.abspath(__file__), list(argv), n)


def _init_rank(args):
    """This process is one rank: bind it to its GPU and join the group (backend "nccl" = RCCL over xGMI; TL3D_DIST_BACKEND=gloo
    and TL3D_SHARE_DEVICE=1 let several ranks share one GPU for tests)."""
    import tl3d  # noqa: F401  (GPU_MAX_HW_QUEUES before the first HIP call)
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.device = 0 if os.environ.get("TL3D_SHARE_DEVICE") == "1" else local
    torch.cuda.set_device(args.device)
    backend = os.environ.get("TL3D_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", args.device))
    else:
        dist.init_process_group(backend)
    return dist


def main(argv=None):
    parser = argparse.ArgumentParser(description="Depth to 3D Reconstruction")
    parser.add_argument("--rgb-folder", type=str, required=True, help="Folder with RGB images")
    parser.add_argument("--depth-folder", type=str, required=True, help="Folder with depth images")
    parser.add_argument("--output", type=str, default="./output/reconstruction.ply", help="Output PLY file path")
    parser.add_argument("--fx", type=float, default=1719.0)
    parser.add_argument("--fy", type=float, default=1719.0)
    parser.add_argument("--cx", type=float, default=540.0)
    parser.add_argument("--cy", type=float, default=960.0)
    parser.add_argument("--voxel-size", type=float, default=0.005)
    parser.add_argument("--subsample", type=int, default=2)
    parser.add_argument("--no-vis", action="store_true")
    # additive
    parser.add_argument("--grid", type=int, default=1024,
                        help="fusion volume budget: at most GRID^3 voxels in total, spread over the axes as the scene needs")
    parser.add_argument("--sdf-trunc", type=float, default=4.0, help="TSDF truncation in voxels")
    parser.add_argument("--icp-iters", type=int, default=15)
    parser.add_argument("--icp-stride", type=int, default=2)
    parser.add_argument("--icp-max-dist", type=float, default=0.05)
    parser.add_argument("--depth-scale", type=float, default=1.0, help="metric scale of the depth files")
    parser.add_argument("--anchors", type=str, default=None,
                        help=".npz of sparse metric anchors (p3_<i>: [n,3], p2_<i>: [n,2] per frame i) for relative depth")
    parser.add_argument("--estimate-scale", action="store_true",
                        help="relative depth without anchors: every view's metric scale is estimated inside its registration (Sim(3) "
                             "point-to-plane ICP) and chained with the reference's 0.7 / 0.3 rule (D2R:650); view 0 keeps --depth-scale")
    parser.add_argument("--scale-update-weight", type=float, default=0.3, help="weight of a view's own scale estimate in the running scale")
    parser.add_argument("--tsdf-min-weight", type=int, default=0, help="> 0: keep only voxels the TSDF saw this often")
    parser.add_argument("--ascii", action="store_true", help="write the reference's ASCII fallback PLY instead of binary")
    parser.add_argument("--device", type=int, default=0)
    parser.add_argument("--stream", dest="stream", action="store_true", default=True,
                        help="(default) decode on worker threads into pinned buffers with asynchronous uploads; host RAM never holds the sequence")
    parser.add_argument("--no-stream", dest="stream", action="store_false",
                        help="decode every frame into host memory first, as the reference does (D2R:434-437)")
    parser.add_argument("--gpus", type=int, default=1,
                        help="one process per GPU: frames shard across ranks, the per-GPU grids are summed with one RCCL all-reduce "
                             "(also honoured under torchrun: RANK / WORLD_SIZE in the environment)")
    args = parser.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        return _spawn_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv))
    dist = None
    if world > 1:
        dist = _init_rank(args)

    from tl3d.config import ReconstructionConfig
    from tl3d.pipeline import DepthToReconstructionPipeline

    config = ReconstructionConfig(fx=args.fx, fy=args.fy, cx=args.cx, cy=args.cy, voxel_size=args.voxel_size,
                                  subsample_factor=args.subsample, depth_scale=args.depth_scale, grid_dim=args.grid,
                                  sdf_trunc_voxels=args.sdf_trunc, icp_iters=args.icp_iters, icp_stride=args.icp_stride,
                                  icp_max_dist=args.icp_max_dist, tsdf_min_weight=args.tsdf_min_weight, device=args.device,
                                  scale_update_weight=args.scale_update_weight)
    pipeline = DepthToReconstructionPipeline(config)
    # a rank decodes every frame on its host (pose chain and scale rule run over the whole sequence) and uploads its share
    streaming = args.stream and dist is None
    if dist is not None and dist.get_rank() != 0:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            num_loaded = pipeline.load_data(args.rgb_folder, args.depth_folder)
    else:
        num_loaded = (pipeline.load_data_streaming if streaming else pipeline.load_data)(args.rgb_folder, args.depth_folder)
    if num_loaded < 2:
        if dist is None or dist.get_rank() == 0:
            print("Failed to load sufficient data")
        if dist is not None:
            dist.destroy_process_group()
        return 0
    anchors = None
    if args.anchors:
        import numpy as np
        z = np.load(args.anchors)
        anchors = {int(k[3:]): (z[k], z["p2_" + k[3:]]) for k in z.files if k.startswith("p3_")}
    if dist is not None:
        points, colors, poses = pipeline.reconstruct_sharded(dist, anchors=anchors, estimate_scale=args.estimate_scale)
        rank = dist.get_rank()
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:                                   # rank 0 holds the merged cloud and writes the file
            return 0
    else:
        points, colors, poses = pipeline.reconstruct(anchors=anchors, estimate_scale=args.estimate_scale)
    if points is not None and len(points) > 0:
        pipeline.save_reconstruction(points, colors, args.output, ascii=args.ascii)
        if not args.no_vis:
            print("(interactive Plotly viewer of the reference is not part of the device path; pass --no-vis to silence)")
    else:
        print("Reconstruction failed")
    return 0


if __name__ == "__main__":
    sys.exit(main())
