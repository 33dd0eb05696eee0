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
    parser.add_argument("--tsdf-min-weight", type=int, default=0, help="> 0: keep only voxels the TSDF saw this often")
    parser.add_argument("--ascii", action="store_true", help="write the reference's ASCII fallback PLY instead of binary")
    parser.add_argument("--device", type=int, default=0)
    parser.add_argument("--stream", action="store_true",
                        help="decode on worker threads into pinned buffers with asynchronous uploads; host RAM never holds the sequence")
    args = parser.parse_args(argv)

    from tl3d.config import ReconstructionConfig
    from tl3d.pipeline import DepthToReconstructionPipeline

    config = ReconstructionConfig(fx=args.fx, fy=args.fy, cx=args.cx, cy=args.cy, voxel_size=args.voxel_size,
                                  subsample_factor=args.subsample, depth_scale=args.depth_scale, grid_dim=args.grid,
                                  sdf_trunc_voxels=args.sdf_trunc, icp_iters=args.icp_iters, icp_stride=args.icp_stride,
                                  icp_max_dist=args.icp_max_dist, tsdf_min_weight=args.tsdf_min_weight, device=args.device)
    pipeline = DepthToReconstructionPipeline(config)
    num_loaded = (pipeline.load_data_streaming if args.stream else pipeline.load_data)(args.rgb_folder, args.depth_folder)
    if num_loaded < 2:
        print("Failed to load sufficient data")
        return 0
    anchors = None
    if args.anchors:
        import numpy as np
        z = np.load(args.anchors)
        anchors = {int(k[3:]): (z[k], z["p2_" + k[3:]]) for k in z.files if k.startswith("p3_")}
    points, colors, poses = pipeline.reconstruct(anchors=anchors)
    if points is not None and len(points) > 0:
        pipeline.save_reconstruction(points, colors, args.output, ascii=args.ascii)
        if not args.no_vis:
            print("(interactive Plotly viewer of the reference is not part of the device path; pass --no-vis to silence)")
    else:
        print("Reconstruction failed")
    return 0


if __name__ == "__main__":
    sys.exit(main())
